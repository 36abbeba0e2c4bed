#!/usr/bin/env python3
"""One-GPU probe of the per-iteration cost of the strip protocol (what each rank of the N-GPU bench executes):
(a) plain batch call, (b) the Python split-step loop without communication, (c) with RCCL traffic to self
(send/recv of the halo rows + all-reduce of the CFL scalar in a 1-rank group).  Timing only: (c) scrambles ghosts."""
import os, sys, time, socket
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hipims-ocl_amd"))
import numpy as np
import torch
import torch.distributed as dist
import hipims_mi as hp
from hipims_mi import strips, synthetic as syn

cols, rows, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, int(sys.argv[2]) if len(sys.argv) > 2 else 1026, 300
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
st, bed, man = syn.s_dam(cols, rows)
r = strips.StripRunner(cols, rows, rank=0, world=1, loop="torch")
r.upload_global(st, bed, man); r.set_target_time(1e9)
def timed(fn, n):
    fn(20); r.barrier(); t0 = time.perf_counter(); fn(n); r.barrier(); return (time.perf_counter() - t0) / n * 1e6
print("grid %dx%d" % (cols, rows))
print("(a) hp_step_batch            : %.1f us/step" % timed(lambda n: r.domain.step_batch(n), steps))
print("(b) python split-step loop    : %.1f us/step" % timed(lambda n: r.step(n), steps))
def comm_loop(n):
    g = r.g
    for _ in range(n):
        r.engine.step_begin()
        new = r.engine.new_state(); m = new.shape[0]
        ops = [dist.P2POp(dist.isend, new[g:2*g], 0), dist.P2POp(dist.irecv, new[0:g], 0),
               dist.P2POp(dist.isend, new[m-2*g:m-g], 0), dist.P2POp(dist.irecv, new[m-g:m], 0)]
        with r.engine.halo_context():
            reqs = dist.batch_isend_irecv(ops)
        if r.engine.needs_reduction():
            dist.all_reduce(r.engine.cfl_slot(), op=dist.ReduceOp.MAX)
        for q in reqs: q.wait()
        r.engine.step_end()
print("(c) + self P2P + all-reduce   : %.1f us/step" % timed(comm_loop, steps))
r.engine.set_halo_overlap(True)
print("(d) split launch, no comm     : %.1f us/step" % timed(lambda n: r.step(n), steps))
print("(e) (c) with halo overlap     : %.1f us/step" % timed(comm_loop, steps))
t0 = time.perf_counter(); comm_loop(200); cpu = (time.perf_counter() - t0) / 200 * 1e6; r.barrier()
print("CPU enqueue cost of (e)       : %.1f us/step (no sync)" % cpu)
r.engine.set_halo_overlap(False)
t0 = time.perf_counter()
for _ in range(200):
    r.engine.step_begin(); r.engine.needs_reduction(); r.engine.step_end()
cpu = (time.perf_counter() - t0) / 200 * 1e6
r.barrier()
print("CPU enqueue cost of (b)       : %.1f us/step (no sync)" % cpu)
r.close()
