#!/usr/bin/env python3
"""One-GPU probe of what ONE rank of a strong-scaling run executes per iteration: the 4096 x 514 strip of the 8-GPU
split of 4096^2 (default; any shape as argv), through (a) the plain batch call, (b) the library's own strip loop
(hp_strip_step_batch over a 1-rank RCCL communicator: the all-reduce is real, there is no neighbour), (c) the same with
the step split into halo and interior launches on two streams, (d), (e) = (b), (c) with the maximum over the strips through
the peer-written mailboxes instead of the all-reduce.  Prints device time per iteration and the host's
enqueue cost per iteration (the call returns without waiting)."""
import os, sys, time, socket
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hipims-ocl_amd"))
import numpy as np
import torch
import hipims_mi as hp
from hipims_mi import strips, synthetic as syn

cols, rows = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 514)
period = int(os.environ.get("PERIOD", "1"))          # 2: two reaches of ghost rows, halo / interior split on every second iteration only
steps = 1500
os.environ["HP_STRIP_REDUCE_ALWAYS"] = "1"            # the 1-rank communicator still reduces: (b), (c) pay the library's all-reduce
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
st, bed, man = syn.s_dam(cols, rows)
r = strips.StripRunner(cols, rows, rank=0, world=1, loop="cxx", exchange_period=period)
r.upload_global(st, bed, man); r.set_target_time(1e9)

def timed(fn, n):
    best = None
    for _ in range(3):                                # the best of three: the variants are compared to a microsecond
        fn(50); r.barrier()
        t0 = time.perf_counter(); fn(n); host = (time.perf_counter() - t0) / n * 1e6
        r.barrier(); dev = (time.perf_counter() - t0) / n * 1e6
        best = (dev, host) if best is None or dev < best[0] else best
    return best

print("strip %d x %d (%.2f Mcell), exchange every %d iteration(s)" % (cols, rows, cols * rows / 1e6, period))
for name, fn, overlap in (("(a) hp_step_batch", lambda n: r.domain.step_batch(n), False),
                          ("(b) hp_strip_step_batch", lambda n: r.domain.strip_step_batch(n), False),
                          ("(c) hp_strip_step_batch, split launches", lambda n: r.domain.strip_step_batch(n), True)):
    r.domain.set_halo_overlap(overlap)
    dev, host = timed(fn, steps)
    print("%-42s: %6.1f us/iteration on the device, %5.1f us/iteration host enqueue" % (name, dev, host))
# (d): as (c), the maximum over the strips through the peer-written mailboxes instead of the library's all-reduce (one rank:
# the advance kernel writes to and polls its OWN mailbox -- the in-kernel cost of the mechanism without the xGMI hop)
assert r.domain.strip_peer_connect([r.domain.strip_peer_ticket()], 0)
for name, overlap in (("(d) as (b), maximum through the mailboxes", False), ("(e) as (c), maximum through the mailboxes", True)):
    r.domain.set_halo_overlap(overlap)
    dev, host = timed(lambda n: r.domain.strip_step_batch(n), steps)
    print("%-42s: %6.1f us/iteration on the device, %5.1f us/iteration host enqueue" % (name, dev, host))
r.close()
