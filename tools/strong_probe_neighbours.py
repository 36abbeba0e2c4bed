#!/usr/bin/env python3
"""Round 4 companion of strong_probe_pair.py.  The pair probe cuts the 4096 x 514 strip of the 8-GPU split into two halves that
SHARE the GPU -- two concurrent launches, each tiled as if it had the chip to itself -- which is not what a GPU of an 8-GPU node
sees.  Here the strip stays WHOLE and gets two real neighbours: three strips, ranks as threads of this process on the one GPU,
the outer two only `thin` rows tall (their launches are a few dozen blocks that ride along), the middle one the 512 owned rows
+ 2 ghost rows of the split.  The middle strip's iteration is then one launch of the full strip's shape with a hand-over at both
ends: ghost rows stored into the neighbours, two mailbox writers, the tail block's round -- everything but the xGMI hop.
usage: strong_probe_neighbours.py [cols owned_rows thin]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hipims-ocl_amd"))
os.environ["HIPIMS_MI_NO_TORCH"] = "1"
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
import hipims_mi as hp
from hipims_mi import synthetic as syn

cols, owned, thin = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (4096, 512, 8)
steps = int(os.environ.get("PROBE_STEPS", "1500"))
rows = thin + owned + thin
st, bed, man = syn.s_dam(cols, rows)
lib = hp.load_library()
hp._check(lib, lib.hp_comm_load(os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so").encode()), "hp_comm_load")

single = hp.Domain(cols, owned + 2)
s1, b1, m1 = syn.s_dam(cols, owned + 2)
single.upload(s1, b1, m1); single.set_target_time(1e9); single.step_batch(50); single.sync()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); single.step_batch(steps); single.sync(); best = min(best, (time.perf_counter() - t0) / steps * 1e6)
print("strip %d x %d alone, one hp_step_batch                               : %6.1f us/iteration" % (cols, owned + 2, best), flush=True)
single.close()

world, g = 3, 1
own = [(0, thin), (thin, thin + owned), (thin + owned, rows)]
parts = [(lo, hi, max(0, lo - g), min(rows, hi + g)) for lo, hi in own]
for level in (2,):
    os.environ["HP_PEER_DIRECT"] = "1"
    uid = hp.comm_unique_id()
    meet = threading.Barrier(world)
    tickets, result = [None] * world, [None] * world

    def rank_main(r):
        own_lo, own_hi, lo, hi = parts[r]
        dom = hp.Domain(cols, hi - lo, global_rows=rows, row_offset=lo)
        dom.upload(st[lo:hi], bed[lo:hi], man[lo:hi])
        dom.set_halo_overlap(False)
        dom.strip_comm_init(uid, r, world)
        tickets[r] = dom.strip_peer_ticket(); meet.wait()
        got = dom.strip_peer_connect(tickets, r)
        dom.set_target_time(1e9); meet.wait()
        dom.strip_update_timestep()
        dom.strip_step_batch(50); dom.sync()
        best = 1e9
        for _ in range(3):
            meet.wait()
            t0 = time.perf_counter(); dom.strip_step_batch(steps); dom.sync(); meet.wait()
            best = min(best, (time.perf_counter() - t0) / steps * 1e6)
        result[r] = (got, best)
        dom.strip_comm_destroy(); dom.close()

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    [t.start() for t in ts]; [t.join() for t in ts]
    print("the same strip between two %d-row neighbours, protocol level %s : %6.1f us/iteration" % (thin, [g_ for g_, _ in result], max(b for _, b in result)), flush=True)
sys.exit(0)
