#!/usr/bin/env python3
"""Companion of strong_probe.py for the transport that needs a neighbour: the 4096 x 514 strip of the 8-GPU split is cut
once more into TWO strips whose ranks are threads of this process sharing the one GPU (collective library = the tests'
in-process double, as in tests/strip_threads_worker.py).  The GPU does the same cell updates per iteration as the one
strip does in strong_probe.py's line (a); what comes on top is what the strip protocol costs per iteration with REAL
neighbours at both ends of a hand-over: level 2 = ghost rows and maxima written by the strips themselves (one flux launch +
one advance launch per rank and iteration), 1 = mailboxes + the double's send/receive, 0 = everything through the double
(whose all-reduce blocks the host: not a timing, listed for completeness).   usage: strong_probe_pair.py [cols rows [rain]]   (rain: fp32 S-RAIN instead of fp64 S-DAM)"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hipims-ocl_amd"))
os.environ["HIPIMS_MI_NO_TORCH"] = "1"
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
import hipims_mi as hp
from hipims_mi import strips, synthetic as syn

cols, rows = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 514)
rain_fp32 = len(sys.argv) > 3 and sys.argv[3] == "rain"        # config C5's shape: fp32, gridded rain on dry terrain (fused into the flux kernel)
world, steps = 2, int(os.environ.get("PROBE_STEPS", "1500"))
precision = "f32" if rain_fp32 else "f64"
if rain_fp32:
    st, bed, man, rn = syn.s_rain_rows(cols, rows, 0, rows, dx=2.0, dtype=np.float32)
else:
    st, bed, man = syn.s_dam(cols, rows)
    rn = None
def with_rain(dom):
    if rn is not None:
        dom.add_gridded(hp.GRIDDED_RAIN_INTENSITY, rn["grids"], rn["resolution"], rn["off_x"], rn["off_y"], rn["interval"])
lib = hp.load_library()
hp._check(lib, lib.hp_comm_load(os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so").encode()), "hp_comm_load")

single = hp.Domain(cols, rows, dx=2.0 if rain_fp32 else 1.0, precision=precision)
single.upload(st, bed, man); with_rain(single); single.set_target_time(1e9)
single.step_batch(50); single.sync()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); single.step_batch(steps); single.sync(); best = min(best, (time.perf_counter() - t0) / steps * 1e6)
print("strip %d x %d in one hp_step_batch                         : %6.1f us/iteration" % (cols, rows, best), flush=True)
single.close()

# Round 4: hp_domain_create searches the tiling of a launch that fits the chip in one round (bands x tile rows; the line above
# runs with it: 43.5 -> 39.0 us).  The search assumes the launch has the chip to ITSELF -- true for a rank on its own GPU, not
# for two half strips that share one: their two concurrent launches are best served by the 8-band tiling of round 3 (measured:
# 44.8 us with it, 47-48 us with searched tilings, profiles/r04f_pair_probe_bisect.txt).  The pair therefore keeps that tiling, and
# what this probe measures stays what it was built for: the PROTOCOL's cost on top of the same rows in one plain batch call
# (the second "strip alone" line, same tiling as the pair).
os.environ["HP_TILING_SEARCH"] = "0"
single = hp.Domain(cols, rows, dx=2.0 if rain_fp32 else 1.0, precision=precision)
single.upload(st, bed, man); with_rain(single); single.set_target_time(1e9)
single.step_batch(50); single.sync()
best = 1e9
for _ in range(3):
    t0 = time.perf_counter(); single.step_batch(steps); single.sync(); best = min(best, (time.perf_counter() - t0) / steps * 1e6)
print("the same with the 8-band tiling the pair below uses                : %6.1f us/iteration" % best, flush=True)
single.close()
for level in (2, 1):
    os.environ["HP_PEER_DIRECT"] = "1" if level == 2 else "0"
    uid = hp.comm_unique_id()
    parts = strips.partition(rows, world, 1)
    meet = threading.Barrier(world)
    tickets, result = [None] * world, [None] * world

    def rank_main(r):
        own_lo, own_hi, lo, hi = parts[r]
        dom = hp.Domain(cols, hi - lo, global_rows=rows, row_offset=lo, dx=2.0 if rain_fp32 else 1.0, precision=precision)
        dom.upload(st[lo:hi], bed[lo:hi], man[lo:hi]); with_rain(dom)
        dom.set_halo_overlap(level != 2)
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
    print("two strips of it, ranks as threads, protocol level %s              : %6.1f us/iteration" % ([g for g, _ in result], max(b for _, b in result)), flush=True)
sys.exit(0)
