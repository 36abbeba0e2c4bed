#!/usr/bin/env python3
"""Per-iteration time of hp_step_batch on grids from the example's 342 x 195 up to 4096^2 (S-DAM, Godunov fp64): what the
launch's own tail block (LaunchTail: no separate advance launch for launches of <= 768 blocks; HP_LAUNCH_TAIL=0 switches it
off) saves where an iteration is bound by the hand-over between dependent kernels."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hipims-ocl_amd")); os.environ["HIPIMS_MI_NO_TORCH"] = "1"
import hipims_mi as hp
from hipims_mi import synthetic as syn
for cols, rows, steps in ((342, 195, 20000), (512, 512, 10000), (1024, 1024, 4000), (4096, 514, 2000), (2048, 2048, 1000), (4096, 4096, 400), (8192, 8192, 100)):
    st, bed, man = syn.s_dam(cols, rows)
    d = hp.Domain(cols, rows); d.upload(st, bed, man); d.set_target_time(1e9)
    d.step_batch(100); d.sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); d.step_batch(steps); d.sync(); best = min(best, (time.perf_counter() - t0) / steps * 1e6)
    print("%5d x %4d: %7.2f us/iteration" % (cols, rows, best), "(HP_LAUNCH_TAIL=0)" if os.environ.get("HP_LAUNCH_TAIL") == "0" else "(HP_TAIL_MAX_BLOCKS=%s)" % os.environ["HP_TAIL_MAX_BLOCKS"] if "HP_TAIL_MAX_BLOCKS" in os.environ else "", flush=True)
    d.close()
