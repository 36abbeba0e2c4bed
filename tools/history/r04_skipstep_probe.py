#!/usr/bin/env python3
"""Round 4: K1 as a COPY.  With the target time reached the timestep is 0 and the flux kernel runs its skip path (CLSchemeGodunov.clc:
:69-70 -- no faces, no update; every row is still loaded, stored and priced): the kernel's own memory structure without its arithmetic.
Against the same launch with arithmetic, and tools/membench's march copy, this says what the arithmetic costs the headline launch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "hipims-ocl_amd"))
os.environ["HIPIMS_MI_NO_TORCH"] = "1"
import numpy as np
import hipims_mi as hp
from hipims_mi import synthetic as syn
cols = rows = 4096
for prec in ("f64", "f32"):
    for scheme, name in ((hp.SCHEME_GODUNOV, "godunov"), (hp.SCHEME_MUSCL_HANCOCK, "muscl")):
        st, bed, man = syn.s_dam(cols, rows, dtype=np.float64 if prec == "f64" else np.float32)
        d = hp.Domain(cols, rows, scheme=scheme, precision=prec)
        d.upload(st, bed, man); d.set_target_time(1e9); d.step_batch(40); d.sync()
        def timed(n=400):
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter(); d.step_batch(n); d.sync(); best = min(best, (time.perf_counter() - t0) / n * 1e3)
            return best
        live = timed()
        sc = d.read_scalars()
        d.set_target_time(sc["time"])                      # reached: dt = 0 from here on
        d.step_batch(5); d.sync()
        skipped = timed()
        sc2 = d.read_scalars()
        print("%s %s 4096^2: with arithmetic %.4f ms, skip path (dt = 0) %.4f ms   (time %.6f -> %.6f, skipped iterations %s)" % (name, prec, live, skipped, sc["time"], sc2["time"], sc2.get("batch_skipped")))
        d.close()
