#!/usr/bin/env python3
"""What the boundary applications cost the pair kernel on S-RAIN: the same film of water stepped (a) with the gridded rain on (every
launch LIVE when the timestep exceeds 1 s: the hydrological gate opens at every iteration) and (b) with the boundaries cleared (the pair
kernel without boundaries).  usage: live_cost_probe.py [f32|f64] [cols] [rows]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hipims-ocl_amd")]
import hipims_mi as hp  # noqa: E402
from hipims_mi import synthetic as syn  # noqa: E402

precision = sys.argv[1] if len(sys.argv) > 1 else "f32"
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
rows = int(sys.argv[3]) if len(sys.argv) > 3 else cols
real = np.float64 if precision == "f64" else np.float32
st, bed, man = np.empty((rows, cols, 4), real), np.empty((rows, cols), real), np.empty((rows, cols), real)
for r0 in range(0, rows, 1024):
    r1 = min(r0 + 1024, rows)
    a, b, c, rain = syn.s_rain_rows(cols, rows, r0, r1, dx=2.0, dtype=real)
    st[r0:r1], bed[r0:r1], man[r0:r1] = a, b, c
dom = hp.Domain(cols, rows, dx=2.0, precision=precision)
dom.upload(st, bed, man)
dom.add_gridded(hp.GRIDDED_RAIN_INTENSITY, rain["grids"], rain["resolution"], rain["off_x"], rain["off_y"], rain["interval"])
dom.set_target_time(1e9)
dom.step_batch(70)


def timed(n, label):
    dom.step_batch(10)
    dom.synchronize() if hasattr(dom, "synchronize") else dom.read_scalars()
    c0 = dom.launch_counts()
    t0 = time.perf_counter()
    dom.step_batch(n)
    sc = dom.read_scalars()
    el = time.perf_counter() - t0
    c1 = dom.launch_counts()
    print(f"{label:46s} {el / n * 1e3:.4f} ms/iteration, {c1[0] - c0[0]} flux launches for {n} iterations, dt = {sc['timestep']:.3f} s, t = {sc['time']:.1f}")


timed(100, f"{precision} {cols}x{rows} rain on")
timed(100, f"{precision} {cols}x{rows} rain on (again)")
dom.clear_boundaries()
timed(100, f"{precision} {cols}x{rows} boundaries cleared")
timed(100, f"{precision} {cols}x{rows} boundaries cleared (again)")
print("pair stats", dom.pair_stats())
dom.close()
