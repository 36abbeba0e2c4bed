"""Tuned vs basic kernel on S-RAIN fp32 (diagnostic, GPU box): where do they part?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hipims-ocl_amd")]
import hipims_mi as hp
from hipims_mi import synthetic as syn

for n, prec, its in ((1024, "f32", 14), (4096, "f32", 14), (8192, "f32", 3), (8192, "f32", 14), (8192, "f64", 14)):
    real = np.float32 if prec == "f32" else np.float64
    st, bed, man, rain = syn.s_rain(n, n, dx=2.0, dtype=real)
    outs = []
    for kernel in (hp.KERNEL_AUTO, hp.KERNEL_BASIC):
        d = hp.Domain(n, n, dx=2.0, precision=prec, kernel=kernel)
        d.upload(st, bed, man)
        d.add_gridded(hp.GRIDDED_RAIN_INTENSITY, rain["grids"], rain["resolution"], rain["off_x"], rain["off_y"], rain["interval"])
        d.set_target_time(1e9)
        d.step_batch(its)
        outs.append(d.download())
        d.close()
    diff = np.abs(outs[0].astype(np.float64) - outs[1].astype(np.float64)).max(axis=2)
    bad = diff > 1e-3
    print(f"n={n} {prec} its={its}: max diff {diff.max():.3e}, bad cells {bad.sum()}")
    if bad.any():
        ys, xs = np.nonzero(bad)
        print("  rows:", np.unique(ys)[:40], "... count", len(np.unique(ys)))
        print("  cols:", np.unique(xs)[:40], "... count", len(np.unique(xs)))
        y, x = ys[0], xs[0]
        print("  first bad cell", (x, y), "auto", outs[0][y, x], "basic", outs[1][y, x], "bed", bed[y, x], "initial", st[y, x])
