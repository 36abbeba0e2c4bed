"""Tuned (persistent march) vs basic kernel, step by step (diagnostic, GPU box): where do they part?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hipims-ocl_amd")]
import hipims_mi as hp
from hipims_mi import synthetic as syn

for cols, rows, prec, uniform in ((64, 64, "f64", False), (64, 64, "f64", True), (200, 300, "f64", True), (64, 64, "f32", True)):
    real = np.float64 if prec == "f64" else np.float32
    st, bed, man = syn.s_rough(cols, rows, dtype=real, manning=0.03 if uniform else None)
    doms = []
    for kernel in (hp.KERNEL_AUTO, hp.KERNEL_BASIC):
        d = hp.Domain(cols, rows, precision=prec, kernel=kernel)
        d.upload(st, bed, man)
        d.set_target_time(1e9)
        doms.append(d)
    for it in range(1, 4):
        outs = []
        for d in doms:
            d.step_batch(1)
            outs.append(d.download().astype(np.float64))
        diff = np.abs(outs[0] - outs[1]).max(axis=2)
        bad = diff > 1e-6
        print(f"{cols}x{rows} {prec} uniform_n={uniform} step {it}: max diff {diff.max():.3e}, bad cells {bad.sum()}, dt {doms[0].read_scalars()['timestep']:.6g} / {doms[1].read_scalars()['timestep']:.6g}")
        if bad.any():
            ys, xs = np.nonzero(bad)
            print("   rows:", np.unique(ys)[:30], "cols:", np.unique(xs)[:40])
            y, x = ys[0], xs[0]
            print("   first", (x, y), "auto", outs[0][y, x], "basic", outs[1][y, x])
            break
    for d in doms:
        d.close()
