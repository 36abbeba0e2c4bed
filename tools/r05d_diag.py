#!/usr/bin/env python3
"""round 5 diagnostic: fp32 FAST (depth form) against the fp32 reference fixture AND against the fp64 truth on the walled dam break
-- is the 2e-3 m that separates FAST from the reference's fp32 run the reference's own fp32 error (its free-surface form at 9999.9 m walls)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hipims-ocl_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import hipims_mi as hp
from hipims_mi import synthetic as syn

def run(precision, mode, wet_right, steps=150, t_target=None):
    real = np.float64 if precision == "f64" else np.float32
    st, bed, man = syn.s_dam(96, 48, dtype=real, wet_right=wet_right)
    dom = hp.Domain(96, 48, precision=precision, math_mode=mode)
    dom.upload(st, bed, man)
    dom.set_target_time(1e9 if t_target is None else t_target)
    if t_target is None:
        dom.step_batch(steps)
    else:
        while t_target - dom.read_scalars()["time"] > 1e-9:
            dom.step_batch(10)
    out = dom.download().astype(np.float64); t = dom.read_scalars()["time"]
    dom.close()
    return np.maximum(0, out[..., 0] - bed.astype(np.float64)), t

g = np.load(os.path.join(ROOT, "tests/golden/f6_f7_trajectories_f32.npz"))
for wet_right, key in ((True, "dam_god"), (False, "damdry_god")):
    st, bed, man = syn.s_dam(96, 48, dtype=np.float32, wet_right=wet_right)
    ref32 = np.maximum(0, g[f"{key}_state150"][..., 0].astype(np.float64) - bed)
    t_ref = float(g[f"{key}_dt"].astype(np.float64).sum())
    truth, _ = run("f64", hp.MATH_STRICT, wet_right, t_target=t_ref)     # fp64 at the reference's elapsed time
    fast32, tf = run("f32", hp.MATH_FAST, wet_right)
    strict32, ts = run("f32", hp.MATH_STRICT, wet_right)
    r = lambda a, b: (float(np.sqrt(np.mean((a - b) ** 2))), float(np.abs(a - b).max()))
    print(key, "t_ref %.6f t_fast %.6f t_strict %.6f" % (t_ref, tf, ts))
    print("   fp32 FAST   vs fp32 reference fixture: rmse %.3e max %.3e" % r(fast32, ref32))
    print("   fp32 STRICT vs fp32 reference fixture: rmse %.3e max %.3e" % r(strict32, ref32))
    print("   fp32 FAST   vs fp64 truth            : rmse %.3e max %.3e" % r(fast32, truth))
    print("   fp32 STRICT vs fp64 truth            : rmse %.3e max %.3e" % r(strict32, truth))
    print("   fp32 reference fixture vs fp64 truth : rmse %.3e max %.3e" % r(ref32, truth))
