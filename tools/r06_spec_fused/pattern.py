"""LAB_NOTES R6: where the speculative STRICT K1 with FUSED boundaries and its own tail block (round 5's deleted instantiation) parts
from the oracle -- which cells, after how many iterations.  usage: HIPIMS_MI_LIB=<variant .so> HP_STRICT_SPECULATE=1 python pattern.py [n]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hipims-ocl_amd")]
import hipims_mi as hp
import oracle
from hipims_mi import synthetic as syn
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cols, rows = 190, 101
st, bed, man = syn.s_rough(cols, rows, manning=None, seed=31)
ref = oracle.OracleSim(cols, rows)
dom = hp.Domain(cols, rows, math_mode=hp.MATH_STRICT)
for s in (ref, dom):
    s.upload(st, bed, man)
    s.add_uniform(hp.UNIFORM_RAIN_INTENSITY, np.array([[0.0, 90.0], [5.0, 30.0], [10.0, 0.0]]), 5.0, 10.0)
dom.set_target_time(0.8); ref.set_target(0.8)
ref.run(n); dom.step_batch(n)
a, b = dom.download(), ref.download()
bad = np.any(a != b, axis=2)
print(f"{n} iterations: {int(bad.sum())} of {bad.size} cells differ; fields: z {int((a[...,0]!=b[...,0]).sum())} zmax {int((a[...,1]!=b[...,1]).sum())} qx {int((a[...,2]!=b[...,2]).sum())} qy {int((a[...,3]!=b[...,3]).sum())}; max |diff| {float(np.nanmax(np.abs(a-b))):.3e}")
if bad.any():
    ys, xs = np.nonzero(bad)
    print("rows with mismatches:", sorted(set(ys.tolist()))[:40])
    print("columns with mismatches (count per column, first 64):", [(int(x), int((xs == x).sum())) for x in sorted(set(xs.tolist()))[:64]])
    wet = (b[..., 0] - bed) > 1e-10
    print("mismatching cells wet in the oracle:", int((bad & wet).sum()), "dry:", int((bad & ~wet).sum()))
    y, x = ys[0], xs[0]
    print("first:", (int(y), int(x)), "engine", a[y, x], "oracle", b[y, x], "bed", bed[y, x], "initial", st[y, x])
print("scalars", dom.read_scalars()["time"], ref.scalars()["t"])
