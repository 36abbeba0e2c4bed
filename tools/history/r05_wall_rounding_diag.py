#!/usr/bin/env python3
"""round 5 diagnostic: where does FAST first leave the oracle on the emerging-bed dam break (and the fp32 walled dam break)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "hipims-ocl_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import hipims_mi as hp
import oracle
from hipims_mi import synthetic as syn
np.set_printoptions(precision=12, linewidth=200)

def trace(name, cols, rows, st, bed, man, precision="f64", tol=1e-9, steps=60, **kw):
    ref = oracle.OracleSim(cols, rows, precision=precision, **{k: v for k, v in kw.items() if k in ("dx", "friction")})
    dom = hp.Domain(cols, rows, precision=precision, math_mode=hp.MATH_FAST, **kw)
    for s in (ref, dom):
        s.upload(st, bed, man)
    dom.set_target_time(1e9); ref.set_target(1e9)
    for it in range(steps):
        dom.step_batch(1); ref.run(1)
        a, b = dom.download().astype(np.float64), ref.download().astype(np.float64)
        d = np.abs(a - b)
        d[~np.isfinite(d)] = 1e30
        if d.max() > tol:
            y, x, c = np.unravel_index(np.argmax(d), d.shape)
            bad = np.argwhere(d.max(axis=2) > tol)
            print(f"{name}: iteration {it + 1}: {len(bad)} cells differ by more than {tol}; worst {d.max():.3e} at (y={y}, x={x}, comp {c}); dt gpu {dom.read_scalars()['timestep']!r} ref {ref.scalars()['dt']!r}")
            print("  differing cells (y, x):", bad[:12].tolist())
            for yy, xx in bad[:3]:
                print(f"  cell ({yy},{xx}) gpu {a[yy, xx]} ref {b[yy, xx]} bed {bed[yy, xx]}")
                for dy, dx_ in ((0, -1), (0, 1), (-1, 0), (1, 0)):
                    y2, x2 = yy + dy, xx + dx_
                    if 0 <= y2 < rows and 0 <= x2 < cols:
                        print(f"     nb ({y2},{x2}) ref {b[y2, x2]} bed {bed[y2, x2]}")
            return
    print(f"{name}: no difference above {tol} in {steps} iterations; t gpu {dom.read_scalars()['time']!r} ref {ref.scalars()['t']!r}")

st, bed, xs, front = syn.emerging_bed_dam_break()
trace("emerging bed f64", st.shape[1], st.shape[0], st, bed, np.zeros(bed.shape), dx=0.05, friction=False, steps=400)
st, bed, man = syn.s_dam(96, 48, dtype=np.float32)
trace("walled dam f32", 96, 48, st, bed, man, precision="f32", tol=2e-5, steps=150)
st, bed, man = syn.s_dam(96, 48)
trace("walled dam f64", 96, 48, st, bed, man, tol=1e-10, steps=150)
