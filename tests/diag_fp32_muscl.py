"""Step-by-step comparison of the fp32 MUSCL kernel (FAST and STRICT) with the oracle on the F6/F7 rough-bed case:
where and when do they part?  (diagnostic, GPU box)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tests/ (uses the oracle: test infrastructure)
sys.path[:0] = [ROOT, os.path.join(ROOT, "hipims-ocl_amd")]
import hipims_mi as hp
import oracle

g = np.load(os.path.join(ROOT, "tests/golden/f6_f7_trajectories_f32.npz"))
st, bed, man = g["rough_state"], g["rough_bed"], g["rough_manning"]
for mode, name in ((hp.MATH_FAST, "fast"), (hp.MATH_STRICT, "strict")):
    ref = oracle.OracleSim(64, 64, scheme=oracle.MUSCL, precision="f32", quirks=oracle.QUIRKS_REFERENCE & ~oracle.Q6_MUSCL_SERIAL)
    dom = hp.Domain(64, 64, scheme=hp.SCHEME_MUSCL_HANCOCK, precision="f32", math_mode=mode)
    for s in (ref, dom):
        s.upload(st, bed, man)
    dom.set_target_time(1e9); ref.set_target(1e9)
    reported = 0
    for i in range(200):
        ref.run(1); dom.step_batch(1)
        a, b = dom.download().astype(np.float64), ref.download().astype(np.float64)
        d = np.abs(a - b)
        m = d[..., 0].max()
        if (m > 1e-6 and reported < 4) or i in (0, 9, 49, 99, 199):
            y, x = np.unravel_index(d[..., 0].argmax(), d[..., 0].shape)
            print(f"{name} step {i+1}: max|dZ| {m:.3e} at (x={x}, y={y}) max|dq| {d[..., 2:].max():.3e} dt gpu/ref {dom.read_scalars()['timestep']:.9g} {ref.scalars()['dt']:.9g}")
            if m > 1e-6:
                reported += 1
                sl = np.s_[max(0, y - 1):y + 2, max(0, x - 1):x + 2]
                print("  gpu Z", a[sl][..., 0].round(7).tolist(), "\n  ref Z", b[sl][..., 0].round(7).tolist(), "\n  bed", bed[sl].round(7).tolist(),
                      "\n  gpu qx", a[sl][..., 2].tolist(), "\n  ref qx", b[sl][..., 2].tolist())
    dom.close()
