"""round 5: the speculative K1 FUSED TAIL kernel against the oracle after ONE real iteration (the rest of the batch is skipped: the
target time is reached), per lane / row / field.   usage: HP_STRICT_SPECULATE=1 python tools/r05fj_spec_debug2.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hipims-ocl_amd")]
import hipims_mi as hp  # noqa: E402
import oracle  # noqa: E402
from hipims_mi import synthetic as syn  # noqa: E402

cols, rows = 190, 101
for target in (0.004, 0.05, 0.15, 0.25):
    st, bed, man = syn.s_rough(cols, rows, manning=None, seed=31)
    ref = oracle.OracleSim(cols, rows, scheme=hp.SCHEME_GODUNOV, quirks=oracle.QUIRKS_REFERENCE)
    dom = hp.Domain(cols, rows, scheme=hp.SCHEME_GODUNOV, math_mode=hp.MATH_STRICT)
    for s in (ref, dom):
        s.upload(st, bed, man)
        s.add_uniform(hp.UNIFORM_RAIN_INTENSITY, np.array([[0.0, 90.0], [5.0, 30.0], [10.0, 0.0]]), 5.0, 10.0)
    dom.set_target_time(target); ref.set_target(target)
    ref.run(8); dom.step_batch(8)
    a, b = dom.download(), ref.download()
    sc, sr = dom.read_scalars(), ref.scalars()
    bad = np.argwhere(a != b)
    print(f"target {target}: successful {sc['batch_successful']} / {sr['batch_ok']}, t {sc['time']} / {sr['t']}: {len(bad)} entries differ, max {float(np.nanmax(np.abs(a - b))):.3e}")
    if len(bad):
        ys, xs, fs = bad[:, 0], bad[:, 1], bad[:, 2]
        lanes = (xs - 1) % 62 + 1
        print("   rows   ", dict(zip(*np.unique(ys, return_counts=True))))
        print("   lanes  ", dict(zip(*np.unique(lanes, return_counts=True))))
        print("   fields ", dict(zip(*np.unique(fs, return_counts=True))))
        for y, x, f in bad[:8]:
            print(f"   [{y},{x}] field {f}: engine {a[y, x, f]!r} oracle {b[y, x, f]!r}  source {st[y, x, f]!r} bed {bed[y, x]!r}")
    dom.close()
