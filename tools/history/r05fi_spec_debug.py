"""round 5: where does the speculative STRICT batch (K1 SPEC flavour with the launch's own tail block) part from the oracle?
usage: HP_STRICT_SPECULATE=1 python tools/r05fi_spec_debug.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hipims-ocl_amd")]
import hipims_mi as hp  # noqa: E402
import oracle  # noqa: E402
from hipims_mi import synthetic as syn  # noqa: E402

cols, rows = 190, 101
for rain in (1, 0):
    for target in (0.8, 1e9):
        for n in (8, 9, 10, 12, 16, 40):
            st, bed, man = syn.s_rough(cols, rows, manning=None, seed=31)
            ref = oracle.OracleSim(cols, rows, scheme=hp.SCHEME_GODUNOV, quirks=oracle.QUIRKS_REFERENCE)
            dom = hp.Domain(cols, rows, scheme=hp.SCHEME_GODUNOV, math_mode=hp.MATH_STRICT)
            for s in (ref, dom):
                s.upload(st, bed, man)
                if rain:
                    s.add_uniform(hp.UNIFORM_RAIN_INTENSITY, np.array([[0.0, 90.0], [5.0, 30.0], [10.0, 0.0]]), 5.0, 10.0)
            dom.set_target_time(target); ref.set_target(target)
            ref.run(n); dom.step_batch(n)
            a, b = dom.download(), ref.download()
            sc, sr = dom.read_scalars(), ref.scalars()
            bad = np.argwhere(a != b)
            print(f"rain {rain} target {target:g} n {n:3d}: cells differing {len(bad):6d}  max |diff| {float(np.nanmax(np.abs(a - b))):.3e}  "
                  f"t {sc['time']:.6f} / {sr['t']:.6f}  ok {sc['batch_successful']} / {sr['batch_ok']}  dt equal {sc['timestep'] == sr['dt']}"
                  + (f"  first {bad[:3].tolist()}" if len(bad) else ""))
            dom.close()
