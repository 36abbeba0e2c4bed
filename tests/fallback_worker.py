"""Worker of tests/test_gpu_fallback_paths.py: the switched-off variants of the engine's fused paths, STRICT against the oracle.
The switches (HP_LAUNCH_TAIL, HP_FUSE_BDY) are read once per process, hence a process of their own.
usage: fallback_worker.py   (environment decides what is switched off)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hipims-ocl_amd")]
import hipims_mi as hp  # noqa: E402
import oracle  # noqa: E402
from hipims_mi import synthetic as syn  # noqa: E402

cols, rows = 190, 101
st, bed, man = syn.s_rough(cols, rows, manning=None, seed=77)
rng = np.random.default_rng(5)
grids = rng.uniform(0.0, 200.0, (3, 3, 4))
series = np.array([[0.0, 120.0], [5.0, 60.0], [10.0, 0.0]])
for scheme in (hp.SCHEME_GODUNOV, hp.SCHEME_MUSCL_HANCOCK, hp.SCHEME_INERTIAL):
    stx = st.copy()
    if scheme == hp.SCHEME_INERTIAL:
        stx[..., 2:] *= 0.1
    for precision in ("f64", "f32"):
        real = np.float64 if precision == "f64" else np.float32
        quirks = oracle.QUIRKS_REFERENCE & ~(oracle.Q6_MUSCL_SERIAL if scheme == hp.SCHEME_MUSCL_HANCOCK else 0)
        ref = oracle.OracleSim(cols, rows, scheme=scheme, precision=precision, quirks=quirks)
        dom = hp.Domain(cols, rows, scheme=scheme, precision=precision, math_mode=hp.MATH_STRICT)
        for s in (ref, dom):
            s.upload(stx.astype(real), bed.astype(real), man.astype(real))
            if scheme != hp.SCHEME_MUSCL_HANCOCK:
                s.add_uniform(hp.UNIFORM_RAIN_INTENSITY, series, 5.0, 10.0)
                s.add_gridded(hp.GRIDDED_RAIN_INTENSITY, grids, 64.0, 0.0, 0.0, 4.0)     # coarse: fusable
        dom.set_target_time(0.9); ref.set_target(0.9)           # a sync point inside the run
        fused = dom.boundaries_fused()
        for n in (40, 1, 64, 25):
            ref.run(n); dom.step_batch(n)
            assert np.array_equal(dom.download(), ref.download()), (scheme, precision, n)
        sc, sr = dom.read_scalars(), ref.scalars()
        assert sc["time"] == sr["t"] and sc["timestep"] == sr["dt"] and sc["batch_skipped"] == sr["batch_skipped"], (sc, sr)
        if scheme == hp.SCHEME_GODUNOV:
            print(f"fused={int(bool(fused))}")
        dom.close()
print("fallback paths bit-identical")
