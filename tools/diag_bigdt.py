#!/usr/bin/env python3
"""bigdt1024 (tests/pairs_oracle_worker.py): engine FAST / STRICT against the oracle, state and time control compared every few iterations
past the first minute -- where do they part?  usage: diag_bigdt.py [cols]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hipims-ocl_amd")]
import hipims_mi as hp  # noqa: E402
import oracle  # noqa: E402
from hipims_mi import synthetic as syn  # noqa: E402

cols = rows = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dx = 40.0
st, bed, man = syn.s_rough(cols, rows, dtype=np.float64, manning=None)
st[..., 0] = np.maximum(bed, st[..., 0] - 0.6); st[..., 1] = st[..., 0]; st[..., 2:] = 0
st[0] = st[-1] = 0; st[:, 0] = st[:, -1] = 0
grids = np.random.default_rng(5).uniform(0.0, 800.0, (4, rows // 128 + 2, cols // 128 + 2))
boundaries = [("uniform", hp.UNIFORM_RAIN_INTENSITY, np.array([[0.0, 600.0], [30.0, 200.0], [60.0, 0.0], [90.0, 0.0]]), 30.0, 90.0),
              ("gridded", hp.GRIDDED_RAIN_INTENSITY, grids, 128 * dx, 0.0, 0.0, 11.0),
              ("uniform", hp.UNIFORM_LOSS_RATE, np.array([[0.0, 900.0], [1000.0, 900.0]]), 1000.0, 1000.0)]
which = os.environ.get("BDY", "012")
boundaries = [b for i, b in enumerate(boundaries) if str(i) in which]
ref = oracle.OracleSim(cols, rows, dx=dx, threads=16)
ref.upload(st, bed, man)
sims = {"fast": hp.Domain(cols, rows, dx=dx), "strict": hp.Domain(cols, rows, dx=dx, math_mode=hp.MATH_STRICT)}
for d in sims.values():
    d.upload(st, bed, man)
for b in boundaries:
    for sim in [ref] + list(sims.values()):
        (sim.add_uniform if b[0] == "uniform" else sim.add_gridded)(*b[1:])
ref.set_target(1e9)
for d in sims.values():
    d.set_target_time(1e9)
done = 0
for n in [590, 8, 2, 1, 1, 1, 1, 1, 2, 5, 10, 20, 50]:
    ref.run(n)
    done += n
    sr = ref.scalars()
    want = ref.download()
    line = f"it {done:4d} oracle t {sr['t']:.9f} dt {sr['dt']:.9f} |"
    for name, d in sims.items():
        d.step_batch(n)
        sc = d.read_scalars()
        got = d.download()
        dz = np.abs(np.maximum(0, got[..., 0] - bed) - np.maximum(0, want[..., 0] - bed))
        line += f" {name}: t-t_ref {sc['time'] - sr['t']:+.3e} dt-dt_ref {sc['timestep'] - sr['dt']:+.3e} max|dh| {dz.max():.3e} at {np.unravel_index(dz.argmax(), dz.shape)} |"
    print(line)
