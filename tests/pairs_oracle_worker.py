"""Worker of tests/test_gpu_pairs_oracle.py (HP_TWO_STEP is read once per process, so every case is its own process): runs one
case on the FAST engine -- with whatever HP_TWO_STEP the caller put into the environment -- and the same case on the oracle
(oracle.OracleSim: the CPU restatement, test infrastructure), and writes both final states, the time-control scalars and the
engine's launch counts to an .npz.  The assertions live in the test.
usage: pairs_oracle_worker.py <case> <precision> <out.npz>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hipims-ocl_amd")]
import hipims_mi as hp  # noqa: E402
import oracle  # noqa: E402
from hipims_mi import synthetic as syn  # noqa: E402

case, precision, out = sys.argv[1:4]
real = np.float64 if precision == "f64" else np.float32
batches = None
math_mode = hp.MATH_FAST
if case.startswith("strict_"):              # the exact mode's pairs: held to the oracle BIT FOR BIT by the test
    case, math_mode = case[len("strict_"):], hp.MATH_STRICT
if case == "f6_rough":                      # fixture F6's rough bed (god_q_state200: the reference's own kernels)
    cols, rows, n = 64, 64, 200
    st, bed, man = syn.s_rough(cols, rows, dtype=real, manning=None)
elif case in ("f6_dam", "f6_damdry"):       # fixture F6's dam breaks (dam/damdry_god_state150)
    cols, rows, n = 96, 48, 150
    st, bed, man = syn.s_dam(cols, rows, dtype=real, wet_right=case == "f6_dam")
elif case == "rough1024":                   # S-ROUGH, a million cells, Manning array, friction on: every tile on the general path
    cols, rows, n = 1024, 1024, 250
    st, bed, man = syn.s_rough(cols, rows, dtype=real, manning=None)
elif case == "damdry1024":                  # S-DAM-DRY: a wet/dry front through still water and dry land
    cols, rows, n = 1024, 1024, 250
    st, bed, man = syn.s_dam(cols, rows, dtype=real, wet_right=False)
elif case == "default1500":                 # above the default's threshold (1.5 M cells, one round of blocks): whatever HP_TWO_STEP-less selection does
    cols, rows, n = 1500, 1100, 120
    st, bed, man = syn.s_rough(cols, rows, dtype=real, manning=None)
    batches = [61, 59]                      # an odd batch: a single iteration between the pairs, K1's FILL flag behind them
else:
    raise SystemExit(case)

ref = oracle.OracleSim(cols, rows, precision=precision, threads=min(16, os.cpu_count() or 1))
ref.upload(st, bed, man)
ref.set_target(1e9)
ref.run(n)
dom = hp.Domain(cols, rows, precision=precision, math_mode=math_mode)
dom.upload(st, bed, man)
dom.set_target_time(1e9)
for b in (batches or [n]):
    dom.step_batch(b)
got, sc, counts = dom.download(), dom.read_scalars(), dom.launch_counts()
sr = ref.scalars()
np.savez(out, got=got, want=ref.download(), bed=bed, t=sc["time"], dt=sc["timestep"], t_ref=sr["t"], dt_ref=sr["dt"],
         ok=sc["batch_successful"], ok_ref=sr["batch_ok"], iterations=sc["iterations"], launches=counts[0])
dom.close()
print(f"{case} {precision}: t = {sc['time']!r}, iterations {sc['iterations']}, flux launches {counts[0]}")
