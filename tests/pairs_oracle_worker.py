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
boundaries, dx = [], 1.0
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
elif case == "f9_uniform":                  # fixture F9 (the reference's kernels): uniform rain + a loss rate -- in PAIRS here, the exact flavour (a loss rate)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from conftest import load_golden
    g = load_golden("f9_rain_f64")
    rows, cols = g["bed"].shape
    n = 420
    st, bed, man = g["state"].astype(real), g["bed"].astype(real), g["manning"].astype(real)
    boundaries = [("uniform", hp.UNIFORM_RAIN_INTENSITY, g["series"], 3600.0, 10800.0), ("uniform", hp.UNIFORM_LOSS_RATE, g["loss"], 10800.0, 10800.0)]
elif case in ("rain1024", "rainloss1024", "bigdt1024", "bigdt700"):  # pairs WITH area boundaries (round 6: godunov_march2 BDY): pools on mostly dry rough
    # terrain, uniform rain + gridded rain on rain cells 128 model cells wide (+ a loss rate: the exact flavour).  bigdt: 40 m cells and 608
    # iterations -- the first 600 at the 0.1 s the reference holds the timestep to during the first minute (CLDynamicTimestep.clc:128-129),
    # then eight with timesteps of up to 7.8 s from the reduction: the hydrological gate opens on every iteration and every pair prices its
    # stored state with and without the next iteration's rain.  (No further: under such timesteps this film amplifies rounding tenfold
    # every five iterations -- FAST and the oracle are 1e-3 m and 2 % in elapsed time apart after a hundred, while the STRICT engine stays
    # the oracle bit for bit: tools/diag_bigdt.py)
    cols, rows, n, dx = (1024, 1024, 250, 2.0) if not case.startswith("bigdt") else (1024, 1024, 608 if case == "bigdt1024" else 700, 40.0)
    st, bed, man = syn.s_rough(cols, rows, dtype=np.float64, manning=None)
    st[..., 0] = np.maximum(bed, st[..., 0] - 0.6); st[..., 1] = st[..., 0]; st[..., 2:] = 0
    st[0] = st[-1] = 0; st[:, 0] = st[:, -1] = 0
    st, bed, man = st.astype(real), bed.astype(real), man.astype(real)
    grids = np.random.default_rng(5).uniform(0.0, 800.0, (4, rows // 128 + 2, cols // 128 + 2))
    boundaries = [("uniform", hp.UNIFORM_RAIN_INTENSITY, np.array([[0.0, 600.0], [30.0, 200.0], [60.0, 0.0], [90.0, 0.0]]), 30.0, 90.0),
                  ("gridded", hp.GRIDDED_RAIN_INTENSITY, grids, 128 * dx, 0.0, 0.0, 11.0)]
    if case != "rain1024":
        boundaries.append(("uniform", hp.UNIFORM_LOSS_RATE, np.array([[0.0, 900.0], [1000.0, 900.0]]), 1000.0, 1000.0))
else:
    raise SystemExit(case)

ref = oracle.OracleSim(cols, rows, dx=dx, precision=precision, threads=min(16, os.cpu_count() or 1))
ref.upload(st, bed, man)
dom = hp.Domain(cols, rows, dx=dx, precision=precision, math_mode=math_mode)
dom.upload(st, bed, man)
for b in boundaries:
    for sim in (ref, dom):
        (sim.add_uniform if b[0] == "uniform" else sim.add_gridded)(*b[1:])
if boundaries:
    assert dom.boundaries_fused()
ref.set_target(1e9)
ref.run(n)
dom.set_target_time(1e9)
for b in (batches or [n]):
    dom.step_batch(b)
got, sc, counts = dom.download(), dom.read_scalars(), dom.launch_counts()
sr = ref.scalars()
np.savez(out, got=got, want=ref.download(), bed=bed, t=sc["time"], dt=sc["timestep"], t_ref=sr["t"], dt_ref=sr["dt"],
         ok=sc["batch_successful"], ok_ref=sr["batch_ok"], iterations=sc["iterations"], launches=counts[0])
dom.close()
print(f"{case} {precision}: t = {sc['time']!r}, iterations {sc['iterations']}, flux launches {counts[0]}")
