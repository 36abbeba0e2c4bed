"""Worker of tests/test_gpu_two_step.py::test_pairs_fuzz and tools/history/r05_two_step_soak.sh: random single-domain Godunov configurations
(shape, precision, workload, Manning array or not, friction, dx, Courant number, dynamic / fixed timestep, batch pattern with
downloads, partial uploads, target-time changes, update-timestep calls and checkpoints in between) run on the FAST engine; prints one
line per seed with a SHA-256 of everything observable.  Run twice (HP_TWO_STEP=0 / 1, the latter with HP_PAIR_EXACT=1: quirk Q3's stamps on every domain): the lines must be identical.
usage: two_step_fuzz_worker.py <first seed> <count>"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hipims-ocl_amd")]
import hipims_mi as hp  # noqa: E402
from hipims_mi import synthetic as syn  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    precision = "f64" if rng.random() < 0.7 else "f32"
    real = np.float64 if precision == "f64" else np.float32
    cols, rows = int(rng.integers(5, 700)), int(rng.integers(5, 500))
    if rng.random() < 0.25:
        cols, rows = int(rng.integers(900, 1500)), int(rng.integers(700, 1100))
    kind = rng.integers(0, 4)
    if kind == 0:
        st, bed, man = syn.s_dam(cols, rows, dtype=real)
    elif kind == 1:
        st, bed, man = syn.s_dam(cols, rows, dtype=real, wet_right=False)
    else:
        st, bed, man = syn.s_rough(cols, rows, dtype=real, manning=(None if rng.random() < 0.5 else 0.03), seed=int(rng.integers(0, 1000)))
    if rng.random() < 0.3:                                # a few null cells (DEM nodata and mask style)
        for _ in range(int(rng.integers(1, 6))):
            y, x = int(rng.integers(1, rows - 1)), int(rng.integers(1, cols - 1))
            st[y, x, 1] = -9999.0
            if rng.random() < 0.5:
                st[y, x, 0] = -9999.0; bed[y, x] = -9999.0
    kw = dict(dx=float(rng.choice([0.5, 1.0, 2.0, 3.0])), courant=float(rng.choice([0.3, 0.5])), friction=bool(rng.random() < 0.8))
    if rng.random() < 0.2:
        kw.update(dynamic_dt=False, dt_fixed=float(rng.choice([0.002, 0.01])))
    # (round 6) the exact mode now and then (STRICT pairs), and area boundaries -- uniform rain, a loss rate, coarse gridded rain, a mass
    # flux, alone or together, with series short enough to change slice inside a run (pairs with boundaries: godunov_march2 BDY)
    strict = rng.random() < 0.2
    with_bdy = rng.random() < 0.45
    if os.environ.get("FUZZ_STRICT_TUNER"):              # (tools/r06_tuner_soak.sh: every seed in the exact mode without boundaries -- the tuner's ground)
        strict, with_bdy = True, False
    if with_bdy and "dt_fixed" in kw:
        kw.pop("dynamic_dt"); kw.pop("dt_fixed")
    dom = hp.Domain(cols, rows, precision=precision, math_mode=hp.MATH_STRICT if strict else hp.MATH_FAST, **kw)
    dom.upload(st, bed, man)
    if with_bdy:
        if rng.random() < 0.7:
            dom.add_uniform(hp.UNIFORM_RAIN_INTENSITY, np.array([[0.0, float(rng.uniform(50, 900))], [0.7, float(rng.uniform(0, 300))], [1.4, 0.0], [2.1, 0.0]]), 0.7, 2.1)
        if rng.random() < 0.4:
            dom.add_uniform(hp.UNIFORM_LOSS_RATE, np.array([[0.0, float(rng.uniform(100, 2000))], [100.0, 0.0]]), 100.0, 100.0)
        if rng.random() < 0.5:
            res = float(rng.choice([64, 80, 200])) * kw["dx"]
            grids = rng.uniform(0.0, 600.0, (3, int(rows * kw["dx"] / res) + 2, int(cols * kw["dx"] / res) + 2))
            dom.add_gridded(hp.GRIDDED_RAIN_INTENSITY if rng.random() < 0.7 else hp.GRIDDED_MASS_FLUX, grids if rng.random() < 0.7 else grids * 1e-4, res, 0.0, 0.0, float(rng.choice([0.5, 3.0])))
    dom.set_target_time(float(rng.choice([1e9, 0.6, 2.0])))
    h = hashlib.sha256()
    blown = False                                    # a fixed timestep beyond the CFL limit ends in NaNs: where they sit and what their payloads are is not held to anything
    dump = os.environ.get("FUZZ_DUMP")               # (diagnosis: every op, the scalars and the state behind it)
    trace = []
    for _ in range(int(rng.integers(3, 9))):
        op = rng.integers(0, 10)
        if dump:
            trace.append((int(op), dom.read_scalars(), dom.download()))
        if op <= 4:
            dom.step_batch(int(rng.integers(1, 40)) + (16 if os.environ.get("FUZZ_STRICT_TUNER") else 0))   # (a sample wants a batch of fourteen iterations and more)
        elif op == 5:
            a = dom.download()
            blown = blown or not np.isfinite(a).all()
            h.update(a.tobytes())
        elif op == 6:
            dom.set_target_time(float(dom.read_scalars()["time"] + rng.choice([0.05, 0.5, 5.0])))
            dom.update_timestep()
        elif op == 7:
            dom.state_save(); dom.step_batch(int(rng.integers(1, 12))); dom.state_restore()
        elif op == 8:
            y0 = int(rng.integers(0, rows - 2))
            patch = dom.download(hp.ARRAY_STATE, y0, 2)
            patch[..., 0] += real(0.01)
            dom.upload_rows(patch, y0)
        else:
            dom.step_batch(int(rng.integers(1, 5)) * 2 + 1)
    sc = dom.read_scalars()
    final = dom.download()
    blown = blown or not np.isfinite(final).all()
    h.update(final.tobytes())
    h.update(repr((sc["time"], sc["timestep"], sc["batch_successful"], sc["batch_skipped"], sc["iterations"])).encode())
    counts = dom.launch_counts()
    if dump:
        trace.append((-1, sc, dom.download()))
        np.savez(os.path.join(dump, f"seed{seed}_{os.environ.get('HP_TWO_STEP', 'x')}.npz"), ops=np.array([t[0] for t in trace]),
                 times=np.array([t[1]["time"] for t in trace]), dts=np.array([t[1]["timestep"] for t in trace]),
                 its=np.array([t[1]["iterations"] for t in trace]), states=np.stack([t[2] for t in trace]))
    print(f"seed {seed} {precision}{' strict' if strict else ''}{' bdy' if with_bdy else ''} {cols}x{rows} kind {kind} iterations {sc['iterations']} t {sc['time']!r} {'non-finite-state-not-compared' if blown else h.hexdigest()}  # launches {counts[0]}")
    dom.close()
