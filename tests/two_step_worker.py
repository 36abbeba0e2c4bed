"""Worker of tests/test_gpu_two_step.py (HP_TWO_STEP is read once per process): runs a scenario on the FAST engine and writes the
final state, the time-control scalars and the number of iteration pairs the engine ran to an .npz.
usage: two_step_worker.py <scenario> <precision> <out.npz>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hipims-ocl_amd")]
import hipims_mi as hp  # noqa: E402
from hipims_mi import synthetic as syn  # noqa: E402

scenario, precision, out = sys.argv[1:4]
real = np.float64 if precision == "f64" else np.float32
kw = {}
math_mode = hp.MATH_FAST
if scenario.startswith("strict_"):          # the same scenarios in the exact mode (round 6: STRICT pairs)
    scenario, math_mode = scenario[len("strict_"):], hp.MATH_STRICT
if scenario == "dam":                       # all wet, walls: the benchmark's shape
    cols, rows = 1030, 700
    st, bed, man = syn.s_dam(cols, rows, dtype=real)
    plan = [("run", 40), ("run", 7), ("download",), ("run", 64), ("run", 1), ("run", 30)]
elif scenario == "rough":                   # wet/dry rough terrain with a Manning array: untouched cells, stopping conditions, friction
    cols, rows = 500, 333
    st, bed, man = syn.s_rough(cols, rows, dtype=real, manning=None)
    plan = [("run", 50), ("run", 33), ("download",), ("run", 100), ("run", 2), ("run", 61)]
elif scenario == "damdry":                  # dam break onto a dry bed + a sync point inside the run (clipped and suspended iterations)
    cols, rows = 600, 120
    st, bed, man = syn.s_dam(cols, rows, dtype=real, wet_right=False)
    plan = [("target", 1.0), ("run", 64), ("run", 64), ("target", 2.5), ("update",), ("run", 90), ("save",), ("run", 20), ("restore",), ("run", 20)]
elif scenario == "fixed":                   # fixed timestep (no reduction at all)
    cols, rows = 400, 300
    st, bed, man = syn.s_rough(cols, rows, dtype=real, manning=0.03)
    kw = dict(dynamic_dt=False, dt_fixed=0.01)
    plan = [("run", 41), ("run", 40)]
elif scenario in ("tune_dam", "tune_rough"):   # big enough for pairs by default: the exact mode's tuner samples a pair against two single iterations
    cols, rows = 2048, 1500
    st, bed, man = (syn.s_dam(cols, rows, dtype=real) if scenario == "tune_dam" else syn.s_rough(cols, rows, dtype=real, manning=0.03))
    plan = [("run", 30), ("run", 21), ("run", 40), ("download",), ("run", 30)]     # (a sample takes a batch of twelve iterations or more)
elif scenario == "c1":                       # config C1, the reference's example model (rain 70 mm/h + drainage 12 mm/h on its own DEM), through the front end
    import tempfile
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from hipims_mi import frontend
    from model_dir import make_newcastle
    import pathlib
    cfg = frontend.parse_configuration(make_newcastle(pathlib.Path(tempfile.mkdtemp(prefix="c1_"))))
    st, bed, man, res = frontend.build_domain(cfg)
    cols, rows = 342, 195
    st, bed, man = st.astype(real), bed.astype(real), man.astype(real)
    kw = dict(dx=res, t_end=cfg.duration)
    plan = [("c1bdy",), ("run", 512), ("run", 512), ("download",), ("run", 511), ("run", 465)]
elif scenario == "rainlater":               # pairs first, then a rain boundary arrives: the first single iteration behind the pairs is K1 with FUSED
    cols, rows = 500, 333                   # boundaries AND the FILL flag (the buffer it writes is two states old); dry land keeps untouched cells
    st, bed, man = syn.s_rough(cols, rows, dtype=real, manning=0.03)
    plan = [("run", 24), ("rain",), ("run", 9), ("run", 30), ("download",), ("run", 5)]
elif scenario in ("rain", "rainloss", "gridrain", "bigdt", "drying", "massflux"):
    # area boundaries under iteration pairs (round 6: godunov_march2 BDY).  Mostly dry rough terrain with pools: rain wets it, the loss
    # rate dries it again (whole regions at once: the reference then leaves their cells untouched, quirk Q3, with STALE values in the
    # other buffer -- what the pair kernel's stamps are for); batches of odd and even length, downloads, a sync point, a checkpoint
    cols, rows = (520, 390) if scenario != "bigdt" else (300, 260)
    st, bed, man = syn.s_rough(cols, rows, dtype=np.float64, manning=None)
    st[..., 0] = np.maximum(bed, st[..., 0] - (0.6 if scenario != "drying" else 0.35)); st[..., 1] = st[..., 0]
    if scenario != "drying":
        st[..., 2:] = 0
    else:                                   # thin moving films over the bumps: fronts that dry out within a step or two
        st[..., 2:] = np.where((st[..., 0] - bed)[..., None] > 1e-3, st[..., 2:] * 0.2, 0.0)
    st[0] = st[-1] = 0; st[:, 0] = st[:, -1] = 0
    st, bed, man = st.astype(real), bed.astype(real), man.astype(real)
    kw = dict(dx=2.0) if scenario != "bigdt" else dict(dx=40.0)       # dx = 40 m: timesteps of a second and more, the hydrological gate opens on EVERY iteration
    plan = [("bdy",), ("run", 40), ("run", 7), ("download",), ("run", 64), ("run", 1), ("target", None), ("run", 90), ("update",), ("run", 33),
            ("save",), ("run", 20), ("restore",), ("run", 20), ("download",), ("run", 31)]
    if scenario == "bigdt":                 # (past the first minute, where tst_Advance_Normal holds the timestep at 0.1 s: CLDynamicTimestep.clc:128-129)
        plan = [("settime", 70.0)] + plan
else:
    raise SystemExit(scenario)
dom = hp.Domain(cols, rows, precision=precision, math_mode=math_mode, **kw)
dom.upload(st, bed, man)
dom.set_target_time(1e9)
for step in plan:
    if step[0] == "run":
        dom.step_batch(step[1])
    elif step[0] == "download":
        dom.download()
    elif step[0] == "target":
        # (None: a sync point a little ahead of wherever the run is -- clipped and suspended iterations inside the next batch)
        dom.set_target_time(step[1] if step[1] is not None else dom.read_scalars()["time"] * 1.02 + 0.5)
    elif step[0] == "c1bdy":
        frontend.attach_boundaries(cfg, dom, cols)
        assert dom.boundaries_fused()
    elif step[0] == "settime":
        dom.set_time(step[1])
    elif step[0] == "bdy":
        rng = np.random.default_rng(3)
        if scenario in ("rain", "rainloss", "bigdt", "drying"):
            span = 300.0 if scenario == "bigdt" else 30.0
            dom.add_uniform(hp.UNIFORM_RAIN_INTENSITY, np.array([[0.0, 600.0 if scenario != "drying" else 0.0], [span, 200.0], [2 * span, 0.0], [3 * span, 0.0]]), span, 3 * span)
        if scenario in ("rainloss", "bigdt", "drying"):
            dom.add_uniform(hp.UNIFORM_LOSS_RATE, np.array([[0.0, 900.0], [1000.0, 900.0]]), 1000.0, 1000.0)
        if scenario in ("gridrain", "bigdt"):
            res = 64 * kw["dx"]
            grids = rng.uniform(0.0, 800.0, (4, rows // 64 + 2, cols // 64 + 2))
            dom.add_gridded(hp.GRIDDED_RAIN_INTENSITY, grids, res, 0.0, 0.0, 11.0)
        if scenario == "massflux":
            res = 80 * kw["dx"]
            grids = rng.uniform(0.0, 0.05, (3, rows // 80 + 2, cols // 80 + 2))
            dom.add_gridded(hp.GRIDDED_MASS_FLUX, grids, res, 0.0, 0.0, 17.0)
            dom.add_uniform(hp.UNIFORM_RAIN_INTENSITY, np.array([[0.0, 300.0], [50.0, 0.0]]), 50.0, 50.0)
        assert dom.boundaries_fused()
    elif step[0] == "update":
        dom.update_timestep()
    elif step[0] == "save":
        dom.state_save()
    elif step[0] == "restore":
        dom.state_restore()
    elif step[0] == "rain":
        dom.add_uniform(hp.UNIFORM_RAIN_INTENSITY, np.array([[0.0, 90.0], [5.0, 30.0], [10.0, 0.0]]), 5.0, 10.0)
final = dom.download()
sc = dom.read_scalars()
counts = dom.launch_counts()
ps = dom.pair_stats() or dict(pairs=0, cold_starts=0, stamped_last=0, stamped_ever=0)
np.savez(out, state=final, t=sc["time"], dt=sc["timestep"], ok=sc["batch_successful"], skipped=sc["batch_skipped"],
         iterations=sc["iterations"], launches=counts[0], pairs=ps["pairs"], cold_starts=ps["cold_starts"], stamped_ever=ps["stamped_ever"],
         tune_samples=ps.get("tune_samples", 0), prefers_pairs=ps.get("prefers_pairs", True), pair_over_single=ps.get("pair_over_single", 0.0),
         stale_used=ps.get("stale_used", 0))
dom.close()
print(f"{scenario} {precision}: t = {sc['time']!r}, iterations {sc['iterations']}, flux launches {counts[0]}")
