"""Seeded differential fuzzing: STRICT HIP engine vs the oracle, bit for bit, over random combinations of everything the hot
path has a switch for -- grid shape (tile and wavefront edges), scheme, precision, friction, uniform vs spatially varying
Manning n, quirks Q1 / Q9, the tuned and the cross-check kernel, dynamic vs fixed timestep, a sync point inside the run, uniform rain, uniform loss, coarse (fused)
and fine (stand-alone pass) gridded rain, cell boundaries with all four definitions, and how the iterations are cut into
batches.  Every case is a few thousand cells and a few hundred iterations: the whole file runs in about a minute.

Bit identity is only promised where the oracle and the engine run the same algorithm: MUSCL-Hancock in snapshot order
(quirk Q6), boundaries in the order added (Q7)."""
import os

import numpy as np
import pytest

import hipims_mi as hp
import oracle
from hipims_mi import synthetic as syn

pytestmark = pytest.mark.gpu

N_CASES = int(os.environ.get("HIPIMS_MI_FUZZ_CASES", "160"))          # (a soak run: HIPIMS_MI_FUZZ_CASES=2000)


def make_case(seed):
    rng = np.random.default_rng(1000 + seed)
    scheme = int(rng.choice([hp.SCHEME_GODUNOV, hp.SCHEME_GODUNOV, hp.SCHEME_MUSCL_HANCOCK, hp.SCHEME_INERTIAL]))
    precision = str(rng.choice(["f64", "f64", "f32"]))
    cols = int(rng.choice([7, 33, 62, 63, 64, 65, 124, 125, 190, 257]))
    rows = int(rng.choice([6, 17, 18, 19, 37, 64, 101]))
    dx = float(rng.choice([1.0, 2.0, 0.5]))
    real = np.float64 if precision == "f64" else np.float32
    st, bed, man = syn.s_rough(cols, rows, dtype=real, seed=int(rng.integers(1, 10 ** 6)), amplitude=float(rng.uniform(0.1, 0.8)),
                               pool_level=float(rng.uniform(-0.2, 0.5)), manning=None if rng.random() < 0.6 else 0.035)
    if scheme == hp.SCHEME_INERTIAL:                       # gentler initial discharges: the scheme is not positivity preserving
        st[..., 2:] *= 0.1
    if rng.random() < 0.25:                                # a few disabled cells (the domain's nulls)
        ys, xs = rng.integers(1, rows - 1, 3), rng.integers(1, cols - 1, 3)
        st[ys, xs, 0] = -9999.0; st[ys, xs, 1] = -9999.0; st[ys, xs, 2:] = 0
    quirks = oracle.QUIRKS_REFERENCE
    if rng.random() < 0.3:
        quirks &= ~hp.QUIRK_CFL_READS_PRIMARY
    if rng.random() < 0.3:
        quirks &= ~hp.QUIRK_BDY_TRUNCATED
    kw = dict(friction=bool(rng.random() < 0.8), dynamic_dt=bool(rng.random() < 0.85))
    fixed_dt = float(rng.choice([0.002, 0.004])) * dx       # well inside the CFL limit of these states: a fixed step that blows
                                                             # the run up (NaNs on both sides) has no bits left to compare
    bdy = []
    if scheme != hp.SCHEME_MUSCL_HANCOCK:                  # (MUSCL-Hancock never applies them, quirk Q8)
        if rng.random() < 0.5:
            bdy.append(("uniform", hp.UNIFORM_RAIN_INTENSITY, np.array([[0.0, rng.uniform(10, 200)], [5.0, rng.uniform(10, 200)], [10.0, 0.0]]), 5.0, 10.0))
        if rng.random() < 0.3:
            bdy.append(("uniform", hp.UNIFORM_LOSS_RATE, np.array([[0.0, rng.uniform(1, 30)], [100.0, 5.0]]), 100.0, 100.0))
        if rng.random() < 0.5:
            coarse = rng.random() < 0.6                    # >= 64 model cells per grid cell: rides in the flux kernel
            res = dx * (64 if coarse else 8)
            gr, gc = int(np.ceil(rows * dx / res)) + 1, int(np.ceil(cols * dx / res)) + 1
            bdy.append(("gridded", int(rng.choice([hp.GRIDDED_RAIN_INTENSITY, hp.GRIDDED_MASS_FLUX])),
                        rng.uniform(0.0, 300.0, (3, gr, gc)) * (1.0 if rng.random() < 0.5 else 1e-3), res, 4.0))
        if rng.random() < 0.3 and cols > 12 and rows > 8:
            cells = np.array([int(rng.integers(2, rows - 2)) * cols + int(rng.integers(2, cols - 2)) for _ in range(4)], np.uint64)
            series = np.array([[0.0, 0.3, 0.05, -0.02], [50.0, 0.5, 0.08, 0.03], [100.0, 0.2, 0.0, 0.0]])
            bdy.append(("cell", int(rng.integers(0, 3)), int(rng.integers(0, 4)), np.unique(cells), series, 50.0, 100.0))
        rng.shuffle(bdy)
    total = int(rng.integers(40, 260))
    cuts, left = [], total
    while left > 0:
        n = int(min(left, rng.choice([1, 1, 2, 3, 7, 20, 64, 1000])))
        cuts.append(n); left -= n
    target = float(rng.choice([1e9, 1e9, rng.uniform(0.05, 1.5)]))
    # (drawn last, so that the cases of earlier rounds keep their configurations) the one-work-item-per-cell cross-check kernel
    kernel = hp.KERNEL_BASIC if (scheme == hp.SCHEME_GODUNOV and rng.random() < 0.15) else hp.KERNEL_AUTO
    # host calls between batches (the reference's scheme does all of these between its batches: setTargetTime, forceTimestep,
    # tst_ResetCounters, tst_UpdateTimestep, and writeAll of a changed state -- CSchemeGodunov.cpp:1741-1811, :1178-1260):
    # (index of the batch after which it happens, what, argument)
    ops = []
    if rng.random() < 0.4 and len(cuts) > 1:
        for _ in range(int(rng.integers(1, 4))):
            what = str(rng.choice(["target", "force_dt", "reset", "update", "upload", "manning", "bed"] + (["rain"] if scheme != hp.SCHEME_MUSCL_HANCOCK else [])))
            arg = float(rng.uniform(0.02, 3.0)) if what == "target" else float(rng.choice([0.001, 0.0005]) * dx) if what == "force_dt" else float(rng.uniform(0.005, 0.05))
            ops.append((int(rng.integers(0, len(cuts) - 1)), what, arg))
    # (round 4, drawn after everything else) the predictor variant of MUSCL-Hancock -- the reference's default (neighbours' bed in
    # .y, quirk Q11) or mch_1st_cacheNone (their Zmax) -- and DEM-nodata style nulls (bed = -9999 as well) next to the mask-style ones
    if rng.random() < 0.3:
        quirks &= ~oracle.Q11_MUSCL_NB_Y_IS_BED
    nulls = np.argwhere(st[..., 1] == -9999.0)
    if len(nulls) and rng.random() < 0.5:
        bed = bed.copy()
        bed[nulls[0][0], nulls[0][1]] = -9999.0
    return dict(kernel=kernel, ops=ops, scheme=scheme, precision=precision, cols=cols, rows=rows, dx=dx, st=st, bed=bed, man=man, quirks=quirks, kw=kw,
                fixed_dt=fixed_dt, bdy=bdy, cuts=cuts, target=target)


def attach(sim, bdy):
    for b in bdy:
        if b[0] == "uniform":
            sim.add_uniform(b[1], b[2], b[3], b[4])
        elif b[0] == "gridded":
            sim.add_gridded(b[1], b[2], b[3], 0.0, 0.0, b[4])
        else:
            sim.add_cell(b[1], b[2], b[3], b[4], b[5], b[6])


@pytest.mark.parametrize("seed", range(N_CASES))
def test_strict_engine_equals_the_oracle_on_a_random_configuration(seed):
    c = make_case(seed)
    oq = c["quirks"] & ~(oracle.Q6_MUSCL_SERIAL if c["scheme"] == hp.SCHEME_MUSCL_HANCOCK else 0)
    ref = oracle.OracleSim(c["cols"], c["rows"], dx=c["dx"], scheme=c["scheme"], precision=c["precision"], quirks=oq,
                           friction=c["kw"]["friction"], dynamic_dt=c["kw"]["dynamic_dt"], fixed_dt=c["fixed_dt"],
                           dt_initial=c["fixed_dt"] if not c["kw"]["dynamic_dt"] else 0.001)
    dom = hp.Domain(c["cols"], c["rows"], dx=c["dx"], scheme=c["scheme"], precision=c["precision"], quirks=oracle.quirks_to_engine(c["quirks"]),
                    friction=c["kw"]["friction"], dynamic_dt=c["kw"]["dynamic_dt"], dt_fixed=c["fixed_dt"],
                    dt_initial=c["fixed_dt"] if not c["kw"]["dynamic_dt"] else 0.001, math_mode=hp.MATH_STRICT, kernel=c["kernel"])
    for s in (ref, dom):
        s.upload(c["st"], c["bed"], c["man"])
        attach(s, c["bdy"])
    dom.set_target_time(c["target"]); ref.set_target(c["target"])
    what = f"seed {seed}: scheme {c['scheme']} {c['precision']} {c['cols']}x{c['rows']} bdy {[b[0] for b in c['bdy']]} cuts {c['cuts'][:6]} target {c['target']:.3g}"
    done = 0
    for i, n in enumerate(c["cuts"]):
        ref.run(n); dom.step_batch(n)
        done += n
        for _, op, arg in (o for o in c["ops"] if o[0] == i):
            what += f" [{op} after batch {i}]"
            if op == "target":
                dom.set_target_time(arg); ref.set_target(arg)
            elif op == "force_dt":
                dom.force_timestep(arg); ref.force_dt(arg)
            elif op == "reset":
                dom.reset_counters(); ref.reset_counters()
            elif op == "update":
                dom.update_timestep(); ref.update_timestep()
            elif op == "manning":                              # a new roughness map (uniform -> varying, or the other way round)
                new_n = np.full((c["rows"], c["cols"]), 0.02 + arg, c["st"].dtype) if c["man"].std() > 0 else \
                    (0.02 + arg * np.random.default_rng(int(arg * 1e6)).random((c["rows"], c["cols"]))).astype(c["st"].dtype)
                dom.upload(manning=new_n); ref.upload(manning=new_n)
            elif op == "bed":                                  # the ground comes up a little under a patch (never above the water on it)
                new_b = c["bed"].copy()
                cur = ref.download()
                if not np.isfinite(cur[c["st"][..., 1] > -9000]).all():
                    pytest.skip("the oracle's own run is not finite: " + what)
                ys, xs = slice(c["rows"] // 2, c["rows"] // 2 + 2), slice(c["cols"] // 3, c["cols"] // 3 + 7)
                room = np.maximum(0.0, cur[ys, xs, 0].astype(np.float64) - new_b[ys, xs])
                new_b[ys, xs] += (np.minimum(room * 0.5, arg)).astype(new_b.dtype) * (new_b[ys, xs] < 9000)
                c["bed"] = new_b
                dom.upload(bed=new_b); ref.upload(bed=new_b)
                dom.update_timestep(); ref.update_timestep()
            elif op == "rain":                                 # one more boundary from here on (the fused set changes mid-run)
                series = np.array([[0.0, 40.0 + 1000.0 * arg], [20.0, 15.0], [40.0, 0.0]])
                dom.add_uniform(hp.UNIFORM_RAIN_INTENSITY, series, 20.0, 40.0); ref.add_uniform(hp.UNIFORM_RAIN_INTENSITY, series, 20.0, 40.0)
            else:                                              # a changed state written to both (raises the level of the wet cells of a patch)
                cur = ref.download()
                if not np.isfinite(cur[c["st"][..., 1] > -9000]).all():
                    pytest.skip("the oracle's own run is not finite: " + what)
                assert np.array_equal(dom.download(), cur), what + " before the upload"
                patch = cur[c["rows"] // 3:c["rows"] // 3 + 3, c["cols"] // 4:c["cols"] // 4 + 9]
                wet = (patch[..., 0] - c["bed"][c["rows"] // 3:c["rows"] // 3 + 3, c["cols"] // 4:c["cols"] // 4 + 9] > 1e-3) & (patch[..., 1] > -9000)
                patch[..., 0][wet] += np.asarray(arg, cur.dtype)
                patch[..., 1][wet] = np.maximum(patch[..., 1][wet], patch[..., 0][wet])
                dom.upload(cur); ref.upload(cur)
                dom.update_timestep(); ref.update_timestep()   # as the reference does after any write of the states
        if n >= 20 or done == sum(c["cuts"]):              # (a download after every single-iteration batch would dominate the run time)
            want = ref.download()
            if not np.isfinite(want[c["st"][..., 1] > -9000]).all():
                pytest.skip("the oracle's own run is not finite: " + what)
            assert np.array_equal(dom.download(), want), what + f" after {done} iterations"
    if not np.isfinite(ref.download()[c["st"][..., 1] > -9000]).all():
        pytest.skip("the oracle's own run is not finite: " + what)
    sc, sr = dom.read_scalars(), ref.scalars()
    assert sc["time"] == sr["t"] and sc["timestep"] == sr["dt"], what
    assert sc["batch_successful"] == sr["batch_ok"] and sc["batch_skipped"] == sr["batch_skipped"], what
    assert sc["time_hydrological"] == sr["t_hydro"], what
    assert np.isfinite(dom.download()[c["st"][..., 1] > -9000]).all(), what
    dom.close()


@pytest.mark.parametrize("seed", range(0, N_CASES, 4))
def test_fast_engine_does_not_depend_on_how_the_iterations_are_batched(seed):
    """The FAST flavour has no oracle to be bit-identical to, but it owes the same bits to itself however the run is cut:
    one batch (area boundaries fused into the flux kernel wherever they qualify), the random cuts, single-iteration
    batches (never fused), and the host-driven split iteration."""
    c = make_case(seed)
    total = sum(c["cuts"])
    outs = []
    for plan in ([total], c["cuts"], [1] * total, "split"):
        dom = hp.Domain(c["cols"], c["rows"], dx=c["dx"], scheme=c["scheme"], precision=c["precision"], quirks=oracle.quirks_to_engine(c["quirks"]),
                        friction=c["kw"]["friction"], dynamic_dt=c["kw"]["dynamic_dt"], dt_fixed=c["fixed_dt"],
                        dt_initial=c["fixed_dt"] if not c["kw"]["dynamic_dt"] else 0.001, math_mode=hp.MATH_FAST, kernel=c["kernel"])
        dom.upload(c["st"], c["bed"], c["man"])
        attach(dom, c["bdy"])
        dom.set_target_time(c["target"])
        if plan == "split":                                    # the host-driven iteration (hp_step_begin / hp_step_end), with and
            dom.set_halo_overlap(seed % 8 == 0)                # without the halo part on its own stream
            for _ in range(total):
                dom.step_begin(); dom.step_end()
        else:
            for n in plan:
                dom.step_batch(n)
        outs.append((dom.download(), dom.read_scalars()))
        dom.close()
    for out, sc in outs[1:]:
        assert np.array_equal(out, outs[0][0], equal_nan=True), seed
        assert sc["time"] == outs[0][1]["time"] and sc["timestep"] == outs[0][1]["timestep"]


@pytest.mark.parametrize("seed", range(0, N_CASES, 4))
def test_a_checkpoint_taken_anywhere_replays_to_the_same_bits(seed):
    """hp_state_save / hp_state_restore (CSchemeGodunov::saveCurrentState / rollbackSimulation, CSchemeGodunov.cpp:1720-1736,
    :1474-1518) on a random configuration: run part of the way, save, wander off for a random number of iterations (with a
    different target time, so that the scalars really move), restore, finish -- and compare with the run that never left,
    state and scalars, in the case's own arithmetic (every other case STRICT)."""
    c = make_case(seed)
    rng = np.random.default_rng(5000 + seed)
    total = sum(c["cuts"])
    at = int(rng.integers(1, total))
    math_mode = hp.MATH_STRICT if seed % 8 == 0 else hp.MATH_FAST

    def fresh():
        dom = hp.Domain(c["cols"], c["rows"], dx=c["dx"], scheme=c["scheme"], precision=c["precision"], quirks=oracle.quirks_to_engine(c["quirks"]),
                        friction=c["kw"]["friction"], dynamic_dt=c["kw"]["dynamic_dt"], dt_fixed=c["fixed_dt"],
                        dt_initial=c["fixed_dt"] if not c["kw"]["dynamic_dt"] else 0.001, math_mode=math_mode, kernel=c["kernel"])
        dom.upload(c["st"], c["bed"], c["man"])
        attach(dom, c["bdy"])
        dom.set_target_time(c["target"])
        return dom

    straight = fresh()
    straight.step_batch(at); straight.step_batch(total - at)
    want, want_sc = straight.download(), straight.read_scalars()
    straight.close()
    dom = fresh()
    dom.step_batch(at)
    dom.state_save()
    dom.set_target_time(c["target"] * 0.5 + 0.01)
    for n in (int(rng.integers(1, 9)), int(rng.integers(1, 40))):
        dom.step_batch(n)
    dom.state_restore()
    dom.set_target_time(c["target"])
    dom.step_batch(total - at)
    got, got_sc = dom.download(), dom.read_scalars()
    dom.close()
    assert np.array_equal(got, want, equal_nan=True), (seed, at, total)
    for k in ("time", "timestep", "time_hydrological", "batch_timesteps"):
        assert got_sc[k] == want_sc[k] or (np.isnan(got_sc[k]) and np.isnan(want_sc[k])), (seed, k, got_sc[k], want_sc[k])
