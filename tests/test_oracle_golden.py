"""The C oracle against the golden fixtures produced by the reference's own kernels (tests/golden/generate.py).

Bit-exact for the strict fixtures (both built with -ffp-contract=off); the "as shipped" (-cl-mad-enable)
fixtures bracket the reference's own compiler-dependent spread.
"""
import numpy as np
import pytest

import oracle
from conftest import load_golden

PRECISIONS = ["f64", "f32"]


def same(a, b):
    """Bitwise equality up to the sign of zero / NaN payloads."""
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_reconstruct_and_hllc(precision):
    g = load_golden(f"f1_f2_reconstruct_hllc_{precision}")
    f = oracle.OracleFunctions(precision, very_small=float(g["very_small"]))
    n = g["sL"].shape[0]
    regions = set()
    for d in range(4):
        for i in range(n):
            oL, oR, stop = f.reconstruct(d, g["sL"][i], g["bL"][i], g["sR"][i], g["bR"][i])
            assert same(oL, g["rec_L"][d, i]) and same(oR, g["rec_R"][d, i]) and stop == g["rec_stop"][d, i], (d, i)
            F = f.hllc(d, oL, oR)
            assert same(F, g["flux"][d, i]), (d, i, F, g["flux"][d, i])
            regions.add((oL[1] < 1e-10, oR[1] < 1e-10))
    assert len(regions) == 4            # wet-wet, dry-wet, wet-dry and dry-dry all exercised
    assert (g["rec_stop"] > 0).any()


@pytest.mark.parametrize("precision", PRECISIONS)
def test_friction(precision):
    g = load_golden(f"f3_friction_{precision}")
    f = oracle.OracleFunctions(precision)
    for i in range(g["state"].shape[0]):
        out = f.friction(g["state"][i], g["bed"][i], g["manning"][i], g["dt"][i])
        assert same(out, g["out"][i]), i
    assert (g["out"][:, 2:] != g["state"][:, 2:]).any()


@pytest.mark.parametrize("precision", PRECISIONS)
def test_minmod_limiter(precision):
    g = load_golden(f"f4_limiter_{precision}")
    f = oracle.OracleFunctions(precision)
    for i in range(g["sL"].shape[0]):
        out = f.limiter(g["sL"][i], g["sC"][i], g["sR"][i], g["bL"][i], g["bC"][i], g["bR"][i])
        assert same(out, g["out"][i]), i
    assert (g["out"] != 0).any() and (g["out"] == 0).all(axis=1).any()


@pytest.mark.parametrize("precision", PRECISIONS)
def test_muscl_predictor_and_reconstruct2(precision):
    g = load_golden(f"f5_muscl_predict_{precision}")
    f = oracle.OracleFunctions(precision)
    m = g["states"].shape[0]
    faces = np.zeros_like(g["faces"])
    for i in range(m):
        faces[i], first = f.mch_1st(g["dt"][i], g["states"][i].reshape(-1), g["beds"][i])
        assert first == g["first"][i]
    assert same(faces, g["faces"])
    assert 0 < g["first"].sum() < m
    perm = g["perm"]
    for d in range(4):
        for i in range(m):
            j = perm[i]
            eL, eR = (faces[i, d], faces[j, (d + 2) % 4]) if d < 2 else (faces[j, (d + 2) % 4], faces[i, d])
            oL, oR, stop = f.reconstruct2(d, g["states"][i, 0], g["beds"][i, 0], g["states"][j, 0], g["beds"][j, 0], eL, eR)
            assert same(oL, g["r2L"][d, i]) and same(oR, g["r2R"][d, i]) and stop == g["r2stop"][d, i], (d, i)


def _run(sim, st, bed, man, n):
    sim.upload(st, bed, man)
    sim.set_target(1e9)
    return sim.run(n)


@pytest.mark.parametrize("precision", PRECISIONS)
def test_trajectories_rough_bed(precision):
    g = load_golden(f"f6_f7_trajectories_{precision}")
    st, bed, man = g["rough_state"], g["rough_bed"], g["rough_manning"]
    cases = [(oracle.GODUNOV, "god", "q", oracle.QUIRKS_REFERENCE),
             (oracle.GODUNOV, "god", "noq1", oracle.QUIRKS_REFERENCE & ~oracle.Q1_CFL_READS_PRIMARY),
             (oracle.MUSCL, "mch", "q", oracle.QUIRKS_REFERENCE)]
    for scheme, sname, qname, quirks in cases:
        sim = oracle.OracleSim(64, 64, scheme=scheme, precision=precision, quirks=quirks)
        sim.upload(st, bed, man)
        sim.set_target(1e9)
        t1 = sim.run(1)
        assert same(sim.download(), g[f"{sname}_{qname}_state1"])
        t2 = sim.run(199)
        assert same(np.concatenate([t1, t2]), g[f"{sname}_{qname}_dt"])
        assert same(sim.download(), g[f"{sname}_{qname}_state200"])
        assert sim.scalars()["t"] == g[f"{sname}_{qname}_t"]
    # quirk Q1 really changes the dt sequence
    assert not same(g["god_q_dt"], g["god_noq1_dt"])


@pytest.mark.parametrize("precision", PRECISIONS)
def test_trajectories_dam_break(precision):
    g = load_golden(f"f6_f7_trajectories_{precision}")
    from hipims_mi import synthetic as syn
    real = np.float64 if precision == "f64" else np.float32
    for wet, key in ((True, "dam"), (False, "damdry")):
        st, bed, man = syn.s_dam(96, 48, dtype=real, wet_right=wet)
        for scheme, sname in ((oracle.GODUNOV, "god"), (oracle.MUSCL, "mch")):
            if key == "damdry" and sname == "mch":
                continue
            sim = oracle.OracleSim(96, 48, scheme=scheme, precision=precision)
            dt = _run(sim, st, bed, man, 150)
            assert same(dt, g[f"{key}_{sname}_dt"])
            assert same(sim.download(), g[f"{key}_{sname}_state150"])


def test_mad_bracket():
    """FMA contraction (what the reference ships with) moves results by rounding only: this is the
    reference's own spread and sets the scale of the tolerance used for the FAST kernels."""
    a = load_golden("f6_f7_trajectories_f64")
    b = load_golden("f6_f7_trajectories_f64_mad")
    for key in ("dam_god_state150", "dam_mch_state150", "god_q_state200"):
        diff = np.abs(a[key] - b[key]).max()
        assert diff < 1e-9, (key, diff)
    assert np.abs(a["dam_god_dt"] - b["dam_god_dt"]).max() < 1e-12


@pytest.mark.parametrize("precision", PRECISIONS)
def test_time_control(precision):
    g = load_golden(f"f8_time_control_{precision}")
    sim = oracle.OracleSim(8, 8, precision=precision, end_time=float(g["end_time"]), dx=float(g["dx"]))
    lib, creal = sim.lib, sim.creal
    sc = sim.Scalars(t=0, dt=0.001, t_hydro=0, t_sync=5.0, batch_dt=0, batch_ok=0, batch_skipped=0)
    import ctypes as C
    saw_negative = False
    for i, v in enumerate(g["speeds"]):
        if i == 150:
            sc.t_sync = 12.0
        if i == 300:
            sc.t_sync = 1e9
        lib.orc_advance(C.byref(sim.p), C.byref(sc), creal(v))
        row = g["advance"][i]
        got = [sc.t, sc.dt, sc.t_hydro, sc.t_sync, sc.batch_dt, sc.batch_ok, sc.batch_skipped]
        assert np.array_equal(np.array(got, np.float64), row), (i, got, row)
        saw_negative |= sc.dt < 0
    assert saw_negative                      # the "suspended at sync point" signal was exercised
    for i in range(g["update_in"].shape[0]):
        t, dt, ts, bdt, v = g["update_in"][i]
        sc = sim.Scalars(t=t, dt=dt, t_hydro=0, t_sync=ts, batch_dt=bdt, batch_ok=0, batch_skipped=0)
        lib.orc_update_timestep(C.byref(sim.p), C.byref(sc), creal(v))
        assert np.array_equal(np.array([sc.dt, sc.batch_dt], np.float64), g["update_out"][i]), i


@pytest.mark.parametrize("precision", PRECISIONS)
def test_rain_boundaries(precision):
    g = load_golden(f"f9_rain_{precision}")
    cols, rows = g["bed"].shape[1], g["bed"].shape[0]
    for name in ("uniform", "gridded"):
        sim = oracle.OracleSim(cols, rows, precision=precision)
        sim.upload(g["state"], g["bed"], g["manning"])
        if name == "uniform":
            sim.add_uniform(oracle.UNIFORM_RAIN_INTENSITY, g["series"], 3600.0, 10800.0)
            sim.add_uniform(oracle.UNIFORM_LOSS_RATE, g["loss"], 10800.0, 10800.0)
        else:
            sim.add_gridded(oracle.GRIDDED_RAIN_INTENSITY, g["grids"], 10.0, 0.0, 0.0, 20.0)
        sim.set_target(1e9)
        dt = sim.run(420)
        assert same(dt, g[f"{name}_dt"])
        assert same(sim.download(), g[f"{name}_state"])
        assert sim.scalars()["t"] == g[f"{name}_t"]
    # quirk Q9 (truncated boundary NDRange) is visible: covering the whole grid changes the answer
    sim = oracle.OracleSim(cols, rows, precision=precision, quirks=oracle.QUIRKS_REFERENCE & ~oracle.Q9_BDY_TRUNCATED)
    sim.upload(g["state"], g["bed"], g["manning"])
    sim.add_gridded(oracle.GRIDDED_RAIN_INTENSITY, g["grids"], 10.0, 0.0, 0.0, 20.0)
    sim.set_target(1e9)
    sim.run(420)
    full = sim.download()
    assert (full[..., 0] - g["bed"]).sum() > (g["gridded_state"][..., 0] - g["bed"]).sum() > 0


CELL_MODES = [("depth_q", 2, 1), ("fsl_vel", 1, 2), ("free_q", 0, 1), ("free_volume", 0, 3)]


@pytest.mark.parametrize("precision", PRECISIONS)
def test_cell_boundary(precision):
    """bdy_Cell: four depth/discharge definitions, linear interpolation in time."""
    g = load_golden(f"f11_cell_boundary_{precision}")
    rows, cols = g["bed"].shape
    finals = []
    for name, dd, qd in CELL_MODES:
        sim = oracle.OracleSim(cols, rows, precision=precision)
        sim.upload(g["state"], g["bed"], g["manning"])
        sim.add_cell(dd, qd, g["cells"], g["series"], 5.0, 20.0)
        sim.set_target(1e9)
        dt = sim.run(300)
        assert same(dt, g[f"{name}_dt"])
        assert same(sim.download(), g[f"{name}_state"])
        assert sim.scalars()["t"] == g[f"{name}_t"]
        finals.append(g[f"{name}_state"])
        assert (g[f"{name}_state"][..., 0] - g["bed"]).max() > 0.05
    assert not same(finals[0], finals[1]) and not same(finals[2], finals[3])


def test_newcastle_example(tmp_path):
    """Config C1: the reference's example model (its own DEM file, rain + drainage), 900 iterations."""
    from hipims_mi import frontend
    from model_dir import make_newcastle
    g = load_golden("f10_newcastle_f64")
    cfg = frontend.parse_configuration(make_newcastle(tmp_path))
    st, bed, man, res = frontend.build_domain(cfg)
    sim = oracle.OracleSim(342, 195, dx=res, end_time=cfg.duration)
    sim.upload(st, bed, man)
    frontend.attach_boundaries(cfg, sim, 342)
    sim.set_target(1e9)
    dt = sim.run(900)
    assert same(dt, g["dt"])
    final = sim.download()
    assert same(final[..., 0], g["z"])
    assert sim.scalars()["t"] == g["t"]
    assert g["depth"].max() > 1e-4


# ---- partial-inertial scheme ----
@pytest.mark.parametrize("precision", PRECISIONS)
def test_inertial_flux_table(precision):
    g = load_golden(f"f12_inertial_{precision}")
    of = oracle.OracleFunctions(precision)
    got = np.array([of.inertial_flux(*a) for a in g["flux_args"]], g["flux_out"].dtype)
    assert same(got, g["flux_out"])
    assert (g["flux_out"] != 0).sum() > 1000 and (g["flux_out"] == 0).sum() > 100      # both regimes are in the table


@pytest.mark.parametrize("precision", PRECISIONS)
def test_inertial_trajectories(precision):
    from hipims_mi import synthetic as syn
    g = load_golden(f"f12_inertial_{precision}")
    real = np.float64 if precision == "f64" else np.float32
    st, bed, man = g["rough_state"], g["rough_bed"], g["rough_manning"]
    for qname, quirks in (("q", oracle.QUIRKS_REFERENCE), ("noq1", oracle.QUIRKS_REFERENCE & ~oracle.Q1_CFL_READS_PRIMARY)):
        sim = oracle.OracleSim(64, 64, scheme=oracle.INERTIAL, precision=precision, quirks=quirks)
        sim.upload(st, bed, man)
        sim.set_target(1e9)
        dts = [sim.run(1)]
        if qname == "q":
            assert same(sim.download(), g["rough_q_state1"])
        dts.append(sim.run(199))
        assert same(np.concatenate(dts), g[f"rough_{qname}_dt"])
        assert same(sim.download(), g[f"rough_{qname}_state"])
        assert sim.scalars()["t"] == g[f"rough_{qname}_t"]
    for kw, key in ((dict(), "dam"), (dict(wet_right=False), "damdry"), (dict(levels=(2.0, 1.6)), "step")):
        st, bed, man = syn.s_dam(96, 48, dtype=real, **kw)
        sim = oracle.OracleSim(96, 48, scheme=oracle.INERTIAL, precision=precision)
        assert same(_run(sim, st, bed, man, 150), g[f"{key}_dt"])
        assert same(sim.download(), g[f"{key}_state"])


@pytest.mark.parametrize("precision", PRECISIONS)
def test_inertial_sync_point_and_rain(precision):
    g = load_golden(f"f12_inertial_{precision}")
    sim = oracle.OracleSim(48, 40, scheme=oracle.INERTIAL, precision=precision)
    sim.upload(g["sync_state"], g["sync_bed"], g["sync_manning"])
    sim.set_target(2.0)
    assert same(sim.run(40), g["sync_a_dt"])
    assert same(sim.download(), g["sync_a_state"])
    assert sim.scalars()["batch_skipped"] > 0                      # the target was reached: skipped iterations happened
    sim.set_target(5.0)
    sim.update_timestep()
    assert same(sim.run(45), g["sync_b_dt"])
    assert same(sim.download(), g["sync_b_state"])
    assert sim.scalars()["t"] == g["sync_b_t"]

    sim = oracle.OracleSim(48, 40, scheme=oracle.INERTIAL, precision=precision)
    sim.upload(g["rain_init"], g["rain_bed"], g["rain_manning"])
    sim.set_target(1e9)
    sim.add_uniform(oracle.UNIFORM_RAIN_INTENSITY, g["rain_series"], 10.0, 30.0)
    sim.add_gridded(oracle.GRIDDED_RAIN_INTENSITY, g["rain_grids"], 10.0, 0.0, 0.0, 15.0)
    assert same(sim.run(260), g["rain_dt"])
    assert same(sim.download(), g["rain_state"])
    assert (g["rain_state"][..., 0] - g["rain_bed"]).max() > 1e-5          # it rained


def test_fixed_timestep_program():
    """TIMESTEP_FIXED: dt is the configured value, but still clipped by the sync time, the early limit and the end time."""
    g = load_golden("f13_fixed_timestep_f64")
    for name, dt, target, end in (("small", 0.02, 2.5, 1e30), ("clipped", 0.25, 1e9, 11.0)):
        sim = oracle.OracleSim(64, 64, dynamic_dt=False, fixed_dt=dt, dt_initial=dt, end_time=end)
        sim.upload(g["state"], g["bed"], g["manning"])
        sim.set_target(target)
        assert same(sim.run(160), g[f"{name}_dt"])
        assert same(sim.download(), g[f"{name}_state"])
        sc = sim.scalars()
        assert (sc["t"], sc["batch_ok"], sc["batch_skipped"]) == (g[f"{name}_t"], g[f"{name}_ok"], g[f"{name}_skipped"])
        if name == "small":
            sim.set_target(4.0)
            sim.update_timestep()
            assert same(sim.run(100), g["resume_dt"]) and same(sim.download(), g["resume_state"])
            assert sim.scalars()["t"] == g["resume_t"] == 4.0 and g["resume_dt"][0] == 0.02
    assert g["clipped_dt"][0] == 0.25 and g["clipped_dt"][1] == 0.1 and g["clipped_dt"][-1] == 0.0 and g["clipped_t"] == 11.0


def test_no_friction_program():
    g = load_golden("f14_no_friction_f64")
    for scheme, name in ((oracle.GODUNOV, "god"), (oracle.MUSCL, "mch")):
        sim = oracle.OracleSim(72, 40, scheme=scheme, friction=False)
        assert same(_run(sim, g["state"], g["bed"], g["manning"], 120), g[f"{name}_dt"])
        assert same(sim.download(), g[f"{name}_state"]) and sim.scalars()["t"] == g[f"{name}_t"]


def test_disabled_cells():
    """The domain's nulls: cells with Zmax = -9999 (or Z == -9999) are carried unchanged, neighbours see their level."""
    g = load_golden("f15_disabled_cells_f64")
    # `mch`: the reference's default predictor (mch_1st_cachePrediction as real work-groups: neighbours' .y = bed);
    # `mchnone`: mch_1st_cacheNone (neighbours' .y = Zmax) -- quirk Q11 on / off
    for scheme, name, quirks in ((oracle.GODUNOV, "god", oracle.QUIRKS_REFERENCE), (oracle.MUSCL, "mch", oracle.QUIRKS_REFERENCE),
                                 (oracle.MUSCL, "mchnone", oracle.QUIRKS_REFERENCE & ~oracle.Q11_MUSCL_NB_Y_IS_BED),
                                 (oracle.INERTIAL, "ine", oracle.QUIRKS_REFERENCE)):
        sim = oracle.OracleSim(64, 48, scheme=scheme, quirks=quirks)
        sim.upload(g["state"], g["bed"], g["manning"])
        sim.set_target(2.5)
        assert same(sim.run(150), g[f"{name}_dt"])
        out = sim.download()
        assert same(out, g[f"{name}_state"]) and sim.scalars()["t"] == g[f"{name}_t"]
        assert same(out[g["disabled"]], g["state"][g["disabled"]]) and g["disabled"].sum() > 100
    assert not same(g["mch_state"], g["mchnone_state"])


def test_pow_stand_in_bracket_beyond_c1():
    """Fixture F17: the F6 / F7 / F12 rough-bed trajectories on the strict program with the host libm's pow instead of the
    correctly rounded cube root every other build here uses (oracle/ref_build/shim.cpp).  The oracle (cube root) equals the
    cube-root build bit for bit (tests above); the libm build sits 6e-17 ... 3e-16 m RMSE away -- the independent bound on the
    one stand-in that carries arithmetic, on Godunov, MUSCL-Hancock and the partial-inertial scheme."""
    twin = load_golden("f17_libm_twins_f64")
    g, gi = load_golden("f6_f7_trajectories_f64"), load_golden("f12_inertial_f64")
    bed = g["rough_bed"]
    depth = lambda s: np.maximum(0, s[..., 0] - bed)
    for key, ref in (("god_q_state200", g["god_q_state200"]), ("mch_q_state200", g["mch_q_state200"]),
                     ("ine_rough_q_state", gi["rough_q_state"])):
        d = depth(twin[key]) - depth(ref)
        rmse, mx = float(np.sqrt(np.mean(d ** 2))), float(np.abs(d).max())
        assert 0 < rmse < 1e-15 and mx < 1e-14, (key, rmse, mx)       # not identical (another pow), and nowhere near the tolerance
    assert float(twin["god_q_t"]) == float(g["god_q_t"]) and float(twin["mch_q_t"]) == float(g["mch_q_t"])


def test_newcastle_full_run_first_output(tmp_path):
    """Fixture F16 (config C1 to its end on the reference's kernels, tests/golden/generate.py: newcastle_full): the oracle under the
    same CModel loop, up to the first output time (600 s, 3.7e3 iterations) -- depth and maxdepth rasters bit for bit.  (The whole
    run is the GPU suite's: tests/test_gpu_c1_full.py; on the CPU it takes minutes.  Round 5: running it to the end is what
    exposed a race in the THREADED reference build's reduction -- oracle/ref_build/shim.cpp: ref_reduce.)"""
    import hashlib
    import os
    from hipims_mi.model import Model
    from model_dir import make_newcastle
    g = load_golden("f16_newcastle_full_f64")

    def make_sim(cfg, cols, rows, res):
        return oracle.OracleSim(cols, rows, precision="f64", dx=res, end_time=cfg.duration, threads=min(8, os.cpu_count() or 1))
    m = Model(make_newcastle(tmp_path), make_sim=make_sim, output_format=None)
    m.scheme.automatic_queue = False
    m.scheme.queue_addition_size = int(g["batch"])
    outs = m.run(max_outputs=1)
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a, dtype=np.float64).tobytes()).hexdigest()
    assert outs[0][0] == float(g["t"][0])
    assert sha(outs[0][1]["depth"]) == str(g["depth_sha256"][0]) and sha(outs[0][1]["maxdepth"]) == str(g["maxdepth_sha256"][0])
    assert np.array_equal(outs[0][1]["depth"], g["depth"][0])
