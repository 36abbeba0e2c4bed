"""Parity of the HIP engine (through the C ABI) with the oracle and the golden fixtures.  GPU only.

Tolerances (fp64, north_star): depth RMSE < 1e-9 m, max |depth diff| < 1e-7 m, dt relative < 1e-12.
STRICT arithmetic is additionally required to be bit-identical (with friction too: tests/test_gpu_strict_friction.py).
fp32: depth RMSE < 1e-4 m.
"""
import os

import numpy as np
import pytest

import hipims_mi as hp
import oracle
from conftest import load_golden, record
from hipims_mi import synthetic as syn

pytestmark = pytest.mark.gpu

KERNELS = [hp.KERNEL_AUTO, hp.KERNEL_BASIC]
MODES = [hp.MATH_FAST, hp.MATH_STRICT]


def compare(dom, ref, precision="f64", check_t=True):
    dg, ug, vg = dom.depth_velocity()
    dr, ur, vr = ref.depth_velocity()
    rmse = float(np.sqrt(np.mean((dg - dr) ** 2)))
    mx = float(np.abs(dg - dr).max())
    if precision == "f64":
        assert rmse < 1e-9 and mx < 1e-7, (rmse, mx)
        assert np.abs(ug - ur).max() < 1e-5 and np.abs(vg - vr).max() < 1e-5
    else:
        assert rmse < 1e-4, rmse
    if check_t:
        sc, sr = dom.read_scalars(), ref.scalars()
        tol = 1e-12 if precision == "f64" else 1e-4
        assert abs(sc["time"] - sr["t"]) <= tol * max(1.0, abs(sr["t"])), (sc["time"], sr["t"])
        assert sc["batch_successful"] == sr["batch_ok"] and sc["batch_skipped"] == sr["batch_skipped"]
    return rmse, mx


def make_pair(cols, rows, st, bed, man, precision="f64", scheme=0, quirks=hp.QUIRKS_REFERENCE, dx=1.0, **kw):
    oq = oracle.quirks_from_engine(quirks, muscl_serial=scheme != hp.SCHEME_MUSCL_HANCOCK)
    ref = oracle.OracleSim(cols, rows, scheme=scheme, precision=precision, quirks=oq, dx=dx,
                           end_time=kw.get("t_end", 1e30))
    dom = hp.Domain(cols, rows, scheme=scheme, precision=precision, quirks=quirks, dx=dx, **kw)
    for s in (ref, dom):
        s.upload(st, bed, man)
    return dom, ref


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("mode", MODES)
def test_godunov_rough_bed_200_steps(kernel, mode):
    st, bed, man = syn.s_rough(64, 64, manning=None)
    dom, ref = make_pair(64, 64, st, bed, man, kernel=kernel, math_mode=mode)
    dom.set_target_time(1e9); ref.set_target(1e9)
    tr_ref = ref.run(200)
    tr_gpu = dom.run(200)
    assert np.abs(tr_gpu - tr_ref).max() <= 1e-12 * tr_ref.max()
    compare(dom, ref)
    # and against the committed fixture produced by the reference's own kernels
    g = load_golden("f6_f7_trajectories_f64")
    depth_g = np.maximum(0, g["god_q_state200"][..., 0] - bed)
    dg, _, _ = dom.depth_velocity()
    assert np.sqrt(np.mean((dg - depth_g) ** 2)) < 1e-9


@pytest.mark.parametrize("kernel", KERNELS)
def test_strict_mode_is_bit_identical_without_friction(kernel):
    """With friction off no transcendental is involved: STRICT must reproduce the oracle bit for bit."""
    st, bed, man = syn.s_rough(72, 40, manning=None)
    ref = oracle.OracleSim(72, 40, friction=False)
    dom = hp.Domain(72, 40, friction=False, math_mode=hp.MATH_STRICT, kernel=kernel)
    for s in (ref, dom):
        s.upload(st, bed, man)
    dom.set_target_time(1e9); ref.set_target(1e9)
    ref.run(120); dom.step_batch(120)
    assert np.array_equal(dom.download(), ref.download())
    assert dom.read_scalars()["time"] == ref.scalars()["t"]
    # and bit for bit against what the reference's own no-friction program produced (fixture F14)
    g = load_golden("f14_no_friction_f64")
    assert np.array_equal(st, g["state"]) and np.array_equal(dom.download(), g["god_state"])
    assert dom.read_scalars()["time"] == float(g["god_t"])


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("wet", [True, False])
def test_dam_break_fixture(kernel, wet):
    key = "dam" if wet else "damdry"
    g = load_golden("f6_f7_trajectories_f64")
    st, bed, man = syn.s_dam(96, 48, wet_right=wet)
    dom = hp.Domain(96, 48, kernel=kernel)
    dom.upload(st, bed, man)
    dom.set_target_time(1e9)
    dom.step_batch(150)
    out = dom.download()
    dg = np.maximum(0, out[..., 0] - bed)
    dr = np.maximum(0, g[f"{key}_god_state150"][..., 0] - bed)
    assert np.sqrt(np.mean((dg - dr) ** 2)) < 1e-9 and np.abs(dg - dr).max() < 1e-7
    t_ref = float(g[f"{key}_god_dt"].sum())
    assert abs(dom.read_scalars()["time"] - t_ref) < 1e-10


def test_quirk_q1_off_matches_oracle():
    st, bed, man = syn.s_rough(64, 64, manning=None)
    q = hp.QUIRKS_REFERENCE & ~hp.QUIRK_CFL_READS_PRIMARY
    for kernel in KERNELS:
        dom, ref = make_pair(64, 64, st, bed, man, quirks=q, kernel=kernel)
        dom.set_target_time(1e9); ref.set_target(1e9)
        ref.run(100); dom.step_batch(100)
        compare(dom, ref)


@pytest.mark.parametrize("kernel", KERNELS)
def test_fp32(kernel):
    st, bed, man = syn.s_rough(64, 64, dtype=np.float32, manning=None)
    dom, ref = make_pair(64, 64, st, bed, man, precision="f32", kernel=kernel, math_mode=hp.MATH_STRICT)
    dom.set_target_time(1e9); ref.set_target(1e9)
    ref.run(100); dom.step_batch(100)
    compare(dom, ref, precision="f32")


@pytest.mark.parametrize("kernel", KERNELS)
def test_sync_point_suspends_and_resumes(kernel):
    """Target-time clipping: dt = t_sync - t, then negative dt = suspended (skipped iterations copy src->dst)."""
    st, bed, man = syn.s_dam(64, 32)
    dom, ref = make_pair(64, 32, st, bed, man, kernel=kernel)
    dom.set_target_time(0.35); ref.set_target(0.35)
    ref.run(40); dom.step_batch(40)
    sc = dom.read_scalars()
    assert sc["timestep"] < 0 and abs(sc["time"] - 0.35) < 1e-12 and sc["batch_skipped"] > 0
    compare(dom, ref)
    dom.set_target_time(0.8); ref.set_target(0.8)
    dom.force_timestep(abs(sc["timestep"])); ref.force_dt(abs(ref.scalars()["dt"]))
    dom.reset_counters(); ref.reset_counters()
    ref.run(60); dom.step_batch(60)
    compare(dom, ref)


@pytest.mark.parametrize("kernel", KERNELS)
def test_uniform_rain_and_gridded_rain_fixture(kernel):
    g = load_golden("f9_rain_f64")
    rows, cols = g["bed"].shape
    for name in ("uniform", "gridded"):
        dom = hp.Domain(cols, rows, kernel=kernel)
        dom.upload(g["state"], g["bed"], g["manning"])
        if name == "uniform":
            dom.add_uniform(hp.UNIFORM_RAIN_INTENSITY, g["series"], 3600.0, 10800.0)
            dom.add_uniform(hp.UNIFORM_LOSS_RATE, g["loss"], 10800.0, 10800.0)
        else:
            dom.add_gridded(hp.GRIDDED_RAIN_INTENSITY, g["grids"], 10.0, 0.0, 0.0, 20.0)
        dom.set_target_time(1e9)
        dom.step_batch(420)
        out = dom.download()
        dg = np.maximum(0, out[..., 0] - g["bed"])
        dr = np.maximum(0, g[f"{name}_state"][..., 0] - g["bed"])
        assert dr.max() > 1e-4
        assert np.sqrt(np.mean((dg - dr) ** 2)) < 1e-9 and np.abs(dg - dr).max() < 1e-7, name
        assert abs(dom.read_scalars()["time"] - float(g[f"{name}_t"])) < 1e-9


@pytest.mark.parametrize("mode", MODES)
def test_newcastle_example_rain_drainage(mode, tmp_path):
    """Config C1: the reference's example model (its own 342x195 @ 2 m DEM file, rain 70 mm/h + drainage 12 mm/h,
    closed edges) read through the front end, 900 iterations, against the fixtures from the reference's kernels.

    STRICT reproduces the reference kernels' run itself (RMSE 0: tests/test_gpu_strict_friction.py holds every bit).
    FAST cannot: steep urban terrain under a millimetre-thin rain film amplifies rounding -- wet/dry decisions at
    VERY_SMALL flip a step earlier or later -- and the reference's OWN builds differ from one another here.  Three of
    them are committed as fixtures: strict (f10_newcastle_f64), as shipped with -cl-mad-enable (COCLProgram.cpp:73;
    _mad) and strict on a platform with a different conforming pow() (_libm).  Their largest pairwise spread after these
    900 iterations is depth RMSE 4.1e-9 m, max 4.5e-7 m, 1.5e-8 relative in elapsed time; FAST is held to TWICE that
    ensemble spread in each measure (achieved: 5.8e-9 m, 5.2e-7 m, 1.3e-8; profiles/r03*_parity_numbers.jsonl)."""
    from itertools import combinations
    from hipims_mi import frontend
    from model_dir import make_newcastle
    refs = [load_golden("f10_newcastle_f64"), load_golden("f10_newcastle_f64_mad"), load_golden("f10_newcastle_f64_libm")]
    g = refs[0]
    cfg = frontend.parse_configuration(make_newcastle(tmp_path))
    st, bed, man, res = frontend.build_domain(cfg)
    depth = lambda z: np.maximum(0, z - bed)
    dr = depth(g["z"])
    pairs = list(combinations(refs, 2))
    bracket_rmse = max(np.sqrt(np.mean((depth(a["z"]) - depth(b["z"])) ** 2)) for a, b in pairs)
    bracket_max = max(np.abs(depth(a["z"]) - depth(b["z"])).max() for a, b in pairs)
    bracket_t = max(abs(float(a["t"]) - float(b["t"])) for a, b in pairs) / float(g["t"])
    assert 1e-9 < bracket_rmse < 1e-8 and bracket_t < 1e-7     # the reference's own spread, as recorded above

    dom = hp.Domain(342, 195, dx=res, t_end=cfg.duration, math_mode=mode)
    dom.upload(st, bed, man)
    frontend.attach_boundaries(cfg, dom, 342)
    dom.set_target_time(1e9)
    dom.step_batch(900)
    dg = depth(dom.download()[..., 0])
    rmse, mx = float(np.sqrt(np.mean((dg - dr) ** 2))), float(np.abs(dg - dr).max())
    dt_rel = abs(dom.read_scalars()["time"] - float(g["t"])) / float(g["t"])
    record("c1_newcastle_900", mode="strict" if mode == hp.MATH_STRICT else "fast", rmse=rmse, max=mx, dt_rel=dt_rel,
           bracket_rmse=float(bracket_rmse), bracket_max=float(bracket_max), bracket_t=float(bracket_t))
    if mode == hp.MATH_STRICT:
        assert rmse == 0.0 and dt_rel == 0.0                   # north_star: depth RMSE < 1e-9 m vs the reference run
    else:
        assert rmse < 2 * bracket_rmse and mx < 2 * bracket_max and dt_rel < 2 * bracket_t


def test_front_end_runs_the_example_on_the_gpu(tmp_path):
    """run_model end to end on the HIP engine vs the same front end on the oracle: rasters at both output times."""
    import oracle as orc
    from hipims_mi import frontend
    from model_dir import make_newcastle
    xml = make_newcastle(tmp_path, duration=120, frequency=60)
    gpu = frontend.run_model(xml, batch=64, output_format=".asc")
    ref = frontend.run_model(xml, batch=64, output_format=None, make_sim=lambda c, cols, rows, res: orc.OracleSim(
        cols, rows, dx=res, courant=c.courant, end_time=c.duration, friction=c.friction, threads=8))
    assert [round(t, 6) for t, _ in gpu] == [60.0, 120.0]
    for (tg, og), (tr, orr) in zip(gpu, ref):
        assert abs(tg - tr) < 1e-9
        for what in ("depth", "fsl", "maxdepth"):
            a, b = og[what], orr[what]
            both = (a != frontend.NODATA) & (b != frontend.NODATA)
            assert (a != frontend.NODATA).sum() == (b != frontend.NODATA).sum() or abs(int((a != frontend.NODATA).sum()) - int((b != frontend.NODATA).sum())) < 20
            assert np.abs(a[both] - b[both]).max() < 1e-6          # same thin-film case, see the bracket above
    back, info = frontend.read_raster(os.path.join(str(tmp_path), "output", "depth_120.asc"))
    assert np.allclose(back, gpu[-1][1]["depth"], rtol=1e-9, atol=1e-12) and info["pixel_size"] == (2.0, 2.0)


@pytest.mark.parametrize("name,dd,qd", [("depth_q", 2, 1), ("fsl_vel", 1, 2), ("free_q", 0, 1), ("free_volume", 0, 3)])
def test_cell_boundary_fixture(name, dd, qd):
    """bdy_Cell against the fixture produced by the reference's kernel (row N2 of SURVEY 8f)."""
    g = load_golden("f11_cell_boundary_f64")
    rows, cols = g["bed"].shape
    dom = hp.Domain(cols, rows)
    dom.upload(g["state"], g["bed"], g["manning"])
    dom.add_cell(dd, qd, g["cells"], g["series"], 5.0, 20.0)
    dom.set_target_time(1e9)
    dom.step_batch(300)
    out = dom.download()
    dg = np.maximum(0, out[..., 0] - g["bed"])
    dr = np.maximum(0, g[f"{name}_state"][..., 0] - g["bed"])
    assert dr.max() > 0.05
    assert np.sqrt(np.mean((dg - dr) ** 2)) < 1e-9 and np.abs(dg - dr).max() < 1e-7
    assert abs(dom.read_scalars()["time"] - float(g[f"{name}_t"])) < 1e-9


def test_cell_boundary_rejects_bad_input():
    dom = hp.Domain(32, 32)
    series = np.zeros((3, 4)); series[:, 0] = [0, 5, 10]
    with pytest.raises(hp.HipimsError, match="outside the grid"):
        dom.add_cell(2, 1, [32 * 32], series, 5.0, 10.0)
    with pytest.raises(hp.HipimsError, match="shorter than its length"):
        dom.add_cell(2, 1, [40], series, 5.0, 20.0)


def test_partial_transfers_and_busy_flag():
    st, bed, man = syn.s_rough(48, 40, manning=None)
    dom = hp.Domain(48, 40)
    dom.upload(st, bed, man)
    assert np.array_equal(dom.download(), st)
    assert np.array_equal(dom.download(hp.ARRAY_BED, 3, 5), bed[3:8])
    patch = st[10:12].copy(); patch[..., 0] += 0.25
    dom.upload_rows(patch, 10)
    assert np.array_equal(dom.download(hp.ARRAY_STATE, 10, 2), patch)
    dom.sync()
    assert dom.is_busy() is False


# ---- MUSCL-Hancock (config C3): fused predictor+corrector kernel, double buffered ----
MUSCL = hp.SCHEME_MUSCL_HANCOCK


@pytest.mark.parametrize("mode", MODES)
def test_muscl_rough_bed_200_steps(mode):
    """Wet/dry rough terrain.  The oracle runs the corrector in snapshot order (what a double-buffered kernel does;
    the reference's in-place corrector is work-item-order dependent, quirk Q6)."""
    st, bed, man = syn.s_rough(64, 64, manning=None)
    dom, ref = make_pair(64, 64, st, bed, man, scheme=MUSCL, math_mode=mode)
    dom.set_target_time(1e9); ref.set_target(1e9)
    tr_ref = ref.run(200)
    tr_gpu = dom.run(200)
    assert np.abs(tr_gpu - tr_ref).max() <= 1e-12 * tr_ref.max()
    compare(dom, ref)
    # ... and against the reference's own kernels: on this case their serial in-place order and the snapshot order give the
    # same bits (fixture F7 `mch_q_state200`, tests/golden/generate.py: trajectories)
    g = load_golden("f6_f7_trajectories_f64")
    assert np.array_equal(g["rough_state"], st) and np.array_equal(g["rough_bed"], bed)
    out = dom.download()
    dg, dr = np.maximum(0, out[..., 0] - bed), np.maximum(0, g["mch_q_state200"][..., 0] - bed)
    rmse, mx = float(np.sqrt(np.mean((dg - dr) ** 2))), float(np.abs(dg - dr).max())
    record("muscl_rough64_200_vs_f7", mode="strict" if mode == hp.MATH_STRICT else "fast", rmse=rmse, max=mx)
    if mode == hp.MATH_STRICT:
        assert np.array_equal(out, g["mch_q_state200"]) and dom.read_scalars()["time"] == float(g["mch_q_t"])
    else:
        assert rmse < 1e-9 and mx < 1e-7 and abs(dom.read_scalars()["time"] - float(g["mch_q_t"])) <= 1e-12 * float(g["mch_q_t"])


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("scheme,key,ref_file,ref_key", [(hp.SCHEME_GODUNOV, "god_q_state200", "f6_f7_trajectories_f64", "god_q_state200"),
                                                         (hp.SCHEME_MUSCL_HANCOCK, "mch_q_state200", "f6_f7_trajectories_f64", "mch_q_state200"),
                                                         (hp.SCHEME_INERTIAL, "ine_rough_q_state", "f12_inertial_f64", "rough_q_state")])
def test_the_pow_stand_in_is_bracketed_beyond_c1(scheme, key, ref_file, ref_key, mode):
    """The one stand-in of the reference build that carries arithmetic is pow (the correctly rounded cube root shared by oracle,
    reference build and STRICT kernels).  Fixture F17 is the same strict program with the host libm's pow: on the rough-bed
    trajectories of all three schemes the two builds part by 6e-17 ... 3e-16 m RMSE (tests/test_oracle_golden.py records it).
    STRICT equals the cube-root build bit for bit, so its distance to the libm build IS that spread; FAST is held to 1e-9 m
    against both builds -- the pow a platform ships moves the answer seven orders of magnitude less than the tolerance."""
    g, twin = load_golden(ref_file), load_golden("f17_libm_twins_f64")
    st, bed, man = g["rough_state"], g["rough_bed"], g["rough_manning"]
    if scheme == hp.SCHEME_INERTIAL:
        st = st.copy(); st[..., 2:] = 0
    dom = hp.Domain(64, 64, scheme=scheme, math_mode=mode)
    dom.upload(st, bed, man)
    dom.set_target_time(1e9)
    dom.step_batch(200)
    depth = lambda s: np.maximum(0, s[..., 0] - bed)
    out = dom.download()
    r = lambda a, b: float(np.sqrt(np.mean((depth(a) - depth(b)) ** 2)))
    spread = r(g[ref_key], twin[key])
    record("pow_bracket", scheme=scheme, mode="strict" if mode == hp.MATH_STRICT else "fast", spread=spread,
           to_crmath_build=r(out, g[ref_key]), to_libm_build=r(out, twin[key]))
    assert spread < 1e-15
    if mode == hp.MATH_STRICT:
        assert np.array_equal(out, g[ref_key]) and r(out, twin[key]) == spread
    else:
        assert r(out, g[ref_key]) < 1e-9 and r(out, twin[key]) < 1e-9
    dom.close()


def test_muscl_strict_is_bit_identical_without_friction():
    st, bed, man = syn.s_rough(72, 40, manning=None)
    ref = oracle.OracleSim(72, 40, scheme=oracle.MUSCL, friction=False, quirks=oracle.QUIRKS_REFERENCE & ~oracle.Q6_MUSCL_SERIAL)
    dom = hp.Domain(72, 40, scheme=MUSCL, friction=False, math_mode=hp.MATH_STRICT)
    for s in (ref, dom):
        s.upload(st, bed, man)
    dom.set_target_time(1e9); ref.set_target(1e9)
    ref.run(120); dom.step_batch(120)
    assert np.array_equal(dom.download(), ref.download())
    assert dom.read_scalars()["time"] == ref.scalars()["t"]


def test_muscl_dam_break_fixture():
    """All-wet dam break: the reference's own (serial, in-place) result applies, so the committed fixture is the check."""
    g = load_golden("f6_f7_trajectories_f64")
    st, bed, man = syn.s_dam(96, 48)
    dom = hp.Domain(96, 48, scheme=MUSCL)
    dom.upload(st, bed, man)
    dom.set_target_time(1e9)
    dom.step_batch(150)
    out = dom.download()
    dg = np.maximum(0, out[..., 0] - bed)
    dr = np.maximum(0, g["dam_mch_state150"][..., 0] - bed)
    assert np.sqrt(np.mean((dg - dr) ** 2)) < 1e-9 and np.abs(dg - dr).max() < 1e-7
    assert abs(dom.read_scalars()["time"] - float(g["dam_mch_dt"].sum())) < 1e-10


def test_muscl_fp32_and_sync_point():
    st, bed, man = syn.s_rough(64, 48, dtype=np.float32, manning=None)
    dom, ref = make_pair(64, 48, st, bed, man, precision="f32", scheme=MUSCL, math_mode=hp.MATH_STRICT)
    dom.set_target_time(0.5); ref.set_target(0.5)
    ref.run(90); dom.step_batch(90)
    assert dom.read_scalars()["batch_skipped"] > 0
    compare(dom, ref, precision="f32")


def test_muscl_full_size_4096_properties():
    n = 4096
    st, bed, man = syn.s_dam(n, n)
    dom = hp.Domain(n, n, scheme=MUSCL)
    dom.upload(st, bed, man)
    dom.set_target_time(1e9)
    dom.step_batch(20)
    out = dom.download()
    # (no volume check: the reference's corrector leaves ring 1 untouched (:569-573) while it exchanges flux with
    #  ring 2, so S-DAM's wet ring-1 cells act as a reservoir -- reference semantics, not conservative)
    assert np.array_equal(out[:2], st[:2]) and np.array_equal(out[:, :2], st[:, :2])       # 2-cell ring never written
    assert np.abs(out[:, :, 0] - out[::-1, :, 0]).max() < 1e-11
    # second-order scheme differs from the first-order one, but only near the front
    god = hp.Domain(n, n)
    god.upload(st, bed, man)
    god.set_target_time(1e9)
    god.step_batch(20)
    diff = np.abs(out[..., 0] - god.download()[..., 0])
    assert diff.max() > 0 and (diff > 1e-6).mean() < 0.05


# ---- BASELINE.json's full size: properties that need no oracle run ----
def test_full_size_4096_mass_conservation_and_kernel_agreement():
    n = 4096
    st, bed, man = syn.s_dam(n, n)
    d0 = np.maximum(0, st[..., 0] - bed).sum()
    outs = []
    for kernel in KERNELS:
        dom = hp.Domain(n, n, kernel=kernel)
        dom.upload(st, bed, man)
        dom.set_target_time(1e9)
        dom.step_batch(30)
        out = dom.download()
        outs.append(out)
        d1 = np.maximum(0, out[..., 0] - bed).sum()
        assert abs(d1 - d0) / d0 < 1e-12                       # closed basin: volume conserved
        assert np.abs(out[:, :, 0] - out[::-1, :, 0]).max() < 1e-11   # north-south mirror symmetry of S-DAM
        assert dom.read_scalars()["cells_calculated"] == 30 * n * n
        dom.close()
    assert np.abs(outs[0] - outs[1]).max() < 1e-9              # tuned kernel == basic kernel


# ---- partial-inertial scheme: K6 inertial_march ----
@pytest.mark.parametrize("mode", MODES)
def test_inertial_rough_bed_200_steps(mode):
    """Spatially varying Manning n: four flux evaluations per cell (the reference's face values differ per side)."""
    g = load_golden("f12_inertial_f64")
    st, bed, man = g["rough_state"], g["rough_bed"], g["rough_manning"]
    dom, ref = make_pair(64, 64, st, bed, man, scheme=hp.SCHEME_INERTIAL, math_mode=mode)
    dom.set_target_time(1e9); ref.set_target(1e9)
    tr_ref, tr_gpu = ref.run(200), dom.run(200)
    assert np.abs(tr_gpu - tr_ref).max() <= 1e-12 * tr_ref.max()
    compare(dom, ref)
    depth_g = np.maximum(0, g["rough_q_state"][..., 0] - bed)
    dg, _, _ = dom.depth_velocity()
    assert np.sqrt(np.mean((dg - depth_g) ** 2)) < 1e-9 and np.abs(dg - depth_g).max() < 1e-7
    # the stored face discharges themselves
    out = dom.download()
    assert np.abs(out[..., 2:] - g["rough_q_state"][..., 2:]).max() < 1e-9


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("key,kw", [("step", dict(levels=(2.0, 1.6))), ("damdry", dict(wet_right=False))])
def test_inertial_uniform_manning_fixture(mode, key, kw):
    """Uniform n: the kernel shares each face between its two cells (two evaluations per cell)."""
    g = load_golden("f12_inertial_f64")
    st, bed, man = syn.s_dam(96, 48, **kw)
    dom = hp.Domain(96, 48, scheme=hp.SCHEME_INERTIAL, math_mode=mode)
    dom.upload(st, bed, man)
    dom.set_target_time(1e9)
    dom.step_batch(150)
    out = dom.download()
    dg = np.maximum(0, out[..., 0] - bed)
    dr = np.maximum(0, g[f"{key}_state"][..., 0] - bed)
    assert np.sqrt(np.mean((dg - dr) ** 2)) < 1e-9 and np.abs(dg - dr).max() < 1e-7
    assert abs(dom.read_scalars()["time"] - float(g[f"{key}_t"])) < 1e-10


def test_inertial_sync_point_leaves_dst_untouched():
    """dt <= 0: ine_cacheDisabled returns before writing (CLSchemeInertial.clc:61-62), so skipped iterations flip to
    a buffer that still holds the state of two iterations earlier -- the fixture from the reference's kernel has it."""
    g = load_golden("f12_inertial_f64")
    dom = hp.Domain(48, 40, scheme=hp.SCHEME_INERTIAL, math_mode=hp.MATH_STRICT)
    dom.upload(g["sync_state"], g["sync_bed"], g["sync_manning"])
    dom.set_target_time(2.0)
    dom.step_batch(40)
    sc = dom.read_scalars()
    assert sc["timestep"] < 0 and sc["batch_skipped"] > 0 and abs(sc["time"] - 2.0) < 1e-12
    assert np.abs(dom.download() - g["sync_a_state"]).max() < 1e-9
    dom.set_target_time(5.0)
    dom.update_timestep()
    dom.step_batch(45)
    assert np.abs(dom.download() - g["sync_b_state"]).max() < 1e-9
    assert abs(dom.read_scalars()["time"] - float(g["sync_b_t"])) < 1e-10


def test_inertial_rain_and_fp32():
    g = load_golden("f12_inertial_f64")
    dom = hp.Domain(48, 40, scheme=hp.SCHEME_INERTIAL)
    dom.upload(g["rain_init"], g["rain_bed"], g["rain_manning"])
    dom.add_uniform(hp.UNIFORM_RAIN_INTENSITY, g["rain_series"], 10.0, 30.0)
    dom.add_gridded(hp.GRIDDED_RAIN_INTENSITY, g["rain_grids"], 10.0, 0.0, 0.0, 15.0)
    dom.set_target_time(1e9)
    dom.step_batch(260)
    dg = np.maximum(0, dom.download()[..., 0] - g["rain_bed"])
    dr = np.maximum(0, g["rain_state"][..., 0] - g["rain_bed"])
    assert dr.max() > 1e-5 and np.sqrt(np.mean((dg - dr) ** 2)) < 1e-9 and np.abs(dg - dr).max() < 1e-7
    g32 = load_golden("f12_inertial_f32")
    for mode in MODES:
        dom = hp.Domain(64, 64, scheme=hp.SCHEME_INERTIAL, precision="f32", math_mode=mode)
        dom.upload(g32["rough_state"], g32["rough_bed"], g32["rough_manning"])
        dom.set_target_time(1e9)
        dom.step_batch(200)
        d = np.maximum(0, dom.download()[..., 0].astype(np.float64) - g32["rough_bed"])
        r = np.maximum(0, g32["rough_q_state"][..., 0].astype(np.float64) - g32["rough_bed"])
        assert np.sqrt(np.mean((d - r) ** 2)) < 1e-4


def test_inertial_full_size_4096_properties():
    """BASELINE size: mass conservation of the closed basin (shared faces) and agreement of FAST with STRICT."""
    cols = rows = 4096
    st, bed, man = syn.s_dam(cols, rows, levels=(2.0, 1.6))
    outs = []
    for mode in MODES:
        dom = hp.Domain(cols, rows, scheme=hp.SCHEME_INERTIAL, math_mode=mode)
        dom.upload(st, bed, man)
        dom.set_target_time(1e9)
        dom.step_batch(300)
        outs.append(dom.download()[..., 0].copy())
        dom.close()
    d0 = np.maximum(0, st[..., 0] - bed).sum()
    for z in outs:
        assert abs(np.maximum(0, z - bed).sum() - d0) / d0 < 1e-12
    assert np.abs(outs[0] - outs[1]).max() < 1e-9


def test_cli_runs_the_example(tmp_path):
    """`python -m hipims_mi -c model.xml` (the reference's own command line, main.cpp:464-567) on the example."""
    import subprocess, sys
    from model_dir import make_newcastle
    xml = make_newcastle(tmp_path, duration=60, frequency=30)
    log = os.path.join(str(tmp_path), "run.log")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hipims-ocl_amd"),
                                                        os.environ.get("PYTHONPATH", "")]))
    r = subprocess.run([sys.executable, "-m", "hipims_mi", "-c", xml, "-l", log, "--batch", "50"], capture_output=True,
                       text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Simulation complete: 2 output times" in r.stdout and "SIMULATION PROGRESS" in open(log).read()
    files = sorted(os.listdir(os.path.join(str(tmp_path), "output")))
    assert "depth_30.img" in files and "depth_60.img" in files     # the example asks for format="HFA": Imagine rasters, as GDAL would write
    # same rasters as the library call with the same fixed batch -- f64 pixels, so bit for bit
    from hipims_mi import frontend
    res = frontend.run_model(make_newcastle(tmp_path / "b", duration=60, frequency=30), batch=50, output_format=None)
    img, info = frontend.read_raster(os.path.join(str(tmp_path), "output", "depth_60.img"))
    assert np.array_equal(img, res[-1][1]["depth"]) and info["pixel_size"] == (2.0, 2.0)


@pytest.mark.parametrize("mode", MODES)
def test_fixed_timestep_fixture(mode):
    """desc.dynamic_dt = 0: no CFL reduction, dt = dt_fixed under the sync / early-limit / end-time clips (fixture from
    the reference's TIMESTEP_FIXED program)."""
    g = load_golden("f13_fixed_timestep_f64")
    for name, dt, target, end in (("small", 0.02, 2.5, 1e30), ("clipped", 0.25, 1e9, 11.0)):
        dom = hp.Domain(64, 64, dynamic_dt=False, dt_fixed=dt, dt_initial=dt, t_end=end, math_mode=mode)
        dom.upload(g["state"], g["bed"], g["manning"])
        dom.set_target_time(target)
        tr = dom.run(160)
        assert np.array_equal(tr, g[f"{name}_dt"])                       # no arithmetic on the way to dt: exact in both modes
        sc = dom.read_scalars()
        assert sc["time"] == float(g[f"{name}_t"]) and sc["batch_skipped"] == int(g[f"{name}_skipped"])
        dg = np.maximum(0, dom.download()[..., 0] - g["bed"]); dr = np.maximum(0, g[f"{name}_state"][..., 0] - g["bed"])
        assert np.sqrt(np.mean((dg - dr) ** 2)) < 1e-9 and np.abs(dg - dr).max() < 1e-7
        if name == "small":
            dom.set_target_time(4.0)
            dom.update_timestep()
            assert np.array_equal(dom.run(100), g["resume_dt"]) and dom.read_scalars()["time"] == 4.0
            assert np.abs(dom.download()[..., 0] - g["resume_state"][..., 0]).max() < 1e-7
        dom.close()


@pytest.mark.parametrize("scheme", [hp.SCHEME_GODUNOV, hp.SCHEME_MUSCL_HANCOCK])
def test_strict_bit_identity_at_scale(scheme):
    """1024 x 1024 wet/dry rough terrain, 300 iterations, friction off (no transcendental): every tile shape, XCD band
    and strip boundary of a million-cell grid must reproduce the oracle bit for bit, including the dt sequence."""
    cols = rows = 1024
    st, bed, man = syn.s_rough(cols, rows, manning=None)
    quirks = oracle.QUIRKS_REFERENCE & ~(oracle.Q6_MUSCL_SERIAL if scheme == hp.SCHEME_MUSCL_HANCOCK else 0)
    ref = oracle.OracleSim(cols, rows, scheme=scheme, friction=False, quirks=quirks, threads=min(16, os.cpu_count() or 1))
    dom = hp.Domain(cols, rows, scheme=scheme, friction=False, math_mode=hp.MATH_STRICT)
    for s in (ref, dom):
        s.upload(st, bed, man)
    dom.set_target_time(1e9); ref.set_target(1e9)
    ref.run(300); dom.step_batch(300)
    assert np.array_equal(dom.download(), ref.download())
    sc, sr = dom.read_scalars(), ref.scalars()
    assert sc["time"] == sr["t"] and sc["timestep"] == sr["dt"]
    dom.close()


@pytest.mark.parametrize("workload", ["s-rough", "s-dam-dry"])
@pytest.mark.parametrize("scheme", [hp.SCHEME_GODUNOV, hp.SCHEME_MUSCL_HANCOCK])
def test_fast_and_strict_against_the_oracle_at_scale_with_friction(scheme, workload):
    """VERDICT r03 (weak #9): FAST has wave-uniform short cuts of its own (hp_math.hpp: face_solve's all-wet / subcritical
    paths, the straight-line friction) that STRICT never executes, and was only ever compared with the oracle on grids of a
    few thousand cells (at 4096^2 against the cross-check kernel, which shares its arithmetic).  Here: 1024 x 1024, friction
    ON, 250 iterations, S-ROUGH (every tile on the general path) and S-DAM-DRY (a wet/dry front through still water and dry
    land: every wave-uniform short cut and its fall-back side by side) -- ONE oracle run, STRICT bit for bit, FAST to
    north_star's tolerance: depth RMSE < 1e-9 m, max < 1e-7 m, elapsed time to 1e-12 relative."""
    cols = rows = 1024
    if workload == "s-rough":
        st, bed, man = syn.s_rough(cols, rows, manning=None)
    else:
        st, bed, man = syn.s_dam(cols, rows, wet_right=False)
    quirks = oracle.QUIRKS_REFERENCE & ~(oracle.Q6_MUSCL_SERIAL if scheme == hp.SCHEME_MUSCL_HANCOCK else 0)
    ref = oracle.OracleSim(cols, rows, scheme=scheme, quirks=quirks, threads=min(16, os.cpu_count() or 1))
    ref.upload(st, bed, man); ref.set_target(1e9)
    ref.run(250)
    want, sr = ref.download(), ref.scalars()
    for mode in (hp.MATH_STRICT, hp.MATH_FAST):
        dom = hp.Domain(cols, rows, scheme=scheme, math_mode=mode)
        dom.upload(st, bed, man); dom.set_target_time(1e9)
        dom.step_batch(250)
        got, sc = dom.download(), dom.read_scalars()
        dom.close()
        if mode == hp.MATH_STRICT:
            assert np.array_equal(got, want) and sc["time"] == sr["t"] and sc["timestep"] == sr["dt"]
            continue
        dg, dr = np.maximum(0, got[..., 0] - bed), np.maximum(0, want[..., 0] - bed)
        rmse, mx = float(np.sqrt(np.mean((dg - dr) ** 2))), float(np.abs(dg - dr).max())
        record("fast_at_scale_1024", scheme=int(scheme), workload=workload, rmse=rmse, max=mx,
               time_rel=abs(sc["time"] - sr["t"]) / sr["t"])
        assert rmse < 1e-9 and mx < 1e-7, (rmse, mx)
        assert abs(sc["time"] - sr["t"]) <= 1e-12 * sr["t"]


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("scheme,name,quirks", [
    (hp.SCHEME_GODUNOV, "god", hp.QUIRKS_REFERENCE), (hp.SCHEME_MUSCL_HANCOCK, "mch", hp.QUIRKS_REFERENCE),
    (hp.SCHEME_MUSCL_HANCOCK, "mchnone", hp.QUIRKS_REFERENCE & ~hp.QUIRK_MUSCL_NEIGHBOUR_Y_IS_BED),
    (hp.SCHEME_INERTIAL, "ine", hp.QUIRKS_REFERENCE)])
def test_disabled_cells_fixture(scheme, name, quirks, mode):
    """Disabled cells (mask-style nulls: Zmax = -9999 over an ordinary bed; Z == -9999; DEM-nodata nulls; a live cell on a
    -9998.5 bed) against the fixture from the reference's kernels: STRICT bit for bit -- every cell, every timestep --, FAST
    within the stated tolerance with the nulls carried unchanged.  `mch` is the reference's DEFAULT MUSCL-Hancock
    configuration (kCachePrediction, CSchemeMUSCLHancock.cpp:46; fixture from mch_1st_cachePrediction run as real
    work-groups): the predictor's first-order test reads the neighbours' BED (CLSchemeMUSCLHancock.clc:201, :232-239,
    :325-330), so cells next to mask-style nulls stay second order; `mchnone` is mch_1st_cacheNone (neighbours' Zmax),
    HP_QUIRK_MUSCL_NEIGHBOUR_Y_IS_BED off."""
    g = load_golden("f15_disabled_cells_f64")
    dom = hp.Domain(64, 48, scheme=scheme, math_mode=mode, quirks=quirks)
    dom.upload(g["state"], g["bed"], g["manning"])
    dom.set_target_time(2.5)
    dom.step_batch(150)
    out, ref, dis = dom.download(), g[f"{name}_state"], g["disabled"]
    assert np.array_equal(out[dis], g["state"][dis])
    if mode == hp.MATH_STRICT:
        assert np.array_equal(out, ref)
        assert dom.read_scalars()["time"] == float(g[f"{name}_t"])
    live = ~dis & (g["state"][..., 0] != -9999.0)
    dg = np.maximum(0, out[..., 0] - g["bed"])[live]; dr = np.maximum(0, ref[..., 0] - g["bed"])[live]
    assert np.sqrt(np.mean((dg - dr) ** 2)) < 1e-9 and np.abs(dg - dr).max() < 1e-7
    assert abs(dom.read_scalars()["time"] - float(g[f"{name}_t"])) <= 1e-12 * float(g[f"{name}_t"])
    dom.close()


def test_muscl_variants_differ_next_to_masked_cells():
    """The two predictor variants are different schemes next to nulls: the engine's two settings reproduce the fixture's two
    runs and not each other's (guards against the quirk flag being ignored)."""
    g = load_golden("f15_disabled_cells_f64")
    outs = {}
    for name, quirks in (("mch", hp.QUIRKS_REFERENCE), ("mchnone", hp.QUIRKS_REFERENCE & ~hp.QUIRK_MUSCL_NEIGHBOUR_Y_IS_BED)):
        dom = hp.Domain(64, 48, scheme=hp.SCHEME_MUSCL_HANCOCK, math_mode=hp.MATH_STRICT, quirks=quirks)
        dom.upload(g["state"], g["bed"], g["manning"]); dom.set_target_time(2.5); dom.step_batch(150)
        outs[name] = dom.download(); dom.close()
    assert np.array_equal(outs["mch"], g["mch_state"]) and np.array_equal(outs["mchnone"], g["mchnone_state"])
    assert not np.array_equal(outs["mch"], outs["mchnone"])


@pytest.mark.parametrize("definition", [hp.GRIDDED_MASS_FLUX, hp.GRIDDED_RAIN_INTENSITY])
def test_gridded_definitions_and_series_end(definition):
    """Gridded mass flux (rate / cell area) and what happens past the end of the series: the reference indexes one
    slice beyond its buffer there (CLBoundaries.clc:229); oracle and engine both hold the last slice instead."""
    st, bed, man = syn.s_rough(72, 56, seed=3, manning=None, pool_level=-10.0, amplitude=0.2)
    st[..., 2:] = 0
    scale = 0.02 if definition == hp.GRIDDED_MASS_FLUX else 100.0
    grids = np.random.default_rng(4).uniform(0, scale, (3, 5, 6))
    dom, ref = make_pair(72, 56, st, bed, man, dx=2.0)
    for s in (dom, ref):
        s.add_gridded(definition, grids, 32.0, 0.0, 0.0, 4.0)           # series ends at t = 12 s
    dom.set_target_time(1e9); ref.set_target(1e9)
    ref.run(400); dom.step_batch(400)
    assert ref.scalars()["t"] > 20.0                                    # well past the end of the series
    compare(dom, ref)
    assert (ref.download()[..., 0] - bed).max() > 1e-4


@pytest.mark.parametrize("scheme", [hp.SCHEME_GODUNOV, hp.SCHEME_MUSCL_HANCOCK, hp.SCHEME_INERTIAL])
def test_transpose_equivariance(scheme):
    """x is the lane direction of the kernels and y the direction they march in -- two very different code paths for the
    same physics.  Swapping the axes of the input (and qx <-> qy) must swap the axes of the result, up to the rounding
    of a differently ordered flux sum."""
    cols, rows = 150, 97
    st, bed, man = syn.s_rough(cols, rows, manning=None)
    if scheme == hp.SCHEME_INERTIAL:
        st[..., 2:] = 0
    a = hp.Domain(cols, rows, scheme=scheme)
    a.upload(st, bed, man)
    stT = np.ascontiguousarray(np.transpose(st, (1, 0, 2))[..., [0, 1, 3, 2]])
    b = hp.Domain(rows, cols, scheme=scheme)
    b.upload(stT, np.ascontiguousarray(bed.T), np.ascontiguousarray(man.T))
    for d in (a, b):
        d.set_target_time(1e9)
        d.step_batch(120)
    ra, rb = a.download(), np.transpose(b.download(), (1, 0, 2))[..., [0, 1, 3, 2]]
    assert abs(a.read_scalars()["time"] - b.read_scalars()["time"]) < 1e-10
    assert np.abs(ra - rb).max() < 1e-9, np.abs(ra - rb).max()
    assert np.abs(ra[..., 2]).max() > 1e-3                               # something actually moved
    a.close(); b.close()


@pytest.mark.parametrize("scheme", [hp.SCHEME_GODUNOV, hp.SCHEME_MUSCL_HANCOCK])
def test_sloshing_bowl_analytic(scheme):
    """Thacker's planar surface in a parabolic bowl (the reference's TestSloshingBowl.js case): the HIP kernels against
    the ANALYTIC solution after a quarter and a full period -- a check that does not pass through the oracle."""
    n, dx = 200, 40.0
    bed, fsl, state, period = syn.sloshing_bowl(n, dx)
    dom = hp.Domain(n, n, dx=dx, scheme=scheme, friction=False)
    dom.upload(state(0.0), bed, np.zeros((n, n)))
    for frac, tol in ((0.25, 0.25), (1.0, 0.8)):
        dom.set_target_time(frac * period)
        if dom.read_scalars()["timestep"] <= 0:
            dom.update_timestep()
        while frac * period - dom.read_scalars()["time"] > 1e-6:
            dom.step_batch(100)
        out, ref = dom.download(), fsl(frac * period)
        wet = (ref - bed) > 0.05
        assert np.sqrt(np.mean((out[..., 0] - ref)[wet] ** 2)) < tol, (scheme, frac)
        if frac == 0.25:
            deep = (out[..., 0] - bed) > 1.0
            assert abs((out[..., 2][deep] / (out[..., 0] - bed)[deep]).mean() - 5.0) < 0.1
    dom.close()


def test_emerging_bed_front_analytic():
    """The reference's TestDamBreakEmergingBed case on the GPU: same front as the oracle (to a cell), bounded by the
    analytic position."""
    st, bed, xs, front = syn.emerging_bed_dam_break()
    dom, ref = make_pair(st.shape[1], st.shape[0], st, bed, np.zeros(bed.shape), dx=0.05, friction=False)
    for s, target, upd in ((dom, dom.set_target_time, dom.update_timestep), (ref, ref.set_target, ref.update_timestep)):
        target(1.0)
    while 1.0 - dom.read_scalars()["time"] > 1e-9:
        dom.step_batch(20)
    while 1.0 - ref.scalars()["t"] > 1e-9:
        ref.run(20)
    xg = xs[(dom.download()[4, :, 0] - bed[4]) > 1e-3].max()
    xr = xs[(ref.download()[4, :, 0] - bed[4]) > 1e-3].max()
    assert abs(xg - xr) <= 0.05 + 1e-12 and 0.6 * front(1.0) < xg < front(1.0)
    compare(dom, ref)
