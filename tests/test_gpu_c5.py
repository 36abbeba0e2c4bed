"""Config C5's path on the HIP engine (fp32 + spatially varying rainfall), the default (FAST) arithmetic in fp32, the
lake-at-rest C-property on the GPU kernels and cell boundaries on the never-written edge ring.  GPU only.

fp32 tolerance: depth RMSE < 1e-4 m against the oracle / the fixtures produced by the reference's own
kernels built with `typedef float cl_double` (COCLProgram.cpp:394-399); bit-identity where two HIP runs must agree.
"""
import numpy as np
import pytest

import hipims_mi as hp
import oracle
from conftest import load_golden, record
from hipims_mi import synthetic as syn

pytestmark = pytest.mark.gpu

MODES = [hp.MATH_FAST, hp.MATH_STRICT]
MODE_NAME = {hp.MATH_FAST: "fast", hp.MATH_STRICT: "strict"}


def depth_of(state, bed):
    return np.maximum(0.0, state[..., 0].astype(np.float64) - bed.astype(np.float64))


def rmse_max(a, b):
    return float(np.sqrt(np.mean((a - b) ** 2))), float(np.abs(a - b).max())


# ---- (i) fp32, default arithmetic, Godunov and MUSCL-Hancock ----
@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("scheme,key,kernel", [(hp.SCHEME_GODUNOV, "god_q", hp.KERNEL_AUTO),
                                                (hp.SCHEME_GODUNOV, "god_q", hp.KERNEL_BASIC),
                                                (hp.SCHEME_MUSCL_HANCOCK, "mch_q", hp.KERNEL_AUTO)])
def test_fp32_rough_bed_vs_oracle_and_fixture(scheme, key, kernel, mode):
    g = load_golden("f6_f7_trajectories_f32")
    st, bed, man = g["rough_state"], g["rough_bed"], g["rough_manning"]
    oq = oracle.QUIRKS_REFERENCE & ~(oracle.Q6_MUSCL_SERIAL if scheme == hp.SCHEME_MUSCL_HANCOCK else 0)
    ref = oracle.OracleSim(64, 64, scheme=scheme, precision="f32", quirks=oq)
    dom = hp.Domain(64, 64, scheme=scheme, precision="f32", math_mode=mode, kernel=kernel)
    for s in (ref, dom):
        s.upload(st, bed, man)
    dom.set_target_time(1e9); ref.set_target(1e9)
    ref.run(200); dom.step_batch(200)
    out = dom.download()
    assert np.isfinite(out).all()
    r, m = rmse_max(depth_of(out, bed), depth_of(ref.download(), bed))
    t_gpu, t_ref = dom.read_scalars()["time"], ref.scalars()["t"]
    record("fp32_rough64_200", scheme=scheme, kernel=kernel, mode=MODE_NAME[mode], rmse=r, max=m, t_gpu=t_gpu, t_ref=t_ref)
    assert abs(t_gpu - t_ref) <= 1e-4 * t_ref
    if scheme == hp.SCHEME_MUSCL_HANCOCK:
        # MUSCL-Hancock on wet/dry terrain is chaotic at fp32 resolution (dry thresholds of 1e-10 m against a level ulp
        # of 3e-8 m): the reference's own fp32 program built strict and as shipped (-cl-mad-enable, COCLProgram.cpp:73)
        # parts by RMSE 2.8e-4 m / max 1.2e-2 m on this very case (fixture f6_f7_bracket_f32_mad).  The engine is held
        # to that bracket (STRICT, which rounds like the strict build, to the 1e-4 of SURVEY 8d).
        gm = load_golden("f6_f7_bracket_f32_mad")
        br, bm = rmse_max(depth_of(g["mch_q_state200"], bed), depth_of(gm["mch_q_state200"], bed))
        assert 1e-4 < br < 1e-3                                        # the reference's own spread, as recorded
        limit = 1e-4 if mode == hp.MATH_STRICT else 1.5 * br
        assert r < limit and m < (1.5 * bm if mode == hp.MATH_FAST else 1e-2), (r, m, br, bm)
    else:
        assert r < 1e-4, (r, m)
    if scheme == hp.SCHEME_GODUNOV:
        # the fixture is what the reference's fp32 program produced (MUSCL's is order dependent on rough terrain, Q6)
        r2, m2 = rmse_max(depth_of(out, bed), depth_of(g[f"{key}_state200"], bed))
        assert r2 < 1e-4, (r2, m2)
        assert abs(t_gpu - float(g[f"{key}_t"])) <= 1e-4 * float(g[f"{key}_t"])
    dom.close()


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("scheme,key", [(hp.SCHEME_GODUNOV, "dam_god"), (hp.SCHEME_GODUNOV, "damdry_god"),
                                        (hp.SCHEME_MUSCL_HANCOCK, "dam_mch")])
def test_fp32_dam_break_fixtures(scheme, key, mode):
    g = load_golden("f6_f7_trajectories_f32")
    st, bed, man = syn.s_dam(96, 48, dtype=np.float32, wet_right=not key.startswith("damdry"))
    dom = hp.Domain(96, 48, scheme=scheme, precision="f32", math_mode=mode)
    dom.upload(st, bed, man)
    dom.set_target_time(1e9)
    dom.step_batch(150)
    out = dom.download()
    r, m = rmse_max(depth_of(out, bed), depth_of(g[f"{key}_state150"], bed))
    t_ref = float(g[f"{key}_dt"].astype(np.float64).sum())
    record("fp32_dam96x48_150", key=key, mode=MODE_NAME[mode], rmse=r, max=m, t_gpu=dom.read_scalars()["time"], t_ref=t_ref)
    assert np.isfinite(out).all() and r < 1e-4, (r, m)                  # on depths of 1-10 m
    assert abs(dom.read_scalars()["time"] - t_ref) <= 2e-4 * t_ref
    dom.close()


# ---- (ii) fp32 rain: uniform rain + loss, gridded rain (fixture F9 from the reference's fp32 kernels) ----
@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("kernel", [hp.KERNEL_AUTO, hp.KERNEL_BASIC])
def test_fp32_rain_fixture(kernel, mode):
    g = load_golden("f9_rain_f32")
    rows, cols = g["bed"].shape
    for name in ("uniform", "gridded"):
        dom = hp.Domain(cols, rows, precision="f32", kernel=kernel, math_mode=mode)
        dom.upload(g["state"], g["bed"], g["manning"])
        if name == "uniform":
            dom.add_uniform(hp.UNIFORM_RAIN_INTENSITY, g["series"], 3600.0, 10800.0)
            dom.add_uniform(hp.UNIFORM_LOSS_RATE, g["loss"], 10800.0, 10800.0)
        else:
            dom.add_gridded(hp.GRIDDED_RAIN_INTENSITY, g["grids"], 10.0, 0.0, 0.0, 20.0)
        dom.set_target_time(1e9)
        dom.step_batch(420)
        out = dom.download()
        dg, dr = depth_of(out, g["bed"]), depth_of(g[f"{name}_state"], g["bed"])
        r, m = rmse_max(dg, dr)
        t_gpu, t_ref = dom.read_scalars()["time"], float(g[f"{name}_t"])
        record("fp32_rain_f9", boundary=name, kernel=kernel, mode=MODE_NAME[mode], rmse=r, max=m, mean_depth=float(dr.mean()),
               t_gpu=t_gpu, t_ref=t_ref)
        assert dr.max() > 1e-4 and np.isfinite(out).all()
        assert r < 1e-4 and r < 0.05 * dr.mean(), (name, r, m, dr.mean())
        assert abs(dg.sum() - dr.sum()) < 1e-2 * dr.sum()                # the rained volume itself
        assert abs(t_gpu - t_ref) <= 1e-3 * t_ref
        dom.close()


def hydro_applied(n_iterations, dt=np.float32(0.1), dt0=np.float32(0.001)):
    """Sum of the hydrological timesteps the boundary kernels see over n iterations of constant dt (fp32): the gate of
    bdy_Gridded (t_hydro >= 1 s, CLBoundaries.clc:224-225) against the accumulator of tst_Advance_Normal
    (reset to dt when > 1 s, else += dt; CLDynamicTimestep.clc:61-66).  The first iteration runs with dt0."""
    t_hydro, total, step = np.float32(0), 0.0, np.float32(dt0)
    for _ in range(n_iterations):
        if t_hydro >= np.float32(1.0):
            total += float(t_hydro)
        t_hydro = step if t_hydro > np.float32(1.0) else np.float32(t_hydro + step)
        step = np.float32(dt)
    return total


# ---- (iii) config C5 at BASELINE.json's full size: 8192 x 8192 fp32, gridded rain on dry terrain ----
def test_s_rain_8192_fp32_full_size_properties():
    """No oracle run at 67 M cells: the closed basin must hold exactly the water that rained (to fp32's resolution of a
    millimetre film on a 10 m level: the reference's FSL formulation, `docs/papers/dam-break-cf`), stay finite, keep
    Zmax >= Z, never go below the bed.  While t < 60 s every timestep is the early limit 0.1 s (thin films: the CFL
    step is seconds), so the sequence of hydrological timesteps is known without running anything."""
    n, dx, its = 8192, 2.0, 250
    st, bed, man, rain = syn.s_rain(n, n, dx=dx, dtype=np.float32)
    dom = hp.Domain(n, n, dx=dx, precision="f32")
    dom.upload(st, bed, man)
    del st, man
    dom.add_gridded(hp.GRIDDED_RAIN_INTENSITY, rain["grids"], rain["resolution"], rain["off_x"], rain["off_y"], rain["interval"])
    dom.set_target_time(1e9)
    dom.step_batch(its)
    sc = dom.read_scalars()
    assert sc["batch_successful"] == its and abs(sc["time"] - (0.001 + 0.1 * (its - 1))) < 1e-3
    out = dom.download()
    assert np.isfinite(out).all()
    inner = np.s_[1:-1, 1:-1]
    depth = out[..., 0].astype(np.float64) - bed.astype(np.float64)
    assert depth[inner].min() >= 0.0 and (out[inner][..., 1] >= out[inner][..., 0]).all()
    # expected volume: every interior cell received rate / 3.6e6 * (sum of hydrological timesteps), slice 0 of the stack
    res = rain["resolution"]
    idx = np.floor(np.arange(n) * dx / res).astype(int)
    rate = rain["grids"][0].astype(np.float64)[np.ix_(idx, idx)]
    expected = (rate[inner] / 3.6e6).sum() * hydro_applied(its)
    got = depth[inner].sum()
    record("s_rain_8192_fp32", iterations=its, t=sc["time"], volume_rel_err=abs(got - expected) / expected,
           mean_depth_mm=1e3 * got / (n - 2) ** 2)
    # fp32 accumulation of 0.4 mm films on a 10 m level costs 2.3e-3; one missed gate opening of the ~22 would be 4.5e-2
    assert abs(got - expected) / expected < 5e-3, (got, expected)
    dom.close()


def test_fp32_strips_with_rain_on_ghost_rows_are_bit_identical():
    """Two strips of one fp32 grid on one GPU (halo overlap on), gridded rain addressed with global rows so that ghost
    rows receive what their owner gives them: bit-identical to the single domain."""
    from test_gpu_strips import run_strips
    cols, rows, steps = 130, 101, 300
    st, bed, man = syn.s_rough(cols, rows, dtype=np.float32, pool_level=-10.0, amplitude=0.2, walls=False)
    st[..., 2:] = 0
    grids = np.random.default_rng(5).uniform(0, 120, (3, 7, 9)).astype(np.float32)
    rain = (grids, 16.0, 0.0, 0.0, 20.0)
    single = hp.Domain(cols, rows, precision="f32")
    single.upload(st, bed, man)
    single.add_gridded(hp.GRIDDED_RAIN_INTENSITY, *rain)
    single.set_target_time(1e9)
    single.step_batch(steps)
    ref = single.download()
    assert (ref[..., 0] - bed).max() > 1e-5
    for nstrips in (2, 3):
        out, sc = run_strips(cols, rows, nstrips, hp.SCHEME_GODUNOV, steps, st, bed, man, rain=rain, overlap=True, precision="f32")
        assert np.array_equal(out, ref)
        assert sc["t"] == single.read_scalars()["time"]


# ---- (iv) lake at rest on the HIP kernels (tools/model-builder/tests/TestLakeAtRest.js:59-70) ----
@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("scheme,kernel", [(hp.SCHEME_GODUNOV, hp.KERNEL_AUTO), (hp.SCHEME_GODUNOV, hp.KERNEL_BASIC),
                                           (hp.SCHEME_MUSCL_HANCOCK, hp.KERNEL_AUTO), (hp.SCHEME_INERTIAL, hp.KERNEL_AUTO)])
def test_lake_at_rest_on_the_gpu(scheme, kernel, mode):
    """Still water over a bumpy bed with dry islands must not move: level unchanged to 0 ulp, discharges exactly zero.
    FAST reorders the pressure and bed-slope terms whose cancellation this property rests on, so it is tested for
    both arithmetic flavours, over several tile widths (islands on tile and wavefront boundaries)."""
    rng = np.random.default_rng(3)
    for cols, rows in ((48, 40), (200, 70)):
        y, x = np.mgrid[0:rows, 0:cols]
        bed = 0.8 * np.exp(-((x - cols / 2) ** 2 + (y - rows / 2) ** 2) / 60.0) + rng.uniform(0, 0.05, (rows, cols))
        bed += 0.7 * np.exp(-((x - 62) ** 2 + (y - 17) ** 2) / 30.0)       # second island astride the 62-column tile edge
        st = np.zeros((rows, cols, 4))
        st[..., 0] = np.maximum(bed, 0.5)
        st[..., 1] = st[..., 0]
        assert ((st[..., 0] - bed) == 0).sum() > 20                        # dry islands present
        dom = hp.Domain(cols, rows, scheme=scheme, kernel=kernel, math_mode=mode)
        dom.upload(st, bed, np.full((rows, cols), 0.03))
        dom.set_target_time(1e9)
        dom.step_batch(100)
        out = dom.download()
        assert np.array_equal(out[..., 0], st[..., 0]), (scheme, kernel, mode, np.abs(out[..., 0] - st[..., 0]).max())
        assert np.abs(out[..., 2:]).max() == 0.0
        assert dom.read_scalars()["time"] > 1.0
        dom.close()


# ---- cell boundaries on the edge ring (never written by the flux kernels, priced by the CFL reduction) ----
@pytest.mark.parametrize("q1", [True, False])
def test_cell_boundary_on_the_edge_ring_drives_the_timestep(q1):
    """bdy_Cell accepts any cell (CBoundaryCell reads arbitrary (x, y) from its map CSV); inflow cells on the domain
    edge are the usual case.  The imposed ring cell carries the fastest wave here, so the dt trace shows whether the
    engine's cached ring maximum follows it (tst_Reduce re-reads every cell each iteration)."""
    cols, rows = 70, 45
    st, bed, man = syn.s_rough(cols, rows, manning=None, walls=False)
    cells = [5 * cols + 0, 6 * cols + 0, (rows - 1) * cols + 30, (rows - 1) * cols + 31]
    series = np.array([[0, 3.0, 12.0, -6.0], [5, 4.0, 25.0, -12.0], [10, 4.0, 25.0, -12.0], [15, 2.0, 5.0, 0.0],
                       [20, 2.0, 5.0, 0.0]])
    quirks = hp.QUIRKS_REFERENCE if q1 else hp.QUIRKS_REFERENCE & ~hp.QUIRK_CFL_READS_PRIMARY
    ref = oracle.OracleSim(cols, rows, quirks=oracle.quirks_from_engine(quirks, muscl_serial=True))
    dom = hp.Domain(cols, rows, quirks=quirks)
    for s in (ref, dom):
        s.upload(st, bed, man)
        s.add_cell(hp.DEPTH_IS_DEPTH, hp.DISCHARGE_IS_DISCHARGE, cells, series, 5.0, 20.0)
    dom.set_target_time(1e9); ref.set_target(1e9)
    tr_ref, tr_gpu = ref.run(150), dom.run(150)
    # the ring cell really is what limits the timestep: without it the trace differs
    plain = oracle.OracleSim(cols, rows, quirks=oracle.quirks_from_engine(quirks, muscl_serial=True))
    plain.upload(st, bed, man); plain.set_target(1e9)
    assert np.abs(plain.run(150) - tr_ref).max() > 1e-3 * tr_ref.max()
    assert np.abs(tr_gpu - tr_ref).max() <= 1e-12 * tr_ref.max(), np.abs(tr_gpu - tr_ref).max()
    r, m = rmse_max(depth_of(dom.download(), bed), depth_of(ref.download(), bed))
    assert r < 1e-9 and m < 1e-7, (r, m)
    dom.close()


# ---- fp32 soak: raw v_rcp_f32 / v_sqrt_f32 in the FAST flavour must not produce infinities on wet/dry fronts ----
@pytest.mark.parametrize("scheme", [hp.SCHEME_GODUNOV, hp.SCHEME_MUSCL_HANCOCK])
def test_fp32_fast_soak_stays_finite_and_conservative(scheme):
    cols, rows, steps = 384, 256, 6000
    st, bed, man = syn.s_dam(cols, rows, dtype=np.float32, wet_right=False)
    # a sloping, bumpy right half so the front runs up and down dry terrain
    y, x = np.mgrid[0:rows, 0:cols]
    bump = (0.002 * np.maximum(0, x - cols // 2) + 0.3 * np.exp(-((x - 280) ** 2 + (y - 128) ** 2) / 400.0)).astype(np.float32)
    bed[1:-1, 1:-1] += bump[1:-1, 1:-1]
    st[..., 0] = np.maximum(st[..., 0], bed); st[..., 1] = st[..., 0]
    st[0] = st[-1] = 0; st[:, 0] = st[:, -1] = 0
    d = hp.Domain(cols, rows, scheme=scheme, precision="f32")
    d.upload(st, bed, man)
    d.set_target_time(1e9)
    d.step_batch(steps)
    out, sc = d.download(), d.read_scalars()
    assert np.isfinite(out).all() and sc["batch_successful"] == steps and np.isfinite(sc["timestep"]) and sc["timestep"] > 0
    inner = np.s_[2:-2, 2:-2]
    depth0, depth = depth_of(st, bed), depth_of(out, bed)
    assert (out[inner][..., 1] >= out[inner][..., 0]).all() and depth.max() < 10.5
    drift = abs(depth[1:-1, 1:-1].sum() - depth0[1:-1, 1:-1].sum()) / depth0[1:-1, 1:-1].sum()
    record("fp32_soak", scheme=scheme, steps=steps, t=sc["time"], volume_drift=drift)
    if scheme == hp.SCHEME_GODUNOV:
        assert drift < 1e-3, drift
    d.close()


# ---- device-side checkpoint (hp_state_save / hp_state_restore): saveCurrentState + rollbackSimulation kept in HBM ----
@pytest.mark.parametrize("scheme", [hp.SCHEME_GODUNOV, hp.SCHEME_MUSCL_HANCOCK, hp.SCHEME_INERTIAL])
@pytest.mark.parametrize("saved_after", [6, 7])          # both ping-pong phases (quirk Q1 prices the primary buffer only)
@pytest.mark.parametrize("rain", [True, False])           # without a boundary every other iteration re-uses the REMEMBERED maximum
def test_state_restore_replays_bit_for_bit(scheme, saved_after, rain):
    st, bed, man = syn.s_rough(192, 96)
    d = hp.Domain(192, 96, scheme=scheme)
    d.upload(st, bed, man)
    if rain:
        d.add_uniform(hp.UNIFORM_RAIN_INTENSITY, [[0.0, 30.0], [3600.0, 30.0]], 3600.0, 3600.0)
    d.set_target_time(1e9)
    d.step_batch(saved_after)
    d.state_save()
    d.step_batch(25)
    first, sc_first = d.download(), d.read_scalars()
    d.step_batch(11)                                      # wander off, odd count: the ping-pong phase differs at restore time
    d.state_restore()
    d.step_batch(25)
    again, sc_again = d.download(), d.read_scalars()
    d.close()
    assert np.array_equal(first, again)
    assert sc_first["time"] == sc_again["time"] and sc_first["timestep"] == sc_again["timestep"]
    assert sc_first["time_hydrological"] == sc_again["time_hydrological"]


def test_state_restore_without_save_is_an_error():
    d = hp.Domain(64, 32)
    with pytest.raises(hp.HipimsError, match="without a saved state"):
        d.state_restore()
    d.close()


# ---- K2 inert rows (still water / dry land skip both face solves and the update): exactness ----
def _inert_mix(cols, rows):
    """Still lake (west), dry sloping land (east), a dam-break front between them, a drop falling into the lake, a dry
    island in it, disabled cells, and rows of moving water: every transition between inert and live rows occurs."""
    y, x = np.mgrid[0:rows, 0:cols].astype(np.float64)
    bed = np.where(x > 0.55 * cols, 0.02 * (x - 0.55 * cols), 0.0)                # flat under the lake, rising land
    bed[(x - 0.2 * cols) ** 2 + (y - 0.5 * rows) ** 2 < 16] = 3.0               # island
    z = np.maximum(bed, np.where(x < 0.55 * cols, 2.0, 0.0))
    z[(x - 0.35 * cols) ** 2 + (y - 0.3 * rows) ** 2 < 9] += 0.25                # the drop
    st = np.zeros((rows, cols, 4)); st[..., 0] = z; st[..., 1] = z
    band = (y > 0.75 * rows) & (x < 0.5 * cols)
    st[..., 2][band] = 0.3                                                        # uniformly moving water: quiet but not inert
    st[5:9, 20:26, 1] = -9999.0                                                   # nulls inside the still lake
    syn._walls(st, bed)
    return st, bed, np.full((rows, cols), 0.03)


@pytest.mark.parametrize("scheme", [hp.SCHEME_MUSCL_HANCOCK, hp.SCHEME_GODUNOV])
@pytest.mark.parametrize("cols,rows", [(200, 96), (130, 70)])
def test_muscl_inert_rows_are_exact_strict_vs_oracle(cols, rows, scheme):
    """K2's inert rows (still water, dry land) and K1's leading dry-land run of a tile, through every transition to live
    rows: STRICT, friction off -> bit for bit against the oracle after each of four batches."""
    st, bed, man = _inert_mix(cols, rows)
    quirks = oracle.QUIRKS_REFERENCE & ~(oracle.Q6_MUSCL_SERIAL if scheme == hp.SCHEME_MUSCL_HANCOCK else 0)
    ref = oracle.OracleSim(cols, rows, scheme=scheme, friction=False, quirks=quirks)
    dom = hp.Domain(cols, rows, scheme=scheme, friction=False, math_mode=hp.MATH_STRICT)
    for s in (ref, dom):
        s.upload(st, bed, man)
    dom.set_target_time(1e9); ref.set_target(1e9)
    for _ in range(4):                                       # the front and the ripples eat into the inert regions
        ref.run(40); dom.step_batch(40)
        assert np.array_equal(dom.download(), ref.download())
        assert dom.read_scalars()["time"] == ref.scalars()["t"]
    dom.close()


def test_muscl_inert_rows_fast_matches_oracle_and_keeps_the_lake_still():
    cols, rows = 260, 128
    st, bed, man = _inert_mix(cols, rows)
    quirks = oracle.QUIRKS_REFERENCE & ~oracle.Q6_MUSCL_SERIAL
    ref = oracle.OracleSim(cols, rows, scheme=hp.SCHEME_MUSCL_HANCOCK, quirks=quirks)
    dom = hp.Domain(cols, rows, scheme=hp.SCHEME_MUSCL_HANCOCK)
    for s in (ref, dom):
        s.upload(st, bed, man)
    dom.set_target_time(1e9); ref.set_target(1e9)
    ref.run(60); dom.step_batch(60)
    out, want = dom.download(), ref.download()
    dg, dr = np.maximum(0, out[..., 0] - bed), np.maximum(0, want[..., 0] - bed)
    assert np.sqrt(np.mean((dg - dr) ** 2)) < 1e-9 and np.abs(dg - dr).max() < 1e-7
    assert abs(dom.read_scalars()["time"] - ref.scalars()["t"]) < 1e-9
    quiet = (want[..., [0, 2, 3]] == st[..., [0, 2, 3]]).all(-1) & (st[..., 0] - bed > 0.1)
    assert quiet.sum() > 3000                                # lake water the ripples have not reached yet ...
    assert np.array_equal(out[quiet][..., [0, 2, 3]], st[quiet][..., [0, 2, 3]])     # ... keeps its bits on the GPU too
    assert np.array_equal(out[5:9, 20:26], st[5:9, 20:26])   # nulls carried through
    dom.close()
