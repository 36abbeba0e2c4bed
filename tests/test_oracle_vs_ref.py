"""The C oracle against the reference's own kernels, live (needs oracle/_ref/*.so, i.e. a container or
GPU box that received the built objects).  Complements the committed fixtures with fresh seeds."""
import numpy as np
import pytest

import oracle
from hipims_mi import synthetic as syn

pytestmark = pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref not built (needs /root/reference)")


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("scheme", [oracle.GODUNOV, oracle.MUSCL, oracle.INERTIAL])
@pytest.mark.parametrize("seed", [1, 2])
def test_random_terrain_trajectory(precision, scheme, seed):
    real = np.float64 if precision == "f64" else np.float32
    st, bed, man = syn.s_rough(40, 33, dtype=real, seed=seed, manning=None, walls=bool(seed % 2))
    if scheme == oracle.INERTIAL and not oracle.have_ref("ine_" + precision):
        pytest.skip("oracle/_ref built before the inertial recipe existed")
    a = oracle.OracleSim(40, 33, scheme=scheme, precision=precision, dx=1.5)
    b = oracle.RefSim(40, 33, scheme=scheme, precision=precision, dx=1.5)
    for s in (a, b):
        s.upload(st, bed, man)
        s.set_target(3.0)                 # reach the sync point inside the run -> negative-dt skips
    ta, tb = a.run(160), b.run(160)
    assert np.array_equal(ta, tb)
    assert np.array_equal(a.download(), b.download(), equal_nan=True)
    sa, sb = a.scalars(), b.scalars()
    assert sa == {k: (type(sa[k])(v)) for k, v in sb.items()}
    assert sa["batch_skipped"] > 0


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("default_variant", [True, False])
def test_muscl_predictor_variants_next_to_nulls(precision, default_variant):
    """The reference's two MUSCL-Hancock predictors, live: its DEFAULT mch_1st_cachePrediction (kCachePrediction,
    CSchemeMUSCLHancock.cpp:46) run as real 16 x 16 work-groups with LDS tile and barrier (oracle/ref_build/shim.cpp) -- the
    neighbours' .y is their BED --, and mch_1st_cacheNone -- their Zmax.  Mask-style nulls (Zmax = -9999 over an ordinary
    bed), DEM-nodata nulls and a live cell on a bed below -9998: the restatement follows each variant bit for bit."""
    real = np.float64 if precision == "f64" else np.float32
    st, bed, man = syn.s_rough(40, 33, dtype=real, seed=4, manning=None)
    rng = np.random.default_rng(41)
    mask = rng.random((33, 40)) < 0.06
    mask[0] = mask[-1] = False; mask[:, 0] = mask[:, -1] = False
    st[mask, 1] = -9999.0
    bed[20:23, 8:11] = -9999.0; st[20:23, 8:11, 0] = -9999.0; st[20:23, 8:11, 1] = -9999.0
    bed[12, 30] = -9998.25; st[12, 30, 0] = -9998.25; st[12, 30, 1] = -9998.25
    quirks = oracle.QUIRKS_REFERENCE if default_variant else oracle.QUIRKS_REFERENCE & ~oracle.Q11_MUSCL_NB_Y_IS_BED
    a = oracle.OracleSim(40, 33, scheme=oracle.MUSCL, precision=precision, quirks=quirks)
    b = oracle.RefSim(40, 33, scheme=oracle.MUSCL, precision=precision, quirks=quirks)
    for s in (a, b):
        s.upload(st, bed, man)
        s.set_target(1e9)
    assert np.array_equal(a.run(60), b.run(60))
    assert np.array_equal(a.download(), b.download(), equal_nan=True)


def test_threads_do_not_change_results():
    st, bed, man = syn.s_rough(64, 48, manning=None)
    outs = []
    for threads in (1, 4):
        s = oracle.OracleSim(64, 48, threads=threads)
        s.upload(st, bed, man)
        s.set_target(1e9)
        s.run(50)
        outs.append(s.download())
    assert np.array_equal(outs[0], outs[1])


def test_muscl_snapshot_equals_serial_when_wet():
    """Quirk Q6 only matters at cells whose neighbours' Zmax crosses VERY_SMALL; an all-wet run is identical."""
    st, bed, man = syn.s_dam(64, 32)
    outs = []
    for quirks in (oracle.QUIRKS_REFERENCE, oracle.QUIRKS_REFERENCE & ~oracle.Q6_MUSCL_SERIAL):
        s = oracle.OracleSim(64, 32, scheme=oracle.MUSCL, quirks=quirks)
        s.upload(st, bed, man)
        s.set_target(1e9)
        s.run(80)
        outs.append(s.download())
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.skipif(not oracle.have_ref("ine_f64"), reason="inertial reference build missing")
@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_inertial_flux_random(precision):
    rng = np.random.default_rng(77)
    of, rf = oracle.OracleFunctions(precision), oracle.RefFunctions(precision)
    for _ in range(3000):
        bu, bd = rng.uniform(-1, 1, 2)
        args = (rng.choice([0.0, 0.03, rng.uniform(0.01, 0.1)]), rng.choice([0.001, 0.05, rng.uniform(1e-4, 1)]),
                rng.choice([0.0, rng.normal(0, 2)]), bu + rng.choice([0, 1e-11, rng.uniform(0, 3)]), bu,
                bd + rng.choice([0, 1e-11, rng.uniform(0, 3)]), bd)
        a, b = of.inertial_flux(*args), rf.inertial_flux(*args)
        assert a == b or (np.isnan(a) and np.isnan(b)), args


@pytest.mark.skipif(not oracle.have_ref("god_f64_fixed"), reason="fixed-timestep reference build missing")
@pytest.mark.parametrize("dt", [0.01, 0.05, 0.3])
def test_fixed_timestep_trajectory(dt):
    st, bed, man = syn.s_rough(40, 33, seed=5, manning=None)
    a = oracle.OracleSim(40, 33, dynamic_dt=False, fixed_dt=dt, dt_initial=dt, end_time=9.0)
    b = oracle.RefSim(40, 33, dynamic_dt=False, fixed_dt=dt, dt_initial=dt, end_time=9.0)
    for s in (a, b):
        s.upload(st, bed, man)
        s.set_target(4.0)
    assert np.array_equal(a.run(90), b.run(90))
    for s in (a, b):
        s.set_target(20.0)
        s.update_timestep()
    assert np.array_equal(a.run(90), b.run(90))
    assert np.array_equal(a.download(), b.download(), equal_nan=True)
    sa, sb = a.scalars(), b.scalars()
    assert sa == {k: (type(sa[k])(v)) for k, v in sb.items()}


@pytest.mark.skipif(not oracle.have_ref("god_f64_nofric"), reason="no-friction reference build missing")
@pytest.mark.parametrize("scheme", [oracle.GODUNOV, oracle.MUSCL])
def test_no_friction_trajectory(scheme):
    st, bed, man = syn.s_rough(40, 33, seed=9, manning=None, walls=False)
    a = oracle.OracleSim(40, 33, scheme=scheme, friction=False)
    b = oracle.RefSim(40, 33, scheme=scheme, friction=False)
    for s in (a, b):
        s.upload(st, bed, man)
        s.set_target(2.0)
    assert np.array_equal(a.run(140), b.run(140))
    assert np.array_equal(a.download(), b.download(), equal_nan=True)


@pytest.mark.parametrize("definition", [oracle.GRIDDED_MASS_FLUX, oracle.GRIDDED_RAIN_ACCUMUL])
def test_gridded_mass_flux_and_accumulation(definition):
    """BOUNDARY_GRIDDED_MASS_FLUX adds rate / (dx dy) per second; RAIN_ACCUMULATION is accepted and ignored by the
    device code (CLBoundaries.clc:237-243): both bit-identical between the restatement and the reference's kernel."""
    st, bed, man = syn.s_rough(40, 33, seed=3, manning=None, pool_level=-10.0, amplitude=0.2)
    st[..., 2:] = 0
    grids = np.random.default_rng(4).uniform(0, 0.02, (4, 5, 6))
    sims = [oracle.OracleSim(40, 33, dx=2.0), oracle.RefSim(40, 33, dx=2.0)]
    for s in sims:
        s.upload(st, bed, man)
        s.add_gridded(definition, grids, 20.0, 0.0, 0.0, 5.0)          # stays inside the 4 x 5 s series
        s.set_target(1e9)
    ta, tb = sims[0].run(150), sims[1].run(150)
    assert np.array_equal(ta, tb) and np.array_equal(sims[0].download(), sims[1].download())
    wet = (sims[0].download()[..., 0] - bed).max()
    assert (wet > 1e-4) == (definition == oracle.GRIDDED_MASS_FLUX) and sims[0].scalars()["t"] < 19.0


def test_uniform_boundary_past_the_end_of_its_series():
    """bdy_Uniform stops applying once t >= TimeseriesLength (CLBoundaries.clc:168-169): rain and loss for 12 s, then
    nothing; piecewise-constant lookup in between.  Bit for bit against the reference's kernel."""
    st, bed, man = syn.s_rough(40, 33, seed=3, manning=None, pool_level=-10.0, amplitude=0.2)
    st[..., 2:] = 0
    series = np.array([[0, 60.0], [4, 20.0], [8, 90.0], [12, 0.0]])
    sims = [oracle.OracleSim(40, 33), oracle.RefSim(40, 33)]
    for s in sims:
        s.upload(st, bed, man)
        s.add_uniform(oracle.UNIFORM_RAIN_INTENSITY, series, 4.0, 12.0)
        s.add_uniform(oracle.UNIFORM_LOSS_RATE, np.array([[0, 5.0], [12, 5.0]]), 12.0, 12.0)
        s.set_target(20.0)
    assert np.array_equal(sims[0].run(400), sims[1].run(400))
    assert np.array_equal(sims[0].download(), sims[1].download()) and sims[0].scalars()["t"] == 20.0
