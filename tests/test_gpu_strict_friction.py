"""STRICT arithmetic WITH friction is bit-identical to the oracle and to the fixtures the reference's own kernels produced.

Round 2 held STRICT-with-friction to tolerances because pow(h, 1/3) (Schemes/CLFriction.clc:43) came from three different
math libraries (glibc on the oracle side, the ROCm device library here, whatever the OpenCL platform ships for the
reference).  Since round 3 the oracle, the reference build's `pow` and the STRICT HIP kernels share ONE routine
(hipims-ocl_amd/csrc/hp_crmath.h: correctly rounded cube root from IEEE basic operations; tests/test_crmath.py), so
every comparison here is np.array_equal -- states, elapsed time and the timestep left in the scalar block."""
import os

import numpy as np
import pytest

import hipims_mi as hp
import oracle
from conftest import load_golden, record
from hipims_mi import synthetic as syn

pytestmark = pytest.mark.gpu

GOD, MCH, INE = hp.SCHEME_GODUNOV, hp.SCHEME_MUSCL_HANCOCK, hp.SCHEME_INERTIAL


def oracle_for(cols, rows, scheme, precision, **kw):
    q = oracle.QUIRKS_REFERENCE & ~(oracle.Q6_MUSCL_SERIAL if scheme == MCH else 0)      # snapshot order: see DESIGN 5
    return oracle.OracleSim(cols, rows, scheme=scheme, precision=precision, quirks=q, **kw)


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("scheme", [GOD, MCH, INE])
@pytest.mark.parametrize("kernel", [hp.KERNEL_AUTO, hp.KERNEL_BASIC])
def test_strict_with_friction_equals_the_oracle_bit_for_bit(scheme, precision, kernel):
    if kernel == hp.KERNEL_BASIC and scheme != GOD:
        pytest.skip("the cross-check kernel exists for the Godunov scheme only")
    real = np.float64 if precision == "f64" else np.float32
    st, bed, man = syn.s_rough(72, 40, dtype=real, manning=None)        # spatially varying Manning n, wet/dry, moving water
    ref = oracle_for(72, 40, scheme, precision)
    dom = hp.Domain(72, 40, scheme=scheme, precision=precision, math_mode=hp.MATH_STRICT, kernel=kernel)
    for s in (ref, dom):
        s.upload(st, bed, man)
    dom.set_target_time(1.5); ref.set_target(1.5)                       # a sync point inside the run
    tr_ref, tr_gpu = ref.run(200), dom.run(200)
    assert np.array_equal(tr_gpu, tr_ref)
    assert np.array_equal(dom.download(), ref.download())
    sc, sr = dom.read_scalars(), ref.scalars()
    assert sc["time"] == sr["t"] and sc["timestep"] == sr["dt"] and sc["batch_skipped"] == sr["batch_skipped"]
    moving = np.abs(dom.download()[..., 2:]).max()
    assert moving > 1e-3                                                 # friction really acted on something
    dom.close()


def test_strict_reproduces_the_reference_kernels_trajectory_fixture():
    """F6: 200 iterations of the reference's OWN Godunov program (friction on, rough bed, Manning array) -- state after the
    first and the 200th iteration and every timestep in between, bit for bit."""
    g = load_golden("f6_f7_trajectories_f64")
    st, bed, man = g["rough_state"], g["rough_bed"], g["rough_manning"]
    for key, quirks in (("god_q", hp.QUIRKS_REFERENCE), ("god_noq1", hp.QUIRKS_REFERENCE & ~hp.QUIRK_CFL_READS_PRIMARY)):
        dom = hp.Domain(64, 64, math_mode=hp.MATH_STRICT, quirks=quirks)
        dom.upload(st, bed, man)
        dom.set_target_time(1e9)
        tr = [dom.read_scalars()["timestep"]]
        dom.step_batch(1)
        if f"{key}_state1" in g.files:
            assert np.array_equal(dom.download(), g[f"{key}_state1"])
        for _ in range(199):
            tr.append(dom.read_scalars()["timestep"])
            dom.step_batch(1)
        assert np.array_equal(np.array(tr), g[f"{key}_dt"])
        assert np.array_equal(dom.download(), g[f"{key}_state200"])
        assert dom.read_scalars()["time"] == float(g[f"{key}_t"])
        dom.close()


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_strict_rain_fixture_is_bit_identical(precision):
    """F9: uniform rain + loss and gridded rain on initially dry terrain through the reference's boundary + flux kernels."""
    g = load_golden("f9_rain_" + precision)
    rows, cols = g["bed"].shape
    for name in ("uniform", "gridded"):
        dom = hp.Domain(cols, rows, precision=precision, math_mode=hp.MATH_STRICT)
        dom.upload(g["state"], g["bed"], g["manning"])
        if name == "uniform":
            dom.add_uniform(hp.UNIFORM_RAIN_INTENSITY, g["series"], 3600.0, 10800.0)
            dom.add_uniform(hp.UNIFORM_LOSS_RATE, g["loss"], 10800.0, 10800.0)
        else:
            dom.add_gridded(hp.GRIDDED_RAIN_INTENSITY, g["grids"], 10.0, 0.0, 0.0, 20.0)
        dom.set_target_time(1e9)
        dom.step_batch(len(g[f"{name}_dt"]))
        assert np.array_equal(dom.download(), g[f"{name}_state"]), name
        assert dom.read_scalars()["time"] == float(g[f"{name}_t"])
        dom.close()


def test_config_c1_strict_is_the_reference_run(tmp_path):
    """BASELINE config #1 (the reference's example model: its own DEM, rain 70 mm/h + drainage 12 mm/h, 900 iterations):
    north_star asks for depth RMSE < 1e-9 m against the reference run.  STRICT delivers the reference kernels' result
    itself: level and discharges of every cell and the elapsed time are equal bit for bit (RMSE = 0)."""
    from hipims_mi import frontend
    from model_dir import make_newcastle
    g = load_golden("f10_newcastle_f64")
    cfg = frontend.parse_configuration(make_newcastle(tmp_path))
    st, bed, man, res = frontend.build_domain(cfg)
    dom = hp.Domain(342, 195, dx=res, t_end=cfg.duration, math_mode=hp.MATH_STRICT)
    dom.upload(st, bed, man)
    frontend.attach_boundaries(cfg, dom, 342)
    dom.set_target_time(1e9)
    dom.step_batch(900)
    out = dom.download()
    depth = np.maximum(0, out[..., 0] - bed)
    rmse = float(np.sqrt(np.mean((depth - np.maximum(0, g["z"] - bed)) ** 2)))
    record("c1_newcastle_900_strict_r03", rmse=rmse, max=float(np.abs(out[..., 0] - g["z"]).max()),
           t=dom.read_scalars()["time"], t_ref=float(g["t"]))
    assert rmse < 1e-9                                                   # north_star's bar
    assert np.array_equal(out[..., 0], g["z"])                           # the fixture keeps the discharges in fp32
    assert np.array_equal(out[..., 2].astype(np.float32), g["qx"]) and np.array_equal(out[..., 3].astype(np.float32), g["qy"])
    assert np.array_equal(np.maximum(0, out[..., 0] - bed).astype(np.float32), g["depth"])
    assert dom.read_scalars()["time"] == float(g["t"])
    dom.close()


@pytest.mark.parametrize("scheme", [GOD, MCH])
def test_strict_with_friction_at_scale(scheme):
    """1024 x 1024 wet/dry rough terrain, Manning array, 200 iterations: every tile shape, XCD band and strip boundary of a
    million-cell grid, friction on."""
    cols = rows = 1024
    st, bed, man = syn.s_rough(cols, rows, manning=None)
    ref = oracle_for(cols, rows, scheme, "f64", threads=min(16, os.cpu_count() or 1))
    dom = hp.Domain(cols, rows, scheme=scheme, math_mode=hp.MATH_STRICT)
    for s in (ref, dom):
        s.upload(st, bed, man)
    dom.set_target_time(1e9); ref.set_target(1e9)
    ref.run(200); dom.step_batch(200)
    assert np.array_equal(dom.download(), ref.download())
    sc, sr = dom.read_scalars(), ref.scalars()
    assert sc["time"] == sr["t"] and sc["timestep"] == sr["dt"]
    dom.close()


def test_strict_cell_boundary_with_critical_depth_is_bit_identical():
    """bdy_Cell's critical depth is the other user of pow(x, 1/3) (CLBoundaries.clc:81): fixture F11's free-depth
    discharge definitions through the reference's kernel, bit for bit."""
    g = load_golden("f11_cell_boundary_f64")
    rows, cols = g["bed"].shape
    for name, dd, qd in (("free_q", 0, 1), ("free_volume", 0, 3)):
        dom = hp.Domain(cols, rows, math_mode=hp.MATH_STRICT)
        dom.upload(g["state"], g["bed"], g["manning"])
        dom.add_cell(dd, qd, g["cells"], g["series"], 5.0, 20.0)
        dom.set_target_time(1e9)
        dom.step_batch(300)
        assert np.array_equal(dom.download(), g[f"{name}_state"]), name
        dom.close()


@pytest.mark.parametrize("dx", [1.5, 3.0, 0.7, 2.0, 0.25])
@pytest.mark.parametrize("scheme", [hp.SCHEME_GODUNOV, hp.SCHEME_MUSCL_HANCOCK])
def test_strict_cell_sizes_that_are_and_are_not_powers_of_two(scheme, dx):
    """Round 4: where dx is a power of two STRICT multiplies by the exact 1 / dx instead of dividing by dx (x / 2^k and x * 2^-k
    are the correctly rounded value of the same real number: hp_math.hpp godunov_update_impl / muscl_predict_impl); every other
    dx keeps the divisions.  Both against the oracle, bit for bit (the seeded fuzz only draws 0.5 / 1 / 2 m cells)."""
    cols, rows = 130, 75
    st, bed, man = syn.s_rough(cols, rows, manning=None, seed=91)
    quirks = oracle.QUIRKS_REFERENCE & ~(oracle.Q6_MUSCL_SERIAL if scheme == hp.SCHEME_MUSCL_HANCOCK else 0)
    ref = oracle.OracleSim(cols, rows, dx=dx, scheme=scheme, quirks=quirks)
    dom = hp.Domain(cols, rows, dx=dx, scheme=scheme, math_mode=hp.MATH_STRICT)
    for s in (ref, dom):
        s.upload(st, bed, man)
    dom.set_target_time(1e9); ref.set_target(1e9)
    assert np.array_equal(dom.run(120), ref.run(120))
    assert np.array_equal(dom.download(), ref.download())
    assert dom.read_scalars()["time"] == ref.scalars()["t"]
    dom.close()


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("quirks_off", [0, hp.QUIRK_CFL_READS_PRIMARY])
def test_strict_still_water_rows_keep_the_reference_bits(precision, quirks_off):
    """K1's STRICT flavour does not solve the faces of a row whose cells, and the rows north and south of it, hold ONE wet state
    at rest (hp_kernels.hpp: STILL_SKIP).  Everything such a row still owes the reference is checked here against the oracle, bit
    for bit, on a grid wide enough for whole wavefronts (62 columns) of still water: a lake at rest beside a dam break that
    eats into it; Zmax below Z in part of the lake (it has to follow, CLSchemeGodunov.clc:375-376), Zmax = -9999 and a NODATA
    cell in the lake (never updated, not priced); a deeper second lake whose wave speed the CFL epilogue must find although only
    still rows hold it; and a step in the bed under still water (equal levels, different beds: NOT still)."""
    real = np.float64 if precision == "f64" else np.float32
    cols, rows = 400, 200
    bed = np.zeros((rows, cols), real)
    bed[0, :] = bed[-1, :] = bed[:, 0] = bed[:, -1] = 9999.9
    z = np.full((rows, cols), 2.0, real)
    z[:, 330:] = 0.5                                                      # the dam: everything east of column 330 is shallow
    z[100:, :326] = 5.0; bed[100:, :326] = 1.0                            # a second, deeper lake on a higher bed (its own still rows)
    bed[20:30, 130:160] = 0.25                                            # a bed step under the first lake: same level, other bed
    zmax = z.copy()
    zmax[5:15, 10:300] = 1.0                                              # has to follow Z up to 2.0
    zmax[16:18, 10:300] = -9999.0                                         # disabled by Zmax
    z[33, 200] = -9999.0                                                  # ... and by Z
    st = np.zeros((rows, cols, 4), real)
    st[..., 0] = np.where(bed > 9000, bed, z); st[..., 1] = np.where(bed > 9000, bed, zmax)
    man = np.full((rows, cols), 0.03, real)
    q = oracle.QUIRKS_REFERENCE & ~oracle.quirks_from_engine(quirks_off)
    ref = oracle.OracleSim(cols, rows, scheme=GOD, precision=precision, quirks=q)
    dom = hp.Domain(cols, rows, scheme=GOD, precision=precision, math_mode=hp.MATH_STRICT, quirks=hp.QUIRKS_REFERENCE & ~quirks_off)
    for s in (ref, dom):
        s.upload(st, bed, man)
    dom.set_target_time(1e9); ref.set_target(1e9)
    for n in (1, 2, 57):                                                  # batches cut at odd places (a disturbance travels one cell a step)
        assert np.array_equal(dom.run(n), ref.run(n))
        assert np.array_equal(dom.download(), ref.download())
    out = dom.download()
    assert np.all(out[5:15, 10:250, 1] == real(2.0)) and np.all(out[5:15, 10:250, 0] == real(2.0))   # Zmax followed, nothing else moved
    assert np.all(out[16:18, 10:300, 1] == real(-9999.0))
    assert np.abs(out[..., 2]).max() > 0.1                               # the dam really broke
    assert np.all(out[165:195, 10:250, 2:] == 0) and np.all(out[165:195, 10:250, 0] == real(5.0))   # ... and the far lake is still still
    sc, sr = dom.read_scalars(), ref.scalars()
    assert sc["time"] == sr["t"] and sc["timestep"] == sr["dt"]
    dom.close()
