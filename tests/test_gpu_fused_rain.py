"""Area boundaries fused into the flux kernel (K1 FUSED, round 3): iteration n stores its result with iteration n+1's rain /
loss already added, instead of a separate in-place pass over the grid at the start of iteration n+1
(reference order: Schemes/CSchemeGodunov.cpp:1637-1643; kernels Boundaries/CLBoundaries.clc:130-246).

What must hold, bit for bit: the state after ANY number of iterations is the reference's (the STRICT fixture tests in
test_gpu_strict_friction.py run fused), however the iterations are cut into batches -- a batch's last iteration never
fuses, so that a download between batches sees the reference's buffer -- including runs through a sync point, where the
sign of the next timestep (the uniform kernel's test, CLBoundaries.clc:165-166) is not known in advance and the
stand-alone pass takes over."""
import numpy as np
import pytest

import hipims_mi as hp
import oracle
from conftest import load_golden
from hipims_mi import synthetic as syn

pytestmark = pytest.mark.gpu


def rain_domain(name, precision, mode, g):
    rows, cols = g["bed"].shape
    dom = hp.Domain(cols, rows, precision=precision, math_mode=mode)
    dom.upload(g["state"], g["bed"], g["manning"])
    if name == "uniform":
        dom.add_uniform(hp.UNIFORM_RAIN_INTENSITY, g["series"], 3600.0, 10800.0)
        dom.add_uniform(hp.UNIFORM_LOSS_RATE, g["loss"], 10800.0, 10800.0)
    else:
        dom.add_gridded(hp.GRIDDED_RAIN_INTENSITY, g["grids"], 10.0, 0.0, 0.0, 20.0)
    dom.set_target_time(1e9)
    return dom


@pytest.mark.parametrize("mode", [hp.MATH_FAST, hp.MATH_STRICT])
@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_uniform_boundaries_are_fused_and_batch_cuts_do_not_change_a_bit(precision, mode):
    """F9's uniform rain + loss on initially dry terrain, 420 iterations: one batch, single-iteration batches (never fused)
    and ragged batches with a download after each give the same states and the same time, to the last bit."""
    g = load_golden("f9_rain_" + precision)
    plans = {"one": [420], "single": [1] * 420, "ragged": [3, 1, 2, 50, 7, 1, 1, 100, 64, 191]}
    results, mid = {}, {}
    for key, plan in plans.items():
        dom = rain_domain("uniform", precision, mode, g)
        assert dom.boundaries_fused()
        done = 0
        for n in plan:
            dom.step_batch(n)
            done += n
            if key != "one" and done in (6, 56, 229):
                mid[(key, done)] = dom.download()
        results[key] = (dom.download(), dom.read_scalars())
        dom.close()
    for key in ("single", "ragged"):
        assert np.array_equal(results[key][0], results["one"][0]), key
        assert results[key][1]["time"] == results["one"][1]["time"] and results[key][1]["timestep"] == results["one"][1]["timestep"]
        assert results[key][1]["time_hydrological"] == results["one"][1]["time_hydrological"]
    for done in (6, 56, 229):
        assert np.array_equal(mid[("single", done)], mid[("ragged", done)]), done      # what a download sees between batches
    if mode == hp.MATH_STRICT:                                                           # ... and all of it is the reference's
        assert np.array_equal(results["one"][0], g["uniform_state"])


def test_coarse_rain_grid_is_fused_fine_grid_is_not_and_both_agree():
    """A rain grid of 64+ model cells per grid cell rides in the flux kernel (a tile meets at most two grid rows, a wavefront
    two grid columns); F9's 10 m grid on 1 m cells keeps the stand-alone pass.  Same physics either way: the coarse case
    is checked against the oracle bit for bit (STRICT), batch cuts included, on a grid whose tiles straddle grid rows."""
    cols, rows, dx = 200, 150, 2.0
    rng = np.random.default_rng(5)
    st, bed, man = syn.s_rough(cols, rows, manning=None)
    st[..., 0] = np.maximum(bed, st[..., 0] - 0.6); st[..., 1] = st[..., 0]; st[..., 2:] = 0        # mostly dry terrain, some pools
    st[0] = st[-1] = 0; st[:, 0] = st[:, -1] = 0
    grids = rng.uniform(0.0, 400.0, (4, 3, 4))                                                # 3 x 4 cells of 128 m (64 model cells)
    ref = oracle.OracleSim(cols, rows, dx=dx)
    ref.upload(st, bed, man)
    ref.add_gridded(hp.GRIDDED_RAIN_INTENSITY, grids, 128.0, 0.0, 0.0, 15.0)
    ref.set_target(1e9)
    ref.run(300)
    for plan in ([300], [1, 2, 97, 1, 199]):
        dom = hp.Domain(cols, rows, dx=dx, math_mode=hp.MATH_STRICT)
        dom.upload(st, bed, man)
        dom.add_gridded(hp.GRIDDED_RAIN_INTENSITY, grids, 128.0, 0.0, 0.0, 15.0)
        assert dom.boundaries_fused()
        dom.set_target_time(1e9)
        for n in plan:
            dom.step_batch(n)
        assert np.array_equal(dom.download(), ref.download()), plan
        assert dom.read_scalars()["time"] == ref.scalars()["t"]
        dom.close()
    assert (ref.download()[..., 0] - bed).max() > 1e-3
    g = load_golden("f9_rain_f64")
    fine = rain_domain("gridded", "f64", hp.MATH_STRICT, g)
    assert not fine.boundaries_fused()
    fine.close()


def test_sync_point_inside_a_batch_with_uniform_rain():
    """Through a sync point: the iteration that lands on it, the suspended (skipped) iterations after it and the resumed
    run, all inside batches -- where the next dt's sign is uncertain the kernel declines and the stand-alone pass applies
    the boundaries.  STRICT against the oracle bit for bit, and the hydrological clock with it."""
    cols, rows = 96, 64
    st, bed, man = syn.s_rough(cols, rows, manning=None)
    series = np.array([[0.0, 80.0], [3600.0, 80.0]])
    ref = oracle.OracleSim(cols, rows)
    dom = hp.Domain(cols, rows, math_mode=hp.MATH_STRICT)
    for s in (ref, dom):
        s.upload(st, bed, man)
        s.add_uniform(hp.UNIFORM_RAIN_INTENSITY, series, 3600.0, 3600.0)
    assert dom.boundaries_fused()
    dom.set_target_time(2.5); ref.set_target(2.5)
    ref.run(120); dom.step_batch(120)
    sc, sr = dom.read_scalars(), ref.scalars()
    assert sc["batch_skipped"] == sr["batch_skipped"] > 0 and sc["time"] == sr["t"] == 2.5
    assert np.array_equal(dom.download(), ref.download())
    dom.set_target_time(6.0); ref.set_target(6.0)
    dom.update_timestep(); ref.update_timestep()
    ref.run(150); dom.step_batch(150)
    assert np.array_equal(dom.download(), ref.download())
    sc, sr = dom.read_scalars(), ref.scalars()
    assert sc["time"] == sr["t"] and sc["timestep"] == sr["dt"] and sc["time_hydrological"] == sr["t_hydro"]
    dom.close()


def test_a_cell_boundary_among_them_keeps_the_stand_alone_passes():
    g = load_golden("f11_cell_boundary_f64")
    rows, cols = g["bed"].shape
    dom = hp.Domain(cols, rows)
    dom.add_uniform(hp.UNIFORM_RAIN_INTENSITY, [[0.0, 10.0], [3600.0, 10.0]], 3600.0, 3600.0)
    assert dom.boundaries_fused()
    dom.add_cell(2, 1, g["cells"], g["series"], 5.0, 20.0)
    assert not dom.boundaries_fused()
    dom.close()


def test_boundaries_added_and_cleared_between_batches_and_a_checkpoint_in_between():
    """The host changes the boundary set while the run is under way, takes a device-side checkpoint mid-way and rolls back
    to it: the fused path (long batches) and the never-fused path (single-iteration batches) must agree bit for bit."""
    cols, rows, dx = 150, 90, 2.0
    st, bed, man = syn.s_rough(cols, rows, manning=None)
    rng = np.random.default_rng(9)
    grids = rng.uniform(0.0, 250.0, (3, 3, 4))

    def run(single):
        dom = hp.Domain(cols, rows, dx=dx)
        dom.upload(st, bed, man)
        dom.set_target_time(1e9)
        step = (lambda n: [dom.step_batch(1) for _ in range(n)]) if single else dom.step_batch
        step(30)
        dom.add_uniform(hp.UNIFORM_RAIN_INTENSITY, [[0.0, 90.0], [3600.0, 90.0]], 3600.0, 3600.0)
        step(41)
        dom.state_save()
        step(25)
        dom.clear_boundaries()
        step(13)
        dom.state_restore()                                   # back to iteration 71 (the boundary set is the host's business)
        dom.add_gridded(hp.GRIDDED_RAIN_INTENSITY, grids, 128.0, 0.0, 0.0, 10.0)
        assert dom.boundaries_fused()
        step(60)
        out, sc = dom.download(), dom.read_scalars()
        dom.close()
        return out, sc

    a, sa = run(False)
    b, sb = run(True)
    assert np.array_equal(a, b)
    assert sa["time"] == sb["time"] and sa["timestep"] == sb["timestep"] and sa["time_hydrological"] == sb["time_hydrological"]
    assert (a[..., 0] - bed).max() > 1e-3
