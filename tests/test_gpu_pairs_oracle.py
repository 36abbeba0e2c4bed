"""The kernel the headline is timed on -- godunov_march2, two Godunov iterations per launch -- directly against the ORACLE and the
committed reference-kernel fixtures (tests/test_gpu_two_step.py holds pairs to single FAST iterations, a
self-comparison; the oracle comparisons of tests/test_gpu_parity.py run below the default's threshold of 1.5 M cells and therefore
on godunov_march).  Every case is a subprocess with HP_TWO_STEP=1 (read once per process): pairs wherever two iterations are to be
had, and the launch count proves that they ran.  Tolerances are north_star's FAST bars: fp64 depth RMSE < 1e-9 m, max < 1e-7 m,
elapsed time to 1e-12 relative; fp32 depth RMSE < 1e-4 m.  Reference: the iteration graph CSchemeGodunov.cpp:1617-1666 (quirk Q1
at :1629, :1634 -- what makes the pair's second timestep known in advance), untouched cells CLSchemeGodunov.clc:248-255 (Q3)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import load_golden, record

pytestmark = pytest.mark.gpu
WORKER = os.path.join(os.path.dirname(__file__), "pairs_oracle_worker.py")


def run(case, precision, tmp_path, two_step="1"):
    out = os.path.join(str(tmp_path), f"{case}_{precision}.npz")
    env = {k: v for k, v in os.environ.items() if k != "HP_TWO_STEP"}
    if two_step is not None:
        env["HP_TWO_STEP"] = two_step
    r = subprocess.run([sys.executable, WORKER, case, precision, out], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    return np.load(out)


def depth_errors(a, b, bed):
    da, db = np.maximum(0, a[..., 0].astype(np.float64) - bed), np.maximum(0, b[..., 0].astype(np.float64) - bed)
    return float(np.sqrt(np.mean((da - db) ** 2))), float(np.abs(da - db).max())


def check(d, precision, case, pairs_expected=True):
    if pairs_expected:
        assert int(d["launches"]) < 0.62 * int(d["iterations"]), (int(d["launches"]), int(d["iterations"]))   # no silent fall-back to single iterations
    rmse, mx = depth_errors(d["got"], d["want"], d["bed"])
    t_rel = abs(float(d["t"]) - float(d["t_ref"])) / float(d["t_ref"])
    record("pairs_vs_oracle", workload=case, precision=precision, rmse=rmse, max=mx, time_rel=t_rel,
           launches=int(d["launches"]), iterations=int(d["iterations"]))
    if precision == "f64":
        assert rmse < 1e-9 and mx < 1e-7, (rmse, mx)
        assert t_rel <= 1e-12, t_rel
    else:
        assert rmse < 1e-4, rmse
        assert t_rel <= 1e-4, t_rel
    assert int(d["ok"]) == int(d["ok_ref"])
    return rmse, mx


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("case", ["rough1024", "damdry1024"])
def test_pair_kernel_against_the_oracle_at_a_million_cells(case, precision, tmp_path):
    """S-ROUGH and S-DAM-DRY 1024^2, friction on, 250 iterations = 125 launches of godunov_march2."""
    check(run(case, precision, tmp_path), precision, case)


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_pair_kernel_against_the_reference_kernel_fixtures(precision, tmp_path):
    """Fixture F6 (tests/golden/generate.py: the reference's own Godunov kernels built for the host): the rough bed after 200
    iterations, both dam breaks after 150 -- run here as 100 / 75 pair launches."""
    g = load_golden(f"f6_f7_trajectories_{precision}")
    for case, key, t_key in (("f6_rough", "god_q_state200", "god_q_t"), ("f6_dam", "dam_god_state150", "dam_god_dt"),
                             ("f6_damdry", "damdry_god_state150", "damdry_god_dt")):
        d = run(case, precision, tmp_path)
        check(d, precision, case)
        rmse, mx = depth_errors(d["got"], g[key], d["bed"])
        record("pairs_vs_fixture_f6", workload=case, precision=precision, rmse=rmse, max=mx)
        t_ref = float(g[t_key]) if g[t_key].ndim == 0 else float(g[t_key].astype(np.float64).sum())
        if precision == "f64":
            assert rmse < 1e-9 and mx < 1e-7, (case, rmse, mx)
            assert abs(float(d["t"]) - t_ref) < 1e-10, case
        else:
            assert rmse < 1e-4, (case, rmse)


@pytest.mark.parametrize("case", ["strict_rough1024", "strict_damdry1024", "strict_f6_rough"])
def test_strict_pair_kernel_is_the_oracle_bit_for_bit(case, tmp_path):
    """godunov_march2<STRICT> (round 6) against the oracle: every cell of the final state, the elapsed time and the timestep, bit for
    bit -- 1024^2 wet/dry rough terrain and the dry-bed dam break with friction, 250 iterations as 125 launches; fixture F6's rough bed
    against the reference's own kernels."""
    d = run(case, "f64", tmp_path)
    assert int(d["launches"]) < 0.62 * int(d["iterations"])
    assert np.array_equal(d["got"], d["want"])
    assert float(d["t"]) == float(d["t_ref"]) and float(d["dt"]) == float(d["dt_ref"])
    if case == "strict_f6_rough":
        assert np.array_equal(d["got"], load_golden("f6_f7_trajectories_f64")["god_q_state200"])


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("case", ["rain1024", "rainloss1024", "bigdt1024"])
def test_pair_kernel_with_area_boundaries_against_the_oracle(case, precision, tmp_path):
    """godunov_march2 with area boundaries (round 6: the second iteration's rain in registers between the two steps, the state stored
    with the next iteration's, priced both ways) against the ORACLE -- tests/test_gpu_two_step.py holds such pairs to single
    iterations, which is a self-comparison.  1024^2 at 2 m, pools on mostly dry rough terrain, uniform rain + gridded rain on rain
    cells 128 model cells wide, 250 iterations as pair launches; with a loss rate on top the engine takes the exact flavour (stamps);
    bigdt1024: 40 m cells, 608 iterations -- past the first minute (the reference holds the timestep at 0.1 s until then) the timestep
    comes out of the reduction at up to 7.8 s, so the hydrological gate opens on every iteration and the elapsed time depends on
    every pair having priced its stored state WITH the next iteration's rain (eight such iterations: the film amplifies rounding
    tenfold every five of them, see the worker).  north_star's FAST bars.
    Reference: Boundaries/CLBoundaries.clc:130-246, the hydrological gate CLDynamicTimestep.clc:61-66."""
    if case == "bigdt1024" and precision == "f32":
        pytest.skip("fp32 rounding under 7-second timesteps on this film is 5e-3 in elapsed time after eight iterations (measured): not a parity case")
    d = run(case, precision, tmp_path)
    assert int(d["launches"]) < 0.62 * int(d["iterations"]), (int(d["launches"]), int(d["iterations"]))
    rmse, mx = depth_errors(d["got"], d["want"], d["bed"])
    t_rel = abs(float(d["t"]) - float(d["t_ref"])) / float(d["t_ref"])
    record("pairs_with_boundaries_vs_oracle", workload=case, precision=precision, rmse=rmse, max=mx, time_rel=t_rel,
           launches=int(d["launches"]), iterations=int(d["iterations"]))
    wet = np.maximum(0, d["want"][..., 0].astype(np.float64) - d["bed"])
    assert (wet > 1e-4).mean() > 0.3                      # the rain has wetted the terrain: the comparison is not of dry land
    if precision == "f64":
        assert rmse < 1e-9 and mx < 1e-7, (rmse, mx)
        assert t_rel <= 1e-12, t_rel
    else:
        assert rmse < 1e-4 and t_rel <= 1e-4, (rmse, t_rel)
    if case == "bigdt1024":
        assert float(d["t"]) > 80.0                            # eight iterations took the run from t = 60 s past 80 s: timesteps far beyond a second
    assert int(d["ok"]) == int(d["ok_ref"])


def test_strict_engine_is_the_oracle_through_the_regime_that_amplifies_rounding(tmp_path):
    """The same 40-m film for 700 iterations -- a hundred of them at timesteps of up to 7.8 s, where FAST arithmetic and the oracle end
    1e-3 m and 2 % in elapsed time apart (rounding grows tenfold every five iterations: tools/diag_bigdt.py) -- in the exact mode:
    every bit of the state, the time and the timestep (single iterations with the boundaries fused into the flux kernel: the exact
    mode does not pair under area boundaries)."""
    d = run("strict_bigdt700", "f64", tmp_path, two_step=None)
    assert np.array_equal(d["got"], d["want"])
    assert float(d["t"]) == float(d["t_ref"]) and float(d["dt"]) == float(d["dt_ref"]) and float(d["t"]) > 500.0


def test_pair_kernel_with_rain_and_loss_against_the_reference_kernel_fixture(tmp_path):
    """Fixture F9's uniform rain + loss rate (tests/golden/generate.py: the reference's own kernels, 420 iterations) run in PAIRS
    (exact flavour: a loss rate dries whole regions at once): the bars of test_gpu_parity's single-iteration test of the same fixture."""
    d = run("f9_uniform", "f64", tmp_path)
    assert int(d["launches"]) < 0.62 * int(d["iterations"])
    check(d, "f64", "f9_uniform")
    g = load_golden("f9_rain_f64")
    rmse, mx = depth_errors(d["got"], g["uniform_state"], d["bed"])
    record("pairs_vs_fixture_f9", workload="f9_uniform", precision="f64", rmse=rmse, max=mx)
    assert rmse < 1e-9 and mx < 1e-7, (rmse, mx)
    assert abs(float(d["t"]) - float(g["uniform_t"])) < 1e-9


def test_default_selection_against_the_oracle(tmp_path):
    """No HP_TWO_STEP in the environment: 1500 x 1100 lies above the default's threshold, so the selection logic itself
    (hp_domain_create: march2_pays, the one-round tiling search) is what chooses godunov_march2 here; two batches of odd length put
    a single iteration (K1 with its FILL flag) between the pairs."""
    d = run("default1500", "f64", tmp_path, two_step=None)
    assert int(d["launches"]) < 0.62 * int(d["iterations"])
    check(d, "f64", "default1500")
