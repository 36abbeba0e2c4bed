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


def test_default_selection_against_the_oracle(tmp_path):
    """No HP_TWO_STEP in the environment: 1500 x 1100 lies above the default's threshold, so the selection logic itself
    (hp_domain_create: march2_pays, the one-round tiling search) is what chooses godunov_march2 here; two batches of odd length put
    a single iteration (K1 with its FILL flag) between the pairs."""
    d = run("default1500", "f64", tmp_path, two_step=None)
    assert int(d["launches"]) < 0.62 * int(d["iterations"])
    check(d, "f64", "default1500")
