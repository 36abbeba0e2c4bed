"""Two iterations per pass (hp_kernels.hpp: godunov_march2, round 5): a pair of Godunov iterations run as ONE launch must be the SAME
computation as the two single-iteration launches -- same per-cell operations in the same order, the intermediate state in registers
instead of memory -- so the FAST engine with pairs (HP_TWO_STEP=1) is held to the FAST engine without (HP_TWO_STEP=0) BIT FOR BIT:
state, time, timestep, counters; over batches of odd and even length, downloads in between, a sync point with clipped and
suspended iterations, tst_UpdateTimestep, a device checkpoint, wet/dry terrain with untouched cells (quirk Q3), fp64 and fp32,
dynamic and fixed timestep.  (This is a self-comparison.  The pair kernel against the ORACLE and the reference-kernel fixtures:
tests/test_gpu_pairs_oracle.py -- the oracle comparisons of tests/test_gpu_parity.py stay below the default's threshold, 1.5 M cells
or two rounds of blocks at 12-row tiles, and run godunov_march.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
WORKER = os.path.join(os.path.dirname(__file__), "two_step_worker.py")


def run(scenario, precision, tmp_path, mode, **env):
    out = os.path.join(str(tmp_path), f"{scenario}_{precision}_{mode}{''.join('_' + k + v for k, v in env.items())}.npz")
    r = subprocess.run([sys.executable, WORKER, scenario, precision, out], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, HP_TWO_STEP=str(mode), **env))
    assert r.returncode == 0, r.stdout + r.stderr
    return np.load(out)


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("scenario", ["dam", "rough", "damdry", "fixed"])
def test_pairs_are_the_same_computation_as_single_iterations(scenario, precision, tmp_path):
    single, pairs = run(scenario, precision, tmp_path, 0), run(scenario, precision, tmp_path, 1)
    assert int(single["launches"]) == int(single["iterations"])                 # one launch per iteration ...
    assert int(pairs["launches"]) < int(pairs["iterations"]) * 0.62             # ... against pairs wherever two iterations were to be had
    assert int(pairs["iterations"]) == int(single["iterations"])
    for key in ("t", "dt", "ok", "skipped"):
        assert pairs[key] == single[key], key
    assert np.array_equal(pairs["state"], single["state"])


def test_the_single_iteration_after_pairs_fills_or_copies_to_the_same_bits(tmp_path):
    """After pairs the non-current buffer is two states old.  The first single iteration behind them either stores the cells the
    reference leaves untouched itself (K1's FILL flag, the default) or is preceded by a device copy of the current state
    (HP_FILL_AFTER_PAIRS=0, the round-5 A/B switch): the same bits, on wet/dry terrain where untouched cells exist, over batches of
    odd length (every one ends in such a single iteration) -- and both equal the run without pairs."""
    outs = {}
    for name, env in (("fill", {"HP_TWO_STEP": "1"}), ("copy", {"HP_TWO_STEP": "1", "HP_FILL_AFTER_PAIRS": "0"}), ("single", {"HP_TWO_STEP": "0"})):
        out = os.path.join(str(tmp_path), f"{name}.npz")
        r = subprocess.run([sys.executable, WORKER, "rough", "f64", out], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stdout + r.stderr
        outs[name] = np.load(out)
    assert int(outs["fill"]["launches"]) == int(outs["copy"]["launches"]) < int(outs["single"]["launches"])
    for other in ("copy", "single"):
        assert np.array_equal(outs["fill"]["state"], outs[other]["state"]), other
        assert outs["fill"]["t"] == outs[other]["t"] and outs["fill"]["dt"] == outs[other]["dt"], other


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_a_boundary_added_behind_pairs(precision, tmp_path):
    """Pairs need a domain without boundary conditions; one may be added later.  The first single iteration behind the pairs then is
    K1 with the rain FUSED in and the FILL flag set (its destination is two states old): the run must equal the one without pairs bit
    for bit -- and the one that copies instead of filling."""
    outs = {}
    for name, env in (("fill", {"HP_TWO_STEP": "1"}), ("copy", {"HP_TWO_STEP": "1", "HP_FILL_AFTER_PAIRS": "0"}), ("single", {"HP_TWO_STEP": "0"})):
        out = os.path.join(str(tmp_path), f"{name}.npz")
        r = subprocess.run([sys.executable, WORKER, "rainlater", precision, out], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stdout + r.stderr
        outs[name] = np.load(out)
    assert int(outs["fill"]["launches"]) == int(outs["copy"]["launches"]) <= int(outs["single"]["launches"]) - 10      # the first 24 iterations: pairs
    for other in ("copy", "single"):
        assert np.array_equal(outs["fill"]["state"], outs[other]["state"]), other
        assert outs["fill"]["t"] == outs[other]["t"] and outs["fill"]["dt"] == outs[other]["dt"], other


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("scenario", ["rain", "rainloss", "gridrain", "massflux", "bigdt", "drying"])
def test_pairs_with_area_boundaries_are_the_same_computation(scenario, precision, tmp_path):
    """Round 6: iteration pairs on domains with rain / loss (godunov_march2 BDY).  The reference applies the boundaries in place before
    every flux kernel (CSchemeGodunov.cpp:1638; CLBoundaries.clc:130-246): the pair applies the second iteration's in registers
    between its two steps, stores its state with the next iteration's, and prices that state both ways (the iteration that reads the
    primary buffer prices it WITH its rain: quirk Q1).  Held to the single-iteration engine (K1 with the fused epilogue, itself held to
    the oracle by test_gpu_fused_rain.py) bit for bit: uniform rain, rain + loss (drying: quirk Q3's stale values, the stamps), gridded
    rain, mass flux + rain, timesteps above a second (every gate open), over odd / even batches, downloads, a sync point,
    tst_UpdateTimestep and a checkpoint."""
    single, pairs = run(scenario, precision, tmp_path, 0), run(scenario, precision, tmp_path, 1, HP_PAIR_EXACT="1")
    assert int(single["launches"]) == int(single["iterations"])
    assert int(pairs["launches"]) < int(pairs["iterations"]) * 0.75             # (every batch of the plan starts cold or ends single)
    assert int(pairs["iterations"]) == int(single["iterations"])
    for key in ("t", "dt", "ok", "skipped"):
        assert pairs[key] == single[key], key
    assert np.array_equal(pairs["state"], single["state"])


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("scenario", ["strict_dam", "strict_rough", "strict_damdry"])
def test_strict_pairs_are_the_same_computation(scenario, precision, tmp_path):
    """Round 6: the exact mode pairs too (godunov_march2<STRICT>: K1's statements in K1's order, still-water rows skipped in both
    stages, quirk Q3's stamps always on).  Same bits as the single STRICT iterations -- which tests/test_gpu_strict_friction.py and
    the fixtures hold to the reference's kernels -- state, time, timestep, counters."""
    single, pairs = run(scenario, precision, tmp_path, 0), run(scenario, precision, tmp_path, 1)
    assert int(single["launches"]) == int(single["iterations"])
    assert int(pairs["launches"]) < int(pairs["iterations"]) * 0.62
    for key in ("t", "dt", "ok", "skipped"):
        assert pairs[key] == single[key], key
    assert np.array_equal(pairs["state"], single["state"])


@pytest.mark.parametrize("scenario", ["strict_tune_dam", "strict_tune_rough"])
def test_the_exact_mode_chooses_between_pairs_and_single_iterations_by_measurement(scenario, tmp_path):
    """STRICT pairs are STRICT single iterations bit for bit, and which is faster depends on the water (still rows are a copy, which a pair
    makes at half the bytes; moving water is bound by instruction issue, where the pair's overheads cost): no HP_TWO_STEP in the
    environment, the engine samples one pair against the two single iterations behind it (events, nothing blocks) and runs what won.
    Whatever it chooses, and the sample itself, leave the same bits as single iterations throughout."""
    env = {k: v for k, v in os.environ.items() if k != "HP_TWO_STEP"}
    outs = {}
    for name, extra in (("tuned", {}), ("single", {"HP_TWO_STEP": "0"})):
        out = os.path.join(str(tmp_path), f"{name}.npz")
        r = subprocess.run([sys.executable, WORKER, scenario, "f64", out], capture_output=True, text=True, timeout=900, env=dict(env, **extra))
        assert r.returncode == 0, r.stdout + r.stderr
        outs[name] = np.load(out)
    t, s1 = outs["tuned"], outs["single"]
    assert np.array_equal(t["state"], s1["state"]) and t["t"] == s1["t"] and t["dt"] == s1["dt"]
    assert int(t["tune_samples"]) >= 1 and int(t["pairs"]) >= 1
    print(f"{scenario}: pair / two single iterations = {float(t['pair_over_single']):.3f}, prefers pairs: {bool(t['prefers_pairs'])}, launches {int(t['launches'])} of {int(t['iterations'])}")


def test_a_restore_discards_the_sample_in_flight(tmp_path):
    """The measurement belongs to the state it was taken on: hp_state_restore (and an upload) start it anew even when a sample is still
    in flight -- bench.py restores its checkpoint right behind a pre-warm phase whose last batch had taken one, and the timed region
    then ran on a choice made for the pre-warmed flood (the strict leg fell from 0.63 to 0.55 when samples became frequent enough
    for that to happen nearly every time).  A batch of sixteen iterations from a fresh state that samples is twelve flux launches (two
    single iterations until the reduction is current, the sample's three pairs and six single iterations, one pair); one that does
    not is nine."""
    code = f"""
import sys; sys.path[:0] = [{os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r}, {os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hipims-ocl_amd")!r}]
import numpy as np, hipims_mi as hp
from hipims_mi import synthetic as syn
st, bed, man = syn.s_dam(2048, 1500, dtype=np.float64)
dom = hp.Domain(2048, 1500, math_mode=hp.MATH_STRICT)
dom.upload(st, bed, man); dom.set_target_time(1e9)
dom.state_save()
counts = []
for k in range(3):
    c0 = dom.launch_counts()[0]; dom.step_batch(16); counts.append(dom.launch_counts()[0] - c0)   # (no synchronisation: the sample is in flight)
    dom.state_restore()
dom.read_scalars()
print("LAUNCHES", counts)
"""
    env = {k: v for k, v in os.environ.items() if k not in ("HP_TWO_STEP", "HP_PAIR_TUNE", "HP_PAIR_STRICT")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("LAUNCHES")][-1]
    assert line == "LAUNCHES [12, 12, 12]", line


@pytest.mark.parametrize("scenario", ["rough", "damdry"])
def test_exact_pairs_without_boundaries(scenario, tmp_path):
    """HP_PAIR_EXACT=1 on domains without boundary conditions (where the default leaves the stamps out, for speed): the same bits as the
    single iterations, as the default's -- wet/dry terrain, a dry-bed dam break through a sync point and a checkpoint."""
    single, exact = run(scenario, "f64", tmp_path, 0), run(scenario, "f64", tmp_path, 1, HP_PAIR_EXACT="1")
    assert int(exact["launches"]) < int(exact["iterations"]) * 0.62
    for key in ("t", "dt", "ok", "skipped"):
        assert exact[key] == single[key], key
    assert np.array_equal(exact["state"], single["state"])
    # the audit VERDICT r05 asked for (weak #1): how many first-step-untouched cells took a stale value that was NOT their current state --
    # the cells the default flavour (no stamps on domains without boundaries) would have got wrong here.  None, on these workloads: what
    # round 5's probe of the reference's kernels found (881 181 cell-iterations), now counted by the engine itself on every exact run.
    assert int(exact["stale_used"]) == 0, int(exact["stale_used"])
    print(f"{scenario}: stamps written for {int(exact['stamped_ever'])} cells, stale values that differed from the current state: {int(exact['stale_used'])}")


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("scenario", ["rain", "gridrain"])
def test_default_pairs_with_rain(scenario, precision, tmp_path):
    """Rain only (nothing removes water): the default runs such domains' pairs WITHOUT the stamps (godunov_march2's HZ = false: 4-9 %
    faster) -- a cell left untouched at a pair's first step passes its current state on where the reference keeps a stale one, which
    differ only if the cell dried out that very step.  Held to the single iterations at FAST's own bar against the oracle (depth RMSE
    1e-9 m fp64, 1e-4 m fp32) -- and, on these workloads, found equal to the last bit."""
    single, pairs = run(scenario, precision, tmp_path, 0), run(scenario, precision, tmp_path, 1)
    assert int(pairs["launches"]) < int(pairs["iterations"]) * 0.75 and int(pairs["stamped_ever"]) == 0
    d = np.abs(pairs["state"][..., 0].astype(np.float64) - single["state"][..., 0])
    rmse = float(np.sqrt(np.mean(d ** 2)))
    assert rmse < (1e-9 if precision == "f64" else 1e-4), rmse
    assert abs(float(pairs["t"]) - float(single["t"])) <= (1e-12 if precision == "f64" else 1e-5) * float(single["t"])
    print(f"default pairs with {scenario} {precision}: depth RMSE vs single iterations {rmse:.3e} m, equal bits: {np.array_equal(pairs['state'], single['state'])}")


def test_config_c1_in_pairs_is_config_c1_in_single_iterations(tmp_path):
    """Config C1 -- the reference's example model, uniform rain + uniform drainage on its own DEM -- 2000 iterations in FAST arithmetic with
    pairs forced (the grid is far below the size from which pairs pay; the loss rate makes them the exact flavour by default) against
    single iterations: every bit of the state, the time, the timestep.  Its cells dry out under the drainage all the time: the stamps
    are written by the thousand and stale values that differ from the current state are really taken (the audit counter)."""
    single, pairs = run("c1", "f64", tmp_path, 0), run("c1", "f64", tmp_path, 1)
    assert int(pairs["launches"]) < int(pairs["iterations"]) * 0.62 and int(pairs["iterations"]) == 2000
    for key in ("t", "dt", "ok", "skipped"):
        assert pairs[key] == single[key], key
    assert np.array_equal(pairs["state"], single["state"])
    assert int(pairs["stamped_ever"]) > 0
    print(f"c1: stamps on {int(pairs['stamped_ever'])} cells, stale values that differed and were taken: {int(pairs['stale_used'])}, cold starts {int(pairs['cold_starts'])}")


def test_the_stamps_are_what_makes_pairs_exact_where_cells_dry_out(tmp_path):
    """Quirk Q3 across pair launches (hp_kernels.hpp: PairAux).  A cell the reference leaves untouched at a pair's first
    step keeps the value of the iteration before the pair -- which only the launch before ever had.  Where the loss rate dries whole
    regions at once such cells are many: with the stamps switched off (HP_PAIR_EXACT=0: round 5's behaviour, the cell's current value
    stands in -- what domains without such boundaries run by default, for speed) the run parts from the single iterations; with them it does not, and the stamps were really written."""
    outs = {}
    for name, env in (("single", {"HP_TWO_STEP": "0"}), ("stamps", {"HP_TWO_STEP": "1"}), ("none", {"HP_TWO_STEP": "1", "HP_PAIR_EXACT": "0"})):
        out = os.path.join(str(tmp_path), f"{name}.npz")
        r = subprocess.run([sys.executable, WORKER, "rainloss", "f64", out], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stdout + r.stderr
        outs[name] = np.load(out)
    assert np.array_equal(outs["stamps"]["state"], outs["single"]["state"]) and outs["stamps"]["t"] == outs["single"]["t"]
    assert int(outs["stamps"]["stamped_ever"]) > 0 and int(outs["none"]["stamped_ever"]) == 0
    assert int(outs["stamps"]["stale_used"]) > 0                                        # (the audit counter: stale values that differed were really taken)
    assert int(outs["stamps"]["cold_starts"]) >= 1 and int(outs["stamps"]["pairs"]) == int(outs["none"]["pairs"]) > 100
    assert not np.array_equal(outs["none"]["state"], outs["single"]["state"])           # (what the stamps are for)


def test_default_takes_pairs_on_big_grids_only(tmp_path):
    out = os.path.join(str(tmp_path), "default.npz")
    env = {k: v for k, v in os.environ.items() if k != "HP_TWO_STEP"}
    r = subprocess.run([sys.executable, WORKER, "dam", "f64", out], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    d = np.load(out)
    assert int(d["launches"]) == int(d["iterations"])                           # 1030 x 700 = 0.72 M cells: below the threshold


def test_pairs_fuzz():
    """Forty random configurations -- shapes from 5 x 5 to 1500 x 1100, both precisions, FAST and (round 6) now and then STRICT, dam breaks
    and wet/dry terrain with null cells, Manning arrays, friction on / off, fixed and dynamic timestep, (round 6) uniform rain / a loss
    rate / gridded rain / a mass flux in random combination, batches of random length with downloads, partial
    uploads, new target times, tst_UpdateTimestep and checkpoints in between: pairs forced on (exact flavour) against pairs off, every
    observable (states at every download, time, timestep, counters) hashed -- the two runs must print the same lines."""
    worker = os.path.join(os.path.dirname(__file__), "two_step_fuzz_worker.py")
    outs = []
    for mode in ("0", "1"):
        r = subprocess.run([sys.executable, worker, "500", "40"], capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, HP_TWO_STEP=mode, HP_PAIR_EXACT="1"))
        assert r.returncode == 0, r.stdout + r.stderr
        lines = [l for l in r.stdout.splitlines() if l.startswith("seed ")]
        assert len(lines) == 40
        outs.append(lines)
    strip = lambda l: l.split("  # ")[0]
    assert [strip(l) for l in outs[0]] == [strip(l) for l in outs[1]]
    launches = lambda ls: sum(int(l.split("launches ")[1]) for l in ls)
    assert launches(outs[1]) < 0.85 * launches(outs[0])                     # (and pairs really ran)
