"""Config C1 to its END: test/newcastle-centre.xml is 7200 simulated seconds -- 1.0e5 iterations with the
target time clipped to each of twelve output times (src/CModel.cpp:723-770, src/Domain/Cartesian/CDomainCartesian.cpp:804-829),
an hour of rain and drainage (src/Boundaries/CLBoundaries.clc:130-184), then an hour of ponding and draining.  Fixture F16
(tests/golden/generate.py: newcastle_full) is that run on the reference's own kernels under the front end's CModel loop with
fixed batches; the engine runs the same loop.

STRICT: every depth and maxdepth raster bit for bit (SHA-256 of the float64 rasters at all twelve output times, the rasters
themselves at four), the simulation time at every output, the iteration counts.
FAST: the deviation curve over the run against the strict reference, beside the reference's own spread (its -cl-mad-enable
build and a build with another conforming pow(), at 2400 s and 7200 s)."""
import hashlib

import numpy as np
import pytest

import hipims_mi as hp
from conftest import load_golden, record
from hipims_mi.model import Model

pytestmark = pytest.mark.gpu


def run_c1(tmp_path, math_mode, batch):
    from model_dir import make_newcastle

    def make_sim(cfg, cols, rows, res):
        return hp.Domain(cols, rows, dx=res, scheme=cfg.scheme, precision=cfg.precision, dry_threshold=cfg.dry_threshold,
                         courant=cfg.courant, t_end=cfg.duration, dynamic_dt=cfg.dynamic_dt, dt_fixed=cfg.timestep,
                         dt_initial=cfg.timestep, friction=cfg.friction, math_mode=math_mode)
    m = Model(make_newcastle(tmp_path), make_sim=make_sim, output_format=None)
    m.scheme.automatic_queue = False                    # the reference's queueMode="fixed": batch boundaries are part of the run
    m.scheme.queue_addition_size = batch
    try:
        outs = m.run()
        return outs, m.scheme.iterations, m.scheme.batch_successful
    finally:
        m.close()


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float64).tobytes()).hexdigest()


def test_config_c1_full_run_strict_is_the_reference_run(tmp_path):
    g = load_golden("f16_newcastle_full_f64")
    outs, iterations, ok = run_c1(tmp_path, hp.MATH_STRICT, int(g["batch"]))
    assert len(outs) == 12
    assert np.array_equal(np.array([t for t, _ in outs]), g["t"])                    # every output time, to the bit
    assert iterations == int(g["iterations"]) and ok == int(g["successful"])
    for k, (t, o) in enumerate(outs):
        assert sha(o["depth"]) == str(g["depth_sha256"][k]), f"depth raster at t = {t}"
        assert sha(o["maxdepth"]) == str(g["maxdepth_sha256"][k]), f"maxdepth raster at t = {t}"
    for j, k in enumerate(g["kept"]):
        assert np.array_equal(outs[int(k)][1]["depth"], g["depth"][j])
    assert np.array_equal(outs[-1][1]["maxdepth"], g["maxdepth"])
    record("c1_full_strict", iterations=iterations, successful=ok, t_end=outs[-1][0], rmse=0.0)


def test_config_c1_full_run_fast_deviation_curve(tmp_path):
    """The FAST mode over the whole run: depth RMSE / max against the strict reference at 600, 2400, 4800 and 7200 s, recorded
    (profiles/r05*_parity_numbers.jsonl) and held to twice the spread of the reference's own three builds where the
    fixtures hold that spread (2400 s, 7200 s).  NODATA cells (dry below 1e-8 m) count as zero depth."""
    from itertools import combinations
    from hipims_mi import frontend
    g = load_golden("f16_newcastle_full_f64")
    others = [load_golden("f16_newcastle_full_f64_mad"), load_golden("f16_newcastle_full_f64_libm")]
    outs, iterations, ok = run_c1(tmp_path, hp.MATH_FAST, int(g["batch"]))
    depth = lambda a: np.where(a == frontend.NODATA, 0.0, a)
    kept = [int(k) for k in g["kept"]]
    curve = []
    for j, k in enumerate(kept):
        d = depth(outs[k][1]["depth"]) - depth(g["depth"][j])
        curve.append((outs[k][0], float(np.sqrt(np.mean(d ** 2))), float(np.abs(d).max())))
    bracket = {}
    for slot, k in enumerate((3, 11)):                  # the other builds hold 2400 s and 7200 s
        j = kept.index(k)
        trio = [depth(g["depth"][j])] + [depth(o["depth"][slot]) for o in others]
        bracket[k] = (max(float(np.sqrt(np.mean((a - b) ** 2))) for a, b in combinations(trio, 2)),
                      max(float(np.abs(a - b).max()) for a, b in combinations(trio, 2)))
    for (t, rmse, mx), k in zip(curve, kept):
        record("c1_full_fast", t=t, rmse=rmse, max=mx, bracket_rmse=bracket.get(k, (None, None))[0],
               bracket_max=bracket.get(k, (None, None))[1], iterations=iterations, iterations_ref=int(g["iterations"]))
    # the timestep sequence follows the water: the iteration count may differ from the reference's by a handful, not by a trend
    assert abs(iterations - int(g["iterations"])) <= 2 * int(g["batch"])
    assert abs(outs[-1][0] - float(g["t"][-1])) < 1e-5
    for (t, rmse, mx), k in zip(curve, kept):
        if k in bracket:
            assert rmse < 2 * bracket[k][0] and mx < 2 * bracket[k][1], (t, rmse, mx, bracket[k])
    # (north_star's 1e-9 m holds for the first minutes only -- 1.2e-8 m at 600 s -- because the reference's OWN builds part by
    # 2e-4 m once water ponds: the bar a non-bit-exact mode can be held to is that spread; STRICT above is the reference run)
