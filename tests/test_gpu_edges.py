"""Edge cases of the HIP path: smallest grids each scheme accepts, widths around the 62/60-column wavefront tiles and
the 64-lane boundary, single-segment and ragged last segments, rejected inputs, and the largest BASELINE
configuration (config C4's whole 16384 x 8192 grid) on one GPU.  GPU only."""
import numpy as np
import pytest

import hipims_mi as hp
import oracle
from hipims_mi import synthetic as syn

pytestmark = pytest.mark.gpu

SIZES_3 = [(3, 3), (4, 3), (3, 7), (5, 5), (63, 9), (64, 9), (65, 9), (125, 20), (126, 5), (200, 3), (187, 18), (62, 17)]
SIZES_5 = [(5, 5), (6, 5), (5, 9), (61, 7), (64, 8), (65, 33), (124, 37), (121, 36), (66, 6)]


def _case(cols, rows, seed):
    st, bed, man = syn.s_rough(cols, rows, seed=seed, manning=None)
    return st, bed, man


@pytest.mark.parametrize("scheme,sizes", [(hp.SCHEME_GODUNOV, SIZES_3), (hp.SCHEME_INERTIAL, SIZES_3),
                                          (hp.SCHEME_MUSCL_HANCOCK, SIZES_5)])
def test_small_and_ragged_grids_match_the_oracle(scheme, sizes):
    for i, (cols, rows) in enumerate(sizes):
        st, bed, man = _case(cols, rows, 100 + i)
        quirks = oracle.QUIRKS_REFERENCE & ~(oracle.Q6_MUSCL_SERIAL if scheme == hp.SCHEME_MUSCL_HANCOCK else 0)
        # STRICT without friction involves no transcendental: bit-identical (the inertial flux always has its pow)
        exact = scheme != hp.SCHEME_INERTIAL
        ref = oracle.OracleSim(cols, rows, scheme=scheme, quirks=quirks, friction=not exact)
        dom = hp.Domain(cols, rows, scheme=scheme, friction=not exact, math_mode=hp.MATH_STRICT)
        for s in (ref, dom):
            s.upload(st, bed, man)
        dom.set_target_time(1e9); ref.set_target(1e9)
        ref.run(40); dom.step_batch(40)
        a, b = dom.download(), ref.download()
        if exact:
            assert np.array_equal(a, b), (scheme, cols, rows)
            assert dom.read_scalars()["time"] == ref.scalars()["t"]
        else:
            assert np.abs(a - b).max() < 1e-9, (scheme, cols, rows, np.abs(a - b).max())
        # FAST on the same case, to the stated tolerance
        fast = hp.Domain(cols, rows, scheme=scheme, friction=not exact)
        fast.upload(st, bed, man)
        fast.set_target_time(1e9)
        fast.step_batch(40)
        assert np.abs(fast.download()[..., 0] - b[..., 0]).max() < 1e-7, (scheme, cols, rows)
        for d in (dom, fast):
            d.close()


def test_rejected_inputs_and_empty_batches():
    with pytest.raises(hp.HipimsError):
        hp.Domain(0, 10)
    with pytest.raises(hp.HipimsError):
        hp.Domain(10, 10, scheme=7)
    d = hp.Domain(4, 4, scheme=hp.SCHEME_MUSCL_HANCOCK)                  # allocates, but cannot step: needs 5 x 5
    st, bed, man = syn.s_dam(4, 4)
    d.upload(st, bed, man)
    with pytest.raises(hp.HipimsError):
        d.step_batch(1)
    d.close()
    d = hp.Domain(8, 8)
    st, bed, man = syn.s_dam(8, 8)
    d.upload(st, bed, man)
    d.set_target_time(1.0)
    d.step_batch(0)                                                      # nothing queued, nothing changes
    assert np.array_equal(d.download(), st) and d.read_scalars()["time"] == 0.0
    with pytest.raises(ValueError):
        d.upload(st[:4], bed, man)                                       # ragged host array
    d.close()


def test_config_c4_whole_grid_on_one_gpu():
    """16384 x 8192 (134 M cells, 10.7 GB of device arrays): the S-DAM input does not vary along y, so every interior
    row must stay bit-identical to every other -- any tile / offset arithmetic that breaks beyond 2^31 bytes shows --
    and the closed basin conserves mass.  Host arrays are built and moved in row blocks."""
    cols, rows, block = 16384, 8192, 512
    d = hp.Domain(cols, rows)
    row = np.zeros((1, cols, 4)); row[0, :cols // 2, 0] = 10.0; row[0, cols // 2:, 0] = 1.0; row[0, :, 1] = row[0, :, 0]
    row[0, 0] = 0; row[0, -1] = 0
    bed_row = np.zeros((1, cols)); bed_row[0, 0] = bed_row[0, -1] = syn.WALL_BED
    bed = np.repeat(bed_row, rows, axis=0); bed[0] = bed[-1] = syn.WALL_BED
    d.upload(bed=bed, manning=np.full((rows, cols), 0.03))
    del bed
    for r0 in range(0, rows, block):
        blk = np.repeat(row, block, axis=0)
        if r0 == 0:
            blk[0] = 0
        if r0 + block == rows:
            blk[-1] = 0
        d.upload_rows(blk, r0)
    d.set_target_time(1e9)
    d.step_batch(120)
    first = d.download(row0=1, nrows=1)[0]
    mass, mass0 = 0.0, float(row[0, 1:-1, 0].sum()) * (rows - 2)
    for r0 in range(0, rows, block):
        blk = d.download(row0=r0, nrows=block)
        lo, hi = (1 if r0 == 0 else 0), (block - 1 if r0 + block == rows else block)
        assert np.array_equal(blk[lo:hi], np.broadcast_to(first, (hi - lo, cols, 4))), r0
        mass += float(blk[lo:hi, 1:-1, 0].sum())
    assert abs(mass - mass0) / mass0 < 1e-12
    assert first[cols // 2 - 40:cols // 2 + 40, 2].max() > 1.0          # the dam-break wave is under way
    d.close()


@pytest.mark.parametrize("scheme,levels", [(hp.SCHEME_GODUNOV, (10.0, 1.0)), (hp.SCHEME_MUSCL_HANCOCK, (10.0, 1.0)),
                                           (hp.SCHEME_INERTIAL, (2.0, 1.6))])
def test_long_run_stays_finite_and_conservative(scheme, levels):
    """30 000 iterations of a closed basin sloshing back and forth (waves reflect off the walls many times): nothing
    drifts -- volume conserved to 1e-10 (Godunov, inertial), no NaN, Zmax never below Z, the batch counters add up."""
    cols, rows, steps = 512, 384, 30000
    st, bed, man = syn.s_dam(cols, rows, levels=levels)
    d = hp.Domain(cols, rows, scheme=scheme)
    d.upload(st, bed, man)
    d.set_target_time(1e9)
    d.step_batch(steps)
    out, sc = d.download(), d.read_scalars()
    assert np.isfinite(out).all()
    depth0, depth = np.maximum(0, st[..., 0] - bed), np.maximum(0, out[..., 0] - bed)
    if scheme != hp.SCHEME_MUSCL_HANCOCK:
        assert abs(depth.sum() - depth0.sum()) / depth0.sum() < 1e-10
    else:
        # the reference's corrector only updates cells 2..n-3 (CLSchemeMUSCLHancock.clc:569-573): ring 1 keeps its
        # initial level for ever and exchanges water with ring 2, so a MUSCL run is not closed; it must stay bounded
        # between the two initial levels
        assert depth[2:-2, 2:-2].min() > 0.5 and depth.max() < 10.5
    inner = np.s_[1:-1, 1:-1]
    assert (out[inner][..., 1] >= out[inner][..., 0]).all()
    assert sc["batch_successful"] == steps and sc["batch_skipped"] == 0 and sc["time"] > 500.0
    d.close()


def test_log_sink_reports_a_rejected_boundary_with_the_reference_level():
    """A gridded boundary that does not cover the interior cells is rejected at hp_boundary_add_gridded (the reference
    would read outside its buffer, CLBoundaries.clc:231-236) and the message reaches the log sink at kLevelModelStop."""
    seen = []
    hp.set_log_sink(lambda level, text: seen.append((level, text)))
    try:
        d = hp.Domain(64, 48, dx=2.0)
        grids = np.zeros((2, 4, 4))
        with pytest.raises(hp.HipimsError, match="does not cover"):
            d.add_gridded(hp.GRIDDED_RAIN_INTENSITY, grids, 8.0, 0.0, 0.0, 300.0)       # 32 m of grid for 128 m of domain
        d.close()
    finally:
        hp.set_log_sink(None)
    assert seen and seen[-1][0] == 2 and "does not cover" in seen[-1][1]


def test_bench_line_carries_the_contract_fields():
    """`python bench.py` (small grid, short run) prints exactly one JSON line with every field the driver and the judge
    read: the metric block, `config.workload`, `roofline` (bound / achieved / peak / unit / frac / traffic) and
    `cpu_baseline` (value / unit / cores / kind / sample)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "10", "--warmup", "3", "--cols", "640",
                          "--rows", "320", "--prewarm-s", "0.05"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout
    b = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in b, k
    assert b["n_gpus"] == 1 and b["steps"] == 10 and b["warmup"] == 3 and b["higher_is_better"] is True
    assert b["scaling"] == "weak" and b["vs_baseline"] is None and b["dtype"] == "f64" and b["data"] == "synthetic"
    assert "workload" in b["config"] and "model" not in b["config"]
    assert abs(b["value"] - 640 * 320 / (b["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * b["value"]
    r = b["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and "traffic" in r
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0
    c = b["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "Mcell-steps/s" and c["sample"]
