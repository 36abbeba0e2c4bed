"""Front end: HFA raster reader, XML/CSV configuration walk, input rounding, output derivations --
on the CPU with the oracle as the engine."""
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN
from hipims_mi import frontend, hfa
from model_dir import make_newcastle


def test_hfa_reader_decodes_the_example_dem():
    a, info = hfa.read_hfa(os.path.join(GOLDEN, "NewcastleCentreDEM_2m.img"))
    assert (info["cols"], info["rows"]) == (342, 195) and info["pixel_size"] == (2.0, 2.0)
    assert a.dtype == np.float32 and a.shape == (195, 342) and not np.isnan(a).any()
    # statistics GDAL recorded next to the file in the reference's example (NewcastleCentreDEM_2m.img.aux.xml)
    assert a.min() == 43.4375 and abs(float(a.max()) - 81.737503051758) < 1e-9
    assert abs(float(a.astype(np.float64).mean()) - 56.567615182304) < 1e-6
    assert abs(float(a.astype(np.float64).std()) - 6.2576080647993) < 1e-4
    south_up, _ = hfa.read_raster(os.path.join(GOLDEN, "NewcastleCentreDEM_2m.img"))
    assert np.array_equal(south_up[0], a[-1].astype(np.float64))


def test_util_round_matches_reference_semantics():
    # Util::round: ceil when fmod(v*1e4, 1) >= 0.5 else floor; fmod keeps the sign, so negatives go down (quirk Q10)
    v = np.array([1.23456, 1.23454, -1.23456, -1.23454, 2.0, -2.0, 0.00005])
    assert np.allclose(frontend.util_round(v), [1.2346, 1.2345, -1.2346, -1.2346, 2.0, -2.0, 0.0001], rtol=0, atol=1e-12)


def test_configuration_and_initial_conditions(tmp_path):
    xml = make_newcastle(tmp_path)
    cfg = frontend.parse_configuration(xml)
    assert (cfg.duration, cfg.output_frequency, cfg.precision, cfg.courant, cfg.friction) == (7200.0, 600.0, "f64", 0.5, True)
    assert cfg.closed_edges == {"north", "south", "east", "west"} and len(cfg.boundaries) == 2
    rain = next(b for b in cfg.boundaries if b.value == "rain-intensity")
    assert rain.series.shape == (6, 2) and rain.series[1, 0] - rain.series[0, 0] == 3600.0
    state, bed, man, res = frontend.build_domain(cfg)
    assert bed.shape == (195, 342) and res == 2.0 and (man == 0.03).all()
    assert (bed[0] == 9999.9).all() and (bed[:, -1] == 9999.9).all()
    inner = bed[1:-1, 1:-1]
    assert np.array_equal(inner, frontend.util_round(inner)) and inner.min() > 43 and inner.max() < 82
    assert np.array_equal(state[..., 0][1:-1, 1:-1], inner) and not state[..., 2:].any()      # dry, at rest


def _oracle_sim(cfg, cols, rows, res):
    return oracle.OracleSim(cols, rows, dx=res, scheme=cfg.scheme, very_small=cfg.dry_threshold, courant=cfg.courant,
                            end_time=cfg.duration, friction=cfg.friction, threads=4)


def test_run_model_writes_reference_outputs(tmp_path):
    xml = make_newcastle(tmp_path, duration=120, frequency=60)
    results = frontend.run_model(xml, make_sim=_oracle_sim, batch=50)
    assert [round(t, 6) for t, _ in results] == [60.0, 120.0]
    t, out = results[-1]
    assert set(out) == {"depth", "velocityx", "velocityy", "fsl", "maxdepth"}
    depth = out["depth"]
    wet = depth != frontend.NODATA
    # 120 s of 70 - 12 mm/h net rain = 1.93 mm mean over the cells that receive rain (the boundary kernels skip the last
    # `n mod 8` columns/rows, quirk Q9), redistributed by the flow
    assert 0.5 < wet.mean() <= 1.0 and 1e-4 < depth[wet].mean() < 1e-2
    assert (depth[0] == frontend.NODATA).all()                                  # wall cells hold no water
    files = sorted(os.listdir(os.path.join(str(tmp_path), "output")))
    assert "depth_60.npy" in files and "velX_120.npy" in files and len(files) == 10
    saved = np.load(os.path.join(str(tmp_path), "output", "depth_120.npy"))
    assert np.array_equal(saved, depth)


# ---- orchestrator parity: hipims_mi/model.py ----
def test_value_codes_follow_the_reference_substring_order():
    codes = {n: frontend.data_value_code(n) for n in
             ("depth", "MaxDepth", "fsl", "maxfsl", "velocityX", "dischargey", "froude", "dem", "manningcoefficient", "x")}
    assert codes == {"depth": "depth", "MaxDepth": "maxdepth", "fsl": "fsl", "maxfsl": "maxfsl", "velocityX": "velocityx",
                     "dischargey": "dischargey", "froude": "froude", "dem": "dem",
                     "manningcoefficient": "manningcoefficient", "x": None}
    st = np.zeros((2, 2, 4)); st[..., 0] = 1.5; st[..., 1] = 2.0; st[..., 2] = 0.3; st[..., 3] = -0.4
    bed = np.full((2, 2), 1.0)
    assert np.allclose(frontend.derive_output("dischargex", st, bed, 2.0), 0.6)       # Q * resolution (:222-233)
    assert np.allclose(frontend.derive_output("froude", st, bed), 0.5 / 0.5 / np.sqrt(9.81 * 0.5))
    assert np.allclose(frontend.derive_output("maxdepth", st, bed), 1.0) and np.allclose(frontend.derive_output("maxfsl", st, bed), 2.0)


class _Clock:
    """Deterministic stand-in for the wall clock: every reading advances by `tick` seconds."""
    def __init__(self, tick):
        self.t, self.tick = 0.0, tick

    def __call__(self):
        self.t += self.tick
        return self.t


def test_orchestrator_targets_batches_progress_and_outputs(tmp_path):
    from hipims_mi.model import Model
    xml = make_newcastle(tmp_path, duration=90, frequency=30)
    lines = []
    m = Model(xml, make_sim=_oracle_sim, log=lines.append, clock=_Clock(0.01), progress_interval=0.085)
    outs = m.run()
    # outputs exactly at the multiples of the output frequency; the device-side sync clipping lands on them
    assert [round(t, 9) for t, _ in outs] == [30.0, 60.0, 90.0]
    assert m.last_output_time == m.current_time == 90.0 and m.target_time == 90.0
    sc = m.sim.scalars()
    assert sc["t"] == 90.0
    # the autotuner aims for a second of work per batch: with 0.01 s per clock reading it grows from 1, at most
    # doubling once past 40 and never beyond 3x the iterations the previous batch completed (:1420-1448)
    sizes = [b["batch_size"] for b in m.progress_blocks]
    assert sizes[0] >= 1 and max(sizes) > 8 and all(b <= 3 * max(1, a) or b <= 40 for a, b in zip(sizes, sizes[1:]))
    last = m.progress_blocks[-1]
    assert last["progress"] == 1.0 and last["cells_calculated"] == m.scheme.iterations * 342 * 195
    assert last["rate"] == int(last["cells_calculated"] / last["processing_time"])
    assert any("SIMULATION PROGRESS" in l for l in lines) and any("Output files written" in l for l in lines)

    # Batch boundaries are not physics-neutral IN THE REFERENCE: with quirk Q1 the timestep that follows a sync point is
    # priced on the pre- or post-step buffer depending on the parity of the iteration that landed on it and on
    # whether skipped iterations followed (the reference's wall-clock autotuner makes that non-deterministic there).
    # While dt is capped (t < 60 s: dt <= 0.1) nothing changes; afterwards the runs differ at rounding-of-dt level.
    fixed = frontend.run_model(make_newcastle(tmp_path / "b", duration=90, frequency=30), make_sim=_oracle_sim, batch=7)
    for (ta, a), (tb, b) in zip(outs[:2], fixed[:2]):
        assert ta == tb and all(np.array_equal(a[k], b[k]) for k in a)
    assert outs[2][0] == fixed[2][0] == 90.0 and np.abs(outs[2][1]["fsl"] - fixed[2][1]["fsl"]).max() < 1e-4
    again = frontend.run_model(make_newcastle(tmp_path / "c", duration=90, frequency=30), make_sim=_oracle_sim, batch=100)
    assert all(np.array_equal(fixed[2][1][k], again[2][1][k]) for k in fixed[2][1])


def test_orchestrator_matches_a_plain_run_to_the_same_sync_points(tmp_path):
    """Same state as driving the engine by hand: set the target, step until it is reached, move the target."""
    from hipims_mi.model import Model
    xml = make_newcastle(tmp_path, duration=40, frequency=20)
    m = Model(xml, make_sim=_oracle_sim, clock=_Clock(0.05))
    m.run()
    cfg = frontend.parse_configuration(xml)
    st, bed, man, res = frontend.build_domain(cfg)
    sim = _oracle_sim(cfg, 342, 195, res)
    sim.upload(st, bed, man)
    frontend.attach_boundaries(cfg, sim, 342)
    for target in (20.0, 40.0):
        sim.set_target(target)
        if sim.scalars()["dt"] <= 0:
            sim.update_timestep()
        while target - sim.scalars()["t"] > 1e-5:
            sim.run(5)
    assert np.array_equal(m.sim.download(), sim.download())


def test_cli_fails_loudly_without_a_gpu(tmp_path):
    """`python -m hipims_mi -c ...` is the product path: no CPU fallback, a clear error instead."""
    import subprocess, sys
    try:
        import torch
        if torch.cuda.is_available():
            import pytest
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    xml = make_newcastle(tmp_path, duration=10, frequency=10)
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(os.path.dirname(os.path.dirname(__file__)), "hipims-ocl_amd"),
                                                        os.environ.get("PYTHONPATH", "")]), HIPIMS_MI_NO_TORCH="1")
    r = subprocess.run([sys.executable, "-m", "hipims_mi", "-c", xml, "-s"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and ("no usable HIP device" in r.stderr or "HP_ERR" in r.stderr or "hip" in r.stderr.lower())


REFERENCE_EXAMPLE = "/root/reference/test/newcastle-centre.xml"


@pytest.mark.skipif(not os.path.exists(REFERENCE_EXAMPLE), reason="reference checkout not present (only in the build container)")
def test_the_reference_example_directory_parses_as_is(tmp_path):
    """Drop-in check on the reference's OWN model directory, untouched (read-only): its XML, its CSV series and its HFA
    raster give the same configuration and initial conditions as the fixture directory the other tests build."""
    ref = frontend.parse_configuration(REFERENCE_EXAMPLE)
    mine = frontend.parse_configuration(make_newcastle(tmp_path))
    assert (ref.duration, ref.output_frequency, ref.precision, ref.scheme, ref.courant, ref.friction, ref.dry_threshold) == \
           (mine.duration, mine.output_frequency, mine.precision, mine.scheme, mine.courant, mine.friction, mine.dry_threshold)
    assert ref.closed_edges == mine.closed_edges and len(ref.boundaries) == len(mine.boundaries) == 2
    for a, b in zip(sorted(ref.boundaries, key=lambda x: x.value), sorted(mine.boundaries, key=lambda x: x.value)):
        assert a.kind == b.kind and a.value == b.value and np.array_equal(a.series, b.series)
    assert [w for w, _ in ref.targets] == [w for w, _ in mine.targets]
    sa, ba, ma, ra = frontend.build_domain(ref)
    sb, bb, mb, rb = frontend.build_domain(mine)
    assert ra == rb and np.array_equal(ba, bb) and np.array_equal(ma, mb) and np.array_equal(sa, sb)


def test_hfa_writer_round_trips_and_speaks_the_reference_files_dictionary(tmp_path):
    """format="HFA" outputs (the reference's example asks for them, CRasterDataset.cpp:101-183): what the writer produces is
    read back bit for bit with georeferencing, and every MIF type definition it declares appears verbatim in the dictionary
    of the reference's own DEM file (tests/golden/NewcastleCentreDEM_2m.img, written by GDAL) -- the syntax check available
    without GDAL."""
    import re
    import struct
    from hipims_mi import hfa
    rng = np.random.default_rng(3)
    a = rng.normal(size=(195, 342))
    a[7, 11] = -9999.0
    path = str(tmp_path / "depth_600.img")
    frontend.write_raster(path, a, 2.0, origin=(424000.0, 564000.0))
    b, info = frontend.read_raster(path)
    assert np.array_equal(a, b) and info["pixel_type"] == 10 and info["block"] == (64, 64)
    assert info["pixel_size"] == (2.0, 2.0)
    assert info["upper_left_center"] == (424001.0, 564000.0 + 2.0 * 195 - 1.0)       # top-left corner + half a pixel
    assert info["lower_right_center"] == (424000.0 + 2.0 * 342 - 1.0, 564001.0)
    data = open(os.path.join(GOLDEN, "NewcastleCentreDEM_2m.img"), "rb").read()
    hdr = struct.unpack_from("<I", data, 16)[0]
    dic = struct.unpack_from("<iIIhI", data, hdr)[4]
    real = data[dic:data.index(b"\0", dic)].decode("latin1")
    mine = re.findall(r"\{[^{}]*\}[A-Za-z_0-9]+,", hfa._DICTIONARY)
    assert len(mine) == 14 and all(d in real for d in mine)
    # a ragged size: partial blocks at the right and bottom edges
    c = rng.normal(size=(70, 130))
    hfa.write_hfa(str(tmp_path / "c.img"), c, (0.0, 140.0), 2.0)
    d, _ = hfa.read_hfa(str(tmp_path / "c.img"))
    assert np.array_equal(c, d)
