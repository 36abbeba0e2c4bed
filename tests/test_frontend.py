"""Front end: HFA raster reader, XML/CSV configuration walk, input rounding, output derivations --
on the CPU with the oracle as the engine."""
import os

import numpy as np

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
    assert rain.series.shape == (4, 2) and rain.series[1, 0] - rain.series[0, 0] == 3600.0
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
