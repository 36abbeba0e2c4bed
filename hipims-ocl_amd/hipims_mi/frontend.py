"""Front end for HiPIMS model directories: read the XML configuration, the CSV boundary series
and the rasters it names, run the model on the HIP engine at the reference's output times and write the reference's
output rasters -- without GDAL / boost / TinyXML.

What is mirrored (paths relative to the reference's src/):
  * configuration walk          Datasets/CXMLDataset.cpp:115-265, CModel.cpp:65-129 (duration, outputFrequency,
                                floatingPointPrecision), Schemes/CSchemeGodunov.cpp:128-333 + CScheme.cpp:60-135
                                (scheme parameters), Boundaries/CBoundaryMap.cpp:104-190 (timeseries elements)
  * initial conditions          Domain/Cartesian/CDomainCartesian.cpp:163-283 (DEM, then depth/FSL, then the rest),
                                CDomain.cpp:294-397 (4-decimal rounding of every input, quirk Q10), raster rows flipped
                                south-up (Datasets/CRasterDataset.cpp:411)
  * CSV series                  Boundaries/CBoundaryUniform.cpp:103-170 (header skipped, interval = t1 - t0,
                                length = last time), CBoundaryCell.cpp:164-300
  * main loop                   CModel.cpp:723-770, :870-891 (sync at every output time), CSchemeGodunov::runSimulation
  * output derivations          Datasets/CRasterDataset.cpp:185-267 (depth, velocity, fsl, maxdepth, maxfsl, froude;
                                threshold 1e-8, NODATA -9999)
Rasters: HFA .img through hipims_mi.hfa, ESRI ASCII .asc and NumPy .npy (read and write).
<domainEdge treatment="closed"> is honoured (bed = 9999.9 on that edge, CDomainCartesian.cpp:773-799); the reference
never parses the element (quirk Q9), so its own behaviour there is undefined.
"""
from __future__ import annotations

import csv
import os
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field

import numpy as np

from . import (DEPTH_IGNORE, DEPTH_IS_DEPTH, DEPTH_IS_FSL, DISCHARGE_IGNORE, DISCHARGE_IS_DISCHARGE,
               DISCHARGE_IS_VELOCITY, DISCHARGE_IS_VOLUME, SCHEME_GODUNOV, SCHEME_INERTIAL, SCHEME_MUSCL_HANCOCK,
               UNIFORM_LOSS_RATE,
               UNIFORM_RAIN_INTENSITY, hfa)

NODATA = -9999.0


def util_round(values, places=4):
    """Util::round (util.cpp:69-82): scale, then ceil if the fractional part (C fmod, sign of the dividend) is >= 0.5
    else floor -- so negative values round towards minus infinity unless they are integers."""
    v = np.asarray(values, dtype=np.float64) * (10 ** places)
    rem = np.fmod(v, 1.0)
    return np.where(rem >= 0.5, np.ceil(v), np.floor(v)) / (10 ** places)


# ------------------------------------------------------------------------------------------------ rasters
def read_raster(path):
    """-> (array[rows, cols] with row 0 = SOUTH, info{cols, rows, pixel_size?})"""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".img":
        return hfa.read_raster(path)
    if ext == ".npy":
        a = np.load(path).astype(np.float64)
        return a, dict(cols=a.shape[1], rows=a.shape[0])
    if ext == ".asc":
        hdr = {}
        with open(path) as f:
            for _ in range(6):
                k, v = f.readline().split()
                hdr[k.lower()] = float(v)
            a = np.loadtxt(f, dtype=np.float64)
        nod = hdr.get("nodata_value", NODATA)
        a = np.where(a == nod, NODATA, a)
        return np.ascontiguousarray(a[::-1]), dict(cols=int(hdr["ncols"]), rows=int(hdr["nrows"]),
                                                   pixel_size=(hdr["cellsize"], hdr["cellsize"]))
    raise ValueError(f"unsupported raster format: {path}")


def write_raster(path, south_up, resolution, origin=(0.0, 0.0)):
    ext = os.path.splitext(path)[1].lower()
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    if ext == ".npy":
        np.save(path, south_up)
    elif ext == ".img":                        # format="HFA" in the reference's XML (CDomainCartesian.cpp:300, CRasterDataset.cpp:101-183)
        hfa.write_raster(path, south_up, resolution, origin)
    elif ext == ".asc":
        with open(path, "w") as f:
            f.write(f"ncols {south_up.shape[1]}\nnrows {south_up.shape[0]}\nxllcorner {origin[0]}\nyllcorner {origin[1]}\n"
                    f"cellsize {resolution}\nNODATA_value {NODATA}\n")
            np.savetxt(f, south_up[::-1], fmt="%.10g")
    else:
        raise ValueError(f"unsupported output format: {path}")


# ------------------------------------------------------------------------------------------------ configuration
@dataclass
class Boundary:
    kind: str                 # "atmospheric" (uniform) | "cell"
    name: str
    value: str                # rain-intensity | loss-rate ; for cell: depthValue/dischargeValue pair
    series: np.ndarray        # [n, 2] or [n, 4]
    depth_value: str = "fsl"
    discharge_value: str = "total"
    cells: list = field(default_factory=list)      # (x, y) for cell boundaries


@dataclass
class Configuration:
    name: str = ""
    duration: float = 0.0
    output_frequency: float = 0.0
    precision: str = "f64"
    scheme: int = SCHEME_GODUNOV
    courant: float = 0.5
    dry_threshold: float = 1e-10
    friction: bool = True
    dynamic_dt: bool = True
    timestep: float = 0.001
    source_dir: str = ""
    target_dir: str = ""
    sources: list = field(default_factory=list)    # (type, value-list, source)
    targets: list = field(default_factory=list)    # (value, target pattern)
    closed_edges: set = field(default_factory=set)
    boundaries: list = field(default_factory=list)
    device_number: int = 1


def _params(elem):
    return {p.get("name").lower(): p.get("value") for p in elem.findall("parameter")}


def _read_csv(path):
    rows = []
    with open(path, newline="") as f:
        for i, row in enumerate(csv.reader(f)):
            if i == 0 or not row or all(not c.strip() for c in row):      # header skipped unconditionally (:103-110)
                continue
            rows.append([float(c) for c in row])
    return np.array(rows, np.float64)


def parse_configuration(xml_path):
    """Datasets/CXMLDataset.cpp:115-265 for the single-domain Cartesian case."""
    base = os.path.dirname(os.path.abspath(xml_path))
    root = ET.parse(xml_path).getroot()
    cfg = Configuration()
    meta = root.find("metadata")
    if meta is not None and meta.find("name") is not None:
        cfg.name = meta.find("name").text or ""
    sim = root.find("simulation")
    sp = _params(sim)
    cfg.duration = float(sp.get("duration", 0))                               # CModel.cpp:78-129
    cfg.output_frequency = float(sp.get("outputfrequency", cfg.duration))
    cfg.precision = "f32" if sp.get("floatingpointprecision", "double").lower() == "single" else "f64"
    dom = sim.find("domainSet").find("domain")
    if (dom.get("type") or "cartesian").lower() != "cartesian":
        raise ValueError("only cartesian domains")
    cfg.device_number = int(dom.get("deviceNumber") or 1)
    data = dom.find("data")
    cfg.source_dir = os.path.join(base, data.get("sourceDir") or "")
    cfg.target_dir = os.path.join(base, data.get("targetDir") or "")
    for ds in data.findall("dataSource"):
        cfg.sources.append(((ds.get("type") or "").lower(), [v.strip().lower() for v in (ds.get("value") or "").split(",")],
                            ds.get("source")))
    cfg.target_formats = []                  # GDAL driver names of the <dataTarget format=...> attributes (CDomainCartesian.cpp:300)
    for dt in data.findall("dataTarget"):
        cfg.targets.append(((dt.get("value") or "").lower(), dt.get("target")))
        cfg.target_formats.append((dt.get("format") or "").upper())
    sch = dom.find("scheme")
    name = (sch.get("name") or "godunov").lower()
    # CScheme::createFromConfig (CScheme.cpp:140-176): "muscl-hancock" | "godunov" | "inertial"
    cfg.scheme = {"muscl-hancock": SCHEME_MUSCL_HANCOCK, "muscl": SCHEME_MUSCL_HANCOCK,
                  "inertial": SCHEME_INERTIAL}.get(name, SCHEME_GODUNOV)
    sp = _params(sch)
    cfg.courant = float(sp.get("courantnumber", 0.5))                         # CSchemeGodunov.cpp:128-333
    cfg.dry_threshold = float(sp.get("drythreshold", 1e-10))
    cfg.friction = sp.get("frictioneffects", "yes").lower() in ("yes", "true", "1")
    if sp.get("timestepmode", "cfl").lower() == "fixed":
        cfg.dynamic_dt = False
    cfg.timestep = float(sp.get("timestepinitial", sp.get("timestepfixed", 0.001)))
    bc = dom.find("boundaryConditions")
    if bc is not None:
        bdir = os.path.join(base, bc.get("sourceDir") or "")
        for e in bc.findall("domainEdge"):
            if (e.get("treatment") or "").lower() == "closed":
                cfg.closed_edges.add((e.get("edge") or "").lower())
        for ts in bc.findall("timeseries"):
            kind = (ts.get("type") or "").lower()
            series = _read_csv(os.path.join(bdir, ts.get("source")))
            b = Boundary(kind=kind, name=ts.get("name") or "", value=(ts.get("value") or "rain-intensity").lower(),
                         series=series, depth_value=(ts.get("depthValue") or "fsl").lower(),
                         discharge_value=(ts.get("dischargeValue") or "total").lower())
            if kind == "cell" and ts.get("mapFile"):
                b.cells = [(int(r[0]), int(r[1])) for r in _read_csv(os.path.join(bdir, ts.get("mapFile")))]
            cfg.boundaries.append(b)
    return cfg


# ------------------------------------------------------------------------------------------------ domain arrays
def build_domain(cfg):
    """loadInitialConditions order: DEM, depth/FSL, everything else (CDomainCartesian.cpp:163-283)."""
    structure = next((s for s in cfg.sources if "structure" in s[1]), None) or next(s for s in cfg.sources if "dem" in s[1])
    ref, info = read_raster(os.path.join(cfg.source_dir, structure[2]))
    rows, cols = ref.shape
    res = float(info.get("pixel_size", (1.0, 1.0))[0])
    bed = np.zeros((rows, cols))
    state = np.zeros((rows, cols, 4))
    man = np.zeros((rows, cols))

    def values(src):
        if src[0] == "constant":
            return np.full((rows, cols), float(src[2]))
        return read_raster(os.path.join(cfg.source_dir, src[2]))[0]

    ordered = sorted(cfg.sources, key=lambda s: 0 if "dem" in s[1] else (1 if ({"depth", "fsl"} & set(s[1])) else 2))
    for src in ordered:
        v = None
        for what in src[1]:
            if what == "structure":
                continue
            v = values(src) if v is None else v
            if what == "dem":
                bed[...] = util_round(v); state[..., 0] = bed                     # CDomain.cpp:304-316
            elif what == "fsl":
                state[..., 0] = util_round(v); state[..., 1] = state[..., 0]
            elif what == "depth":
                state[..., 0] = util_round(bed + v); state[..., 1] = state[..., 0]
            elif what == "disabled":
                state[..., 1] = np.where((v > 1.0) & (v < 9999.0), -9999.0, state[..., 1])
            elif what == "dischargex":
                state[..., 2] = util_round(v)
            elif what == "dischargey":
                state[..., 3] = util_round(v)
            elif what == "velocityx":
                state[..., 2] = util_round(v * (state[..., 0] - bed))
            elif what == "velocityy":
                state[..., 3] = util_round(v * (state[..., 0] - bed))
            elif what == "manningcoefficient":
                man[...] = util_round(v)
    for edge in cfg.closed_edges:                                                 # CDomainCartesian.cpp:773-799
        sl = {"north": np.s_[-1, :], "south": np.s_[0, :], "east": np.s_[:, -1], "west": np.s_[:, 0]}[edge]
        bed[sl] = 9999.9
    return state, bed, man, res


def _cell_codes(b):
    depth = {"fsl": DEPTH_IS_FSL, "depth": DEPTH_IS_DEPTH, "ignore": DEPTH_IGNORE, "disabled": DEPTH_IGNORE}[b.depth_value]
    disc = {"total": DISCHARGE_IS_DISCHARGE, "cell": DISCHARGE_IS_DISCHARGE, "velocity": DISCHARGE_IS_VELOCITY,
            "ignore": DISCHARGE_IGNORE, "disabled": DISCHARGE_IGNORE, "volume": DISCHARGE_IS_VOLUME,
            "surging": DISCHARGE_IS_VOLUME}[b.discharge_value]
    return depth, disc


def attach_boundaries(cfg, sim, cols):
    """Works for hipims_mi.Domain and for the oracle's simulation objects (same add_* surface)."""
    for b in cfg.boundaries:
        s = b.series
        if b.kind == "atmospheric":
            interval, length = s[1, 0] - s[0, 0], s[-1, 0]                          # CBoundaryUniform.cpp:156-160
            definition = UNIFORM_LOSS_RATE if b.value == "loss-rate" else UNIFORM_RAIN_INTENSITY
            sim.add_uniform(definition, s[:, :2], interval, length)
        elif b.kind == "cell":
            depth, disc = _cell_codes(b)
            ser = s[:, :4].copy()
            if b.discharge_value == "total":                                        # CBoundaryCell.cpp:398-402
                ser[:, 2:4] /= max(1, len(b.cells))
            sim.add_cell(depth, disc, [y * cols + x for x, y in b.cells], ser, s[1, 0] - s[0, 0], s[-1, 0])
        else:
            raise ValueError(f"unsupported timeseries type {b.kind}")


# ------------------------------------------------------------------------------------------------ outputs
def data_value_code(name):
    """CDomain::getDataValueCode (CDomain.cpp:464-500): SUBSTRING matches, in the reference's order (so "maxdepth"
    wins over "depth", "maxfsl" over "fsl")."""
    n = (name or "").lower()
    if "dem" in n:
        return "dem"
    if "maxdepth" in n:
        return "maxdepth"
    if "depth" in n:
        return "depth"
    for key in ("disabled", "dischargex", "dischargey"):
        if key in n:
            return key
    if "maxfsl" in n:
        return "maxfsl"
    if "fsl" in n:
        return "fsl"
    for key in ("manningcoefficient", "velocityx", "velocityy", "froude"):
        if key in n:
            return key
    return None


def derive_output(what, state, bed, resolution=1.0):
    """Datasets/CRasterDataset.cpp:185-267."""
    code = data_value_code(what)
    z, zmax, qx, qy = (state[..., k].astype(np.float64) for k in range(4))
    bed = bed.astype(np.float64)
    depth = z - bed
    with np.errstate(divide="ignore", invalid="ignore"):
        if code == "depth":
            d = np.maximum(0.0, depth)
            return np.where(d < 1e-8, NODATA, d)
        if code == "maxdepth":
            d = np.maximum(0.0, zmax - bed)
            return np.where((d < 1e-8) | (d <= -9990.0) | (d >= 9999.0), NODATA, d)
        if code == "fsl":
            return np.where((z < bed + 1e-8) | (bed > 9999.0), NODATA, z)
        if code == "maxfsl":
            return np.where((zmax < bed + 1e-8) | (bed > 9999.0), NODATA, zmax)
        if code == "dischargex":
            return qx * resolution
        if code == "dischargey":
            return qy * resolution
        if code == "velocityx":
            return np.where(depth > 1e-8, qx / depth, NODATA)
        if code == "velocityy":
            return np.where(depth > 1e-8, qy / depth, NODATA)
        if code == "froude":
            return np.where(depth > 1e-8, np.sqrt((qx / depth) ** 2 + (qy / depth) ** 2) / np.sqrt(9.81 * depth), NODATA)
    raise ValueError(f"unknown output {what}")


def run_model(xml_path, make_sim=None, batch=None, output_format=".npy", max_outputs=None, log=None):
    """CModel::runModel for one domain (see model.py): advance to each output time, write the configured rasters
    there.  `batch` fixes the batch size (the reference's `queueMode="fixed"`); default = the autotuner.
    make_sim(cfg, cols, rows, res) may return any object with the Domain surface (the tests pass the oracle);
    the default is the HIP engine.  Returns [(time, {value: south-up array})]."""
    from .model import Model
    m = Model(xml_path, make_sim=make_sim, output_format=output_format, log=log)
    if batch:
        m.scheme.automatic_queue = False
        m.scheme.queue_addition_size = int(batch)
    try:
        return m.run(max_outputs=max_outputs)
    finally:
        m.close()
