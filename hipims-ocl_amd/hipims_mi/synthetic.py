"""Synthetic inputs of SURVEY.md section 8(d), shared by bench.py and the tests.

All arrays follow the reference's host layout (src/Domain/CDomain.h:26-33, CDomain.cpp:171-180):
state ``[rows][cols][4] = {Z (free-surface level), Zmax, Qx, Qy}``, row 0 = south;
``bed[rows][cols]``; ``manning[rows][cols]``.
"""
from __future__ import annotations

import numpy as np

WALL_BED = 9999.9     # closed-edge bed elevation, src/Domain/Cartesian/CDomainCartesian.cpp:790-796


def _walls(state, bed):
    """Outer ring as imposeBoundaryModification leaves it: bed = 9999.9, Z = 0."""
    for sl in (np.s_[0, :], np.s_[-1, :], np.s_[:, 0], np.s_[:, -1]):
        bed[sl] = WALL_BED
        state[sl] = 0.0
    return state, bed


def round4(x):
    """Input rounding of the reference (util.cpp:69-82 via CDomain.cpp:304-395): 4 decimal places."""
    return np.round(np.asarray(x, dtype=np.float64) * 1e4) / 1e4


def s_dam(cols, rows, dtype=np.float64, wet_right=True, manning=0.03, levels=(10.0, 1.0)):
    """S-DAM: flat bed, Z = 10 m for x < cols/2 else 1 m (all wet); S-DAM-DRY: right half dry.  `levels` changes the
    two water levels (the partial-inertial scheme is only meaningful for a gentle step)."""
    state = np.zeros((rows, cols, 4), dtype)
    bed = np.zeros((rows, cols), dtype)
    state[:, : cols // 2, 0] = levels[0]
    state[:, cols // 2:, 0] = levels[1] if wet_right else 0.0
    state[..., 1] = state[..., 0]
    _walls(state, bed)
    return state, bed, np.full((rows, cols), manning, dtype)


def s_rough(cols, rows, dtype=np.float64, seed=20240917, amplitude=0.5, pool_level=0.35, manning=0.03,
            walls=True):
    """S-ROUGH: undulating bed + noise, partially flooded (wet/dry fronts, bed steps), moving water."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:rows, 0:cols].astype(np.float64)
    bed = amplitude * np.sin(2 * np.pi * x / 37.0) * np.cos(2 * np.pi * y / 29.0) + rng.uniform(-0.01, 0.01, (rows, cols))
    bed = round4(bed)
    level = np.where(x < cols / 2, pool_level + 0.4, pool_level)
    z = np.maximum(bed, level)
    state = np.zeros((rows, cols, 4), np.float64)
    state[..., 0] = round4(z)
    state[..., 1] = state[..., 0]
    wet = (state[..., 0] - bed) > 1e-3
    state[..., 2] = round4(np.where(wet, rng.uniform(-0.2, 0.2, (rows, cols)), 0.0))
    state[..., 3] = round4(np.where(wet, rng.uniform(-0.2, 0.2, (rows, cols)), 0.0))
    man = round4(rng.uniform(0.02, 0.05, (rows, cols))) if manning is None else np.full((rows, cols), manning)
    if walls:
        _walls(state, bed)
    return state.astype(dtype), bed.astype(dtype), man.astype(dtype)


def s_rain(cols, rows, dx=2.0, dtype=np.float32, seed=7, grid_cells=64, slices=13, interval=300.0):
    """S-RAIN: initially dry undulating terrain + gridded rainfall stack (mm/h), config C5."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:rows, 0:cols].astype(np.float64)
    bed = round4(5.0 * 0.5 * np.sin(2 * np.pi * x / 37.0) * np.cos(2 * np.pi * y / 29.0)) + 10.0
    state = np.zeros((rows, cols, 4), np.float64)
    state[..., 0] = bed
    state[..., 1] = bed
    _walls(state, bed)
    resolution = dx * max(cols, rows) / grid_cells
    grids = rng.uniform(0.0, 120.0, (slices, grid_cells, grid_cells))
    rain = dict(grids=grids.astype(dtype), resolution=resolution, off_x=0.0, off_y=0.0, interval=interval)
    return state.astype(dtype), bed.astype(dtype), np.full((rows, cols), 0.03, dtype), rain


def s_rain_rows(cols, rows, row_lo, row_hi, dx=2.0, dtype=np.float32, seed=7, grid_cells=64, slices=13, interval=300.0):
    """Rows [row_lo, row_hi) of S-RAIN for a `cols x rows` grid (strip-wise construction for large grids)."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[row_lo:row_hi, 0:cols].astype(np.float64)
    bed = round4(5.0 * 0.5 * np.sin(2 * np.pi * x / 37.0) * np.cos(2 * np.pi * y / 29.0)) + 10.0
    n = row_hi - row_lo
    state = np.zeros((n, cols, 4), np.float64)
    state[..., 0] = bed
    state[..., 1] = bed
    bed[:, 0] = bed[:, -1] = WALL_BED
    state[:, 0] = 0.0
    state[:, -1] = 0.0
    if row_lo == 0:
        bed[0] = WALL_BED; state[0] = 0.0
    if row_hi == rows:
        bed[-1] = WALL_BED; state[-1] = 0.0
    resolution = dx * max(cols, rows) / grid_cells
    grids = rng.uniform(0.0, 120.0, (slices, grid_cells, grid_cells))
    rain = dict(grids=grids.astype(dtype), resolution=resolution, off_x=0.0, off_y=0.0, interval=interval)
    return state.astype(dtype), bed.astype(dtype), np.full((n, cols), 0.03, dtype), rain


def sloshing_bowl(n=200, dx=40.0, h0=10.0, a=3000.0, b=5.0, g=9.81):
    """Planar free surface sloshing round a frictionless parabolic bowl (Thacker's solution; the reference's model
    builder ships it as tools/model-builder/tests/TestSloshingBowl.js:92-118, there with rows counted north-down).
    Returns (bed, fsl(t), state(t), period): bed = h0 r^2 / a^2, eta = h0 - (b s / g)(x cos st + y sin st),
    (u, v) = (b sin st, -b cos st), s = sqrt(2 g h0) / a -- in this package's south-up, v-positive-north convention."""
    s = np.sqrt(2.0 * g * h0) / a
    xs = (np.arange(n) - n / 2 + 0.5) * dx
    x, y = np.meshgrid(xs, xs)
    bed = h0 * (x ** 2 + y ** 2) / a ** 2

    def fsl(t):
        f = h0 - (b * s / g) * (np.cos(s * t) * x + np.sin(s * t) * y)
        return np.where(f > bed, f, bed)

    def state(t):
        z = fsl(t)
        d = z - bed
        st = np.zeros((n, n, 4))
        st[..., 0] = z
        st[..., 1] = z
        st[..., 2] = d * b * np.sin(s * t)
        st[..., 3] = -d * b * np.cos(s * t)
        return st

    return bed, fsl, state, 2.0 * np.pi / s


def emerging_bed_dam_break(cols=640, rows=8, dx=0.05, h0=1.0, alpha=np.pi / 60.0, g=9.81):
    """Dam break onto a dry bed that rises at angle alpha (Xing et al. 2010; the reference's model builder ships it as
    tools/model-builder/tests/TestDamBreakEmergingBed.js:62-69): dam at x = 0, still water of level h0 to its left.
    Returns (state, bed, x of the cell centres, front(t)) with the analytic front position
    x_f = 2 t sqrt(g h0 cos alpha) - g t^2 tan(alpha) / 2."""
    xs = (np.arange(cols) - cols / 2 + 0.5) * dx
    bed = np.tile(xs * np.tan(alpha), (rows, 1))
    depth = np.where(xs <= 0, np.maximum(0.0, h0 - xs * np.tan(alpha)), 0.0)
    state = np.zeros((rows, cols, 4))
    state[..., 0] = bed + depth
    state[..., 1] = state[..., 0]
    _walls(state, bed)
    return state, bed, xs, (lambda t: 2 * t * np.sqrt(g * h0 * np.cos(alpha)) - 0.5 * g * t * t * np.tan(alpha))
