#!/usr/bin/env python3
"""STRICT engine against the oracle, bit for bit, on grids TALL enough for the launch geometry of the big runs (XCD bands of
>= 256 rows, 18-row tiles, several rounds of blocks, many column strips) -- shapes the test suite's fuzz (<= 257 x 101) never
reaches.  Random shape, scheme, precision, rain, a few dozen iterations each (the oracle is a scalar CPU code).
With a third argument `wide`: 5000-17000 columns by 30-90 rows instead (hundreds of column strips per row of tiles).
usage: big_shape_fuzz.py <first seed> <count> [wide]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hipims-ocl_amd")]
os.environ["HIPIMS_MI_NO_TORCH"] = "1"
import numpy as np
import hipims_mi as hp
import oracle
from hipims_mi import synthetic as syn

first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    scheme = int(rng.choice([hp.SCHEME_GODUNOV, hp.SCHEME_GODUNOV, hp.SCHEME_MUSCL_HANCOCK, hp.SCHEME_INERTIAL]))
    precision = str(rng.choice(["f64", "f64", "f32"]))
    cols, rows = (int(rng.integers(90, 700)), int(rng.integers(2049, 2400))) if len(sys.argv) < 4 else (int(rng.integers(5000, 17000)), int(rng.integers(30, 90)))
    real = np.float64 if precision == "f64" else np.float32
    rain = scheme == hp.SCHEME_GODUNOV and rng.random() < 0.5
    if rain:
        st, bed, man, rn = syn.s_rain_rows(cols, rows, 0, rows, dx=2.0, dtype=real)
        dx = 2.0
    else:
        st, bed, man = syn.s_rough(cols, rows, dtype=real, seed=int(rng.integers(1, 10 ** 6)), manning=None if rng.random() < 0.5 else 0.03)
        dx = 1.0
    if scheme == hp.SCHEME_INERTIAL:
        st[..., 2:] *= 0.1
    iters = int(rng.integers(20, 45))
    oq = oracle.QUIRKS_REFERENCE & ~(oracle.Q6_MUSCL_SERIAL if scheme == hp.SCHEME_MUSCL_HANCOCK else 0)
    ref = oracle.OracleSim(cols, rows, dx=dx, scheme=scheme, precision=precision, quirks=oq)
    dom = hp.Domain(cols, rows, dx=dx, scheme=scheme, precision=precision, math_mode=hp.MATH_STRICT)
    for s in (ref, dom):
        s.upload(st, bed, man)
        if rain:
            s.add_gridded(hp.GRIDDED_RAIN_INTENSITY, rn["grids"], rn["resolution"], rn["off_x"], rn["off_y"], rn["interval"])
    dom.set_target_time(1e9); ref.set_target(1e9)
    t0 = time.time()
    ref.run(iters); dom.step_batch(iters)
    same = np.array_equal(dom.download(), ref.download())
    sc, sr = dom.read_scalars(), ref.scalars()
    same = same and sc["time"] == sr["t"] and sc["timestep"] == sr["dt"]
    bad += not same
    print("seed", seed, "ok" if same else "FAILED", "scheme", scheme, precision, "grid", (cols, rows), "rain" if rain else "rough", iters, "iterations, %.1f s" % (time.time() - t0), flush=True)
    dom.close()
print("failed:", bad, "of", count)
sys.exit(1 if bad else 0)
