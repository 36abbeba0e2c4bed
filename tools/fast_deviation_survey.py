#!/usr/bin/env python3
"""How far does FAST arithmetic end up from STRICT (= the reference's kernels bit for bit, tests/test_gpu_fuzz_strict.py) over
the fuzz file's random configurations?  For every seed both flavours run the same case; reported: depth RMSE and largest
depth difference (depth = max(0, Z - zb), the parity protocol's quantity), relative difference of the simulated
time.  north_star's tolerance: fp64 depth RMSE < 1e-9 m, max < 1e-7 m; fp32 RMSE < 1e-4 m.
The yardstick next to it: STRICT run again from an input in which one wet cell in ten has its level moved by ONE ulp -- how far
the reference's own algorithm carries a last-bit difference over the same iterations (wet/dry fronts and thin films amplify).
usage: fast_deviation_survey.py <seeds>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hipims-ocl_amd"), os.path.join(ROOT, "tests")]
os.environ["HIPIMS_MI_NO_TORCH"] = "1"
import numpy as np
import hipims_mi as hp
import test_gpu_fuzz_strict as fz

n = int(sys.argv[1])
rows, outside = [], 0
for seed in range(n):
    c = fz.make_case(seed)
    out = []
    nudged = c["st"].copy()
    prng = np.random.default_rng(77 + seed)
    pick = (nudged[..., 0] - c["bed"] > 1e-3) & (nudged[..., 1] > -9000) & (prng.random(nudged.shape[:2]) < 0.1)
    nudged[..., 0][pick] = np.nextafter(nudged[..., 0][pick], np.inf).astype(nudged.dtype)
    for mode, start in ((hp.MATH_STRICT, c["st"]), (hp.MATH_FAST, c["st"]), (hp.MATH_STRICT, nudged)):
        dom = hp.Domain(c["cols"], c["rows"], dx=c["dx"], scheme=c["scheme"], precision=c["precision"], quirks=fz.oracle.quirks_to_engine(c["quirks"]),
                        friction=c["kw"]["friction"], dynamic_dt=c["kw"]["dynamic_dt"], dt_fixed=c["fixed_dt"],
                        dt_initial=c["fixed_dt"] if not c["kw"]["dynamic_dt"] else 0.001, math_mode=mode)
        dom.upload(start, c["bed"], c["man"]); fz.attach(dom, c["bdy"]); dom.set_target_time(c["target"])
        dom.step_batch(sum(c["cuts"]))
        out.append((dom.download().astype(np.float64), dom.read_scalars()["time"]))
        dom.close()
    (a, ta), (b, tb), (e, te) = out
    live = c["st"][..., 1] > -9000
    if not np.isfinite(a[live]).all():
        continue
    bed = c["bed"].astype(np.float64)
    da, db, de = (np.maximum(0.0, v[..., 0] - bed)[live] for v in (a, b, e))
    wet = da > 1e-6
    speed = float((np.sqrt(a[..., 2] ** 2 + a[..., 3] ** 2)[live][wet] / da[wet]).max()) if wet.any() else 0.0
    if da.max() > 50.0 or speed > 50.0:          # not a flood any more (a random mass-flux grid that pours kilometres of water in one
        outside += 1                             # gate opening, a frictionless thin film at 1e4 m/s): absolute tolerances mean nothing there
        continue
    rows.append((c["precision"], c["scheme"], float(np.sqrt(np.mean((da - db) ** 2))), float(np.abs(da - db).max()), abs(ta - tb) / max(abs(ta), 1e-30), seed, sum(c["cuts"]),
                 float(np.sqrt(np.mean((da - de) ** 2))) if np.isfinite(e[live]).all() else float("inf")))
print(f"{n} seeds; {outside} cases left out because the STRICT solution itself ends deeper than 50 m or faster than 50 m/s")
for prec in ("f64", "f32"):
    r = [x for x in rows if x[0] == prec]
    rm, mx, dt = np.array([x[2] for x in r]), np.array([x[3] for x in r]), np.array([x[4] for x in r])
    print(f"{prec}: {len(r)} cases, 40-260 iterations each")
    nd = np.array([x[7] for x in r])
    for name, v in (("depth RMSE [m]", rm), ("largest depth difference [m]", mx), ("relative difference of simulated time", dt),
                    ("depth RMSE of the one-ulp-nudged STRICT run [m]", nd)):
        print(f"  {name:40s} median {np.median(v):.2e}  90 % {np.quantile(v, 0.9):.2e}  99 % {np.quantile(v, 0.99):.2e}  worst {v.max():.2e}")
    worst = sorted(r, key=lambda x: -x[2])[:4]
    print("  worst by RMSE:", [(f"seed {x[5]}", f"scheme {x[1]}", f"{x[6]} iterations", f"rmse {x[2]:.2e}", f"max {x[3]:.2e}", f"nudged STRICT rmse {x[7]:.2e}") for x in worst])
    tol = 1e-9 if prec == "f64" else 1e-4
    over = [x for x in r if x[2] >= tol]
    print(f"  cases with FAST RMSE >= {tol:g} m: {len(over)}; of these the nudged STRICT run is itself >= {tol:g} m away in {sum(x[7] >= tol for x in over)}"
          f" and within a factor 100 of FAST's distance in {sum(x[7] * 100 >= x[2] for x in over)}")
    ratio = np.array([x[2] / x[7] for x in r if x[7] > 0 and np.isfinite(x[7])])
    print(f"  FAST distance / nudged-STRICT distance: median {np.median(ratio):.2g}  90 % {np.quantile(ratio, 0.9):.2g}  99 % {np.quantile(ratio, 0.99):.2g}  worst {ratio.max():.2g}")
