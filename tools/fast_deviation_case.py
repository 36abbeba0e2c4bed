#!/usr/bin/env python3
"""One seed of fast_deviation_survey.py, looked at over time: the difference between FAST and STRICT every few iterations next
to the size of the STRICT solution itself (largest depth, largest speed).   usage: fast_deviation_case.py <seed> [<seed> ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hipims-ocl_amd"), os.path.join(ROOT, "tests")]
os.environ["HIPIMS_MI_NO_TORCH"] = "1"
import numpy as np
import hipims_mi as hp
import test_gpu_fuzz_strict as fz

for seed in map(int, sys.argv[1:]):
    c = fz.make_case(seed)
    print(f"seed {seed}: scheme {c['scheme']} {c['precision']} {c['cols']}x{c['rows']} dx {c['dx']} {c['kw']} fixed_dt {c['fixed_dt']} boundaries {[(b[0], b[1]) for b in c['bdy']]} target {c['target']:.3g} iterations {sum(c['cuts'])}")
    doms = []
    for mode in (hp.MATH_STRICT, hp.MATH_FAST):
        dom = hp.Domain(c["cols"], c["rows"], dx=c["dx"], scheme=c["scheme"], precision=c["precision"], quirks=fz.oracle.quirks_to_engine(c["quirks"]),
                        friction=c["kw"]["friction"], dynamic_dt=c["kw"]["dynamic_dt"], dt_fixed=c["fixed_dt"],
                        dt_initial=c["fixed_dt"] if not c["kw"]["dynamic_dt"] else 0.001, math_mode=mode)
        dom.upload(c["st"], c["bed"], c["man"]); fz.attach(dom, c["bdy"]); dom.set_target_time(c["target"])
        doms.append(dom)
    live = c["st"][..., 1] > -9000
    bed = c["bed"].astype(np.float64)
    total, done, step = sum(c["cuts"]), 0, max(1, sum(c["cuts"]) // 12)
    while done < total:
        n = min(step, total - done)
        for d in doms:
            d.step_batch(n)
        done += n
        a, b = (d.download().astype(np.float64) for d in doms)
        ha, hb = np.maximum(0.0, a[..., 0] - bed), np.maximum(0.0, b[..., 0] - bed)
        wet = live & (ha > 1e-6)
        speed = np.sqrt(a[..., 2] ** 2 + a[..., 3] ** 2)[wet] / ha[wet] if wet.any() else np.zeros(1)
        sa, sb = doms[0].read_scalars(), doms[1].read_scalars()
        print(f"  after {done:4d}: depth difference max {np.abs(ha - hb)[live].max():.2e} rmse {np.sqrt(np.mean((ha - hb)[live] ** 2)):.2e} | STRICT: deepest {ha[live].max():.3g} m, fastest {speed.max():.3g} m/s, "
              f"t {sa['time']:.6g} dt {sa['timestep']:.3g} | FAST: t {sb['time']:.6g} dt {sb['timestep']:.3g}")
    for d in doms:
        d.close()
