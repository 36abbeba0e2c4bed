#!/usr/bin/env python3
"""The exact mode's choice between pairs and single iterations along a developing S-DAM flood: what the tuner measured, what it chose,
and what each flavour costs at that point (HP_PAIR_TUNE=0 HP_TWO_STEP=1 forces pairs, HP_PAIR_STRICT=0 single iterations).
usage: strict_tuner_probe.py [workload: s-dam|s-rough]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hipims-ocl_amd")]
import hipims_mi as hp  # noqa: E402
from hipims_mi import synthetic as syn  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "s-dam"
cols = rows = 4096
st, bed, man = (syn.s_dam(cols, rows, dtype=np.float64) if workload == "s-dam" else syn.s_rough(cols, rows, dtype=np.float64, manning=0.03))
dom = hp.Domain(cols, rows, math_mode=hp.MATH_STRICT)
dom.upload(st, bed, man)
dom.set_target_time(1e9)
done = 0
for chunk in [50, 200, 250, 250, 250, 250, 250, 250, 250]:
    dom.read_scalars()
    c0 = dom.launch_counts()
    t0 = time.perf_counter()
    dom.step_batch(chunk)
    dom.read_scalars()
    el = time.perf_counter() - t0
    c1 = dom.launch_counts()
    done += chunk
    ps = dom.pair_stats()
    print(f"{os.environ.get('LABEL', 'default'):14s} iterations {done - chunk:5d}..{done:5d}: {el / chunk * 1e3:.4f} ms/iteration, {c1[0] - c0[0]:4d} flux launches; tuner samples {ps['tune_samples']}, "
          f"switches {ps['tune_switches']}, prefers pairs {ps['prefers_pairs']}, pair / single per iteration {ps['pair_over_single']:.3f}")
dom.close()
