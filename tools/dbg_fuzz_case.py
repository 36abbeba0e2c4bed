#!/usr/bin/env python3
"""Run one seed of tests/test_gpu_fuzz_strict.py's STRICT-vs-oracle test outside pytest and say WHERE the first mismatch is.
usage: dbg_fuzz_case.py <seed>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hipims-ocl_amd"), os.path.join(ROOT, "tests")]
os.environ["HIPIMS_MI_NO_TORCH"] = "1"
import numpy as np
import test_gpu_fuzz_strict as fz
orig = np.array_equal
def spy(a, b, **k):
    r = orig(a, b, **k)
    if not r and np.asarray(a).ndim == 3:
        a, b = np.asarray(a), np.asarray(b)
        d = np.argwhere((a != b).any(axis=-1))
        print("MISMATCH in", len(d), "cells, first", d[:6].tolist(), "largest difference", np.nanmax(np.abs(a.astype(float) - b.astype(float))))
        y, x = d[0]
        print("  engine", a[y, x], "\n  oracle", b[y, x])
    return r
fz.np.array_equal = spy
c = fz.make_case(int(sys.argv[1]))
print("ops", c["ops"], "cuts", c["cuts"], "manning varies:", bool(c["man"].std() > 0), "kernel", c["kernel"], c["kw"])
try:
    fz.test_strict_engine_equals_the_oracle_on_a_random_configuration(int(sys.argv[1]))
    print("passed")
except AssertionError as e:
    print("ASSERT", str(e)[:400])
