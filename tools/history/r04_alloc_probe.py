#!/usr/bin/env python3
"""Round 4: is the headline launch bimodal (0.247 / 0.257 ms) because of where the buffers land?  One process creates the 4096^2
domain again and again -- sometimes with a dummy allocation in front that shifts every later address -- and times the same steps."""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "hipims-ocl_amd"))
os.environ["HIPIMS_MI_NO_TORCH"] = "1"
import numpy as np
import hipims_mi as hp
from hipims_mi import synthetic as syn
hip = ctypes.CDLL("libamdhip64.so")
cols = rows = 4096
st, bed, man = syn.s_dam(cols, rows, dtype=np.float64)
keep = []
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    pad_mb = [0, 0, 1, 3, 7, 33, 64, 100, 2, 5, 129, 257][trial % 12]
    if pad_mb:
        p = ctypes.c_void_p()
        hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(pad_mb * 1024 * 1024 + 4096 * trial))
        keep.append(p)
    if len(sys.argv) > 2 and trial in (3, 4, 8):            # an extra stream in front of these trials: does the pattern follow the streams?
        sp = ctypes.c_void_p(); hip.hipStreamCreateWithFlags(ctypes.byref(sp), 1); keep.append(sp)
        print("   (extra stream created)")
    d = hp.Domain(cols, rows)
    d.upload(st, bed, man); d.set_target_time(1e9); d.step_batch(20); d.sync()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); d.step_batch(200); d.sync(); ts.append((time.perf_counter() - t0) / 200 * 1e3)
    print("trial %2d  dummy allocation in front %3d MiB   %.4f %.4f %.4f ms per step" % (trial, pad_mb, *ts), flush=True)
    d.close()
