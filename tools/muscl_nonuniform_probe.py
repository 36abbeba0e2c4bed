#!/usr/bin/env python3
"""MUSCL step time with a spatially varying Manning array (the non-uniform instantiation) for two library builds."""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    os.environ["HIPIMS_MI_LIB"] = os.path.join(ROOT, "hipims-ocl_amd", "lib", sys.argv[1])
    os.environ.setdefault("HIPIMS_MI_NO_TORCH", "1")
    sys.path.insert(0, os.path.join(ROOT, "hipims-ocl_amd"))
    import numpy as np, hipims_mi as hp
    from hipims_mi import synthetic as syn
    st, bed, man = syn.s_dam(4096, 2048)
    man = man + np.random.default_rng(1).uniform(0, 0.01, man.shape)
    d = hp.Domain(4096, 2048, scheme=hp.SCHEME_MUSCL_HANCOCK); d.upload(st, bed, man); d.set_target_time(1e9)
    d.step_batch(50); d.sync(); t0 = time.perf_counter(); d.step_batch(300); d.sync()
    print("NONUNIFORM", sys.argv[1], round((time.perf_counter() - t0) / 300 * 1e6, 1), "us/step", flush=True)
else:
    for lib in ("A.so", "B.so", "A.so", "B.so"):
        subprocess.run([sys.executable, __file__, lib])
