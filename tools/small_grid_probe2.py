#!/usr/bin/env python3
"""Step time per scheme on small grids with the library's own choice of tile height."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hipims-ocl_amd"))
os.environ.setdefault("HIPIMS_MI_NO_TORCH", "1")
import hipims_mi as hp
from hipims_mi import synthetic as syn
for cols, rows in ((342, 195), (512, 512), (1024, 1024), (2048, 2048)):
    line = []
    for scheme, name in ((0, "godunov"), (1, "muscl"), (2, "inertial")):
        st, bed, man = syn.s_dam(cols, rows, levels=(2.0, 1.6))
        for env in ({}, {"HP_MARCH_RSEG": "16", "HP_MUSCL_RSEG": "32", "HP_INERTIAL_RSEG": "32"}):
            for k in ("HP_MARCH_RSEG", "HP_MUSCL_RSEG", "HP_INERTIAL_RSEG"):
                os.environ.pop(k, None)
            os.environ.update(env)
            d = hp.Domain(cols, rows, scheme=scheme); d.upload(st, bed, man); d.set_target_time(1e9)
            d.step_batch(200); d.sync(); t0 = time.perf_counter(); d.step_batch(2000); d.sync()
            line.append("%s %s %6.1f us" % (name, "fixed" if env else "auto ", (time.perf_counter() - t0) / 2000 * 1e6)); d.close()
    print("%5dx%-5d " % (cols, rows) + " | ".join(line), flush=True)
