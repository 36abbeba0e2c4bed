#!/usr/bin/env python3
"""Step time of small grids against the tile height (HP_MARCH_RSEG read at domain creation)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hipims-ocl_amd"))
os.environ.setdefault("HIPIMS_MI_NO_TORCH", "1")
import hipims_mi as hp
from hipims_mi import synthetic as syn
for cols, rows in ((342, 195), (512, 512), (1024, 1024), (2048, 2048), (4096, 1026)):
    st, bed, man = syn.s_dam(cols, rows)
    line = []
    for rseg in (1, 2, 4, 8, 16):
        os.environ["HP_MARCH_RSEG"] = str(rseg)
        d = hp.Domain(cols, rows); d.upload(st, bed, man); d.set_target_time(1e9)
        d.step_batch(200); d.sync(); t0 = time.perf_counter(); d.step_batch(2000); d.sync()
        line.append("rseg %2d: %6.1f us" % (rseg, (time.perf_counter() - t0) / 2000 * 1e6)); d.close()
    print("%5dx%-5d " % (cols, rows) + "  ".join(line), flush=True)
