#!/usr/bin/env python3
"""Config C1 end to end on the GPU: the reference's example model directory (test/newcastle-centre.xml shape:
342 x 195 cells at 2 m from its own DEM file, 70 mm/h rain + 12 mm/h drainage, 7200 s, outputs every 600 s) through
`python -m hipims_mi`'s code path; prints wall time, iterations and the final wet area."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hipims-ocl_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from hipims_mi import frontend
from hipims_mi.model import Model
from model_dir import make_newcastle

with tempfile.TemporaryDirectory() as tmp:
    xml = make_newcastle(tmp)
    t0 = time.perf_counter()
    m = Model(xml, output_format=".asc")
    outs = m.run()
    el = time.perf_counter() - t0
    depth = outs[-1][1]["depth"]
    wet = depth != frontend.NODATA
    print(f"C1: {len(outs)} outputs to t = {outs[-1][0]:.0f} s in {el:.2f} s wall ({m.seconds:.2f} s in the run loop), "
          f"{m.scheme.iterations} iterations ({m.scheme.batch_successful} successful), "
          f"rate {m.scheme.cells_calculated / m.seconds / 1e6:.0f} Mcell-steps/s, "
          f"wet cells {wet.mean() * 100:.1f} %, mean depth {depth[wet].mean() * 1000:.2f} mm, max {depth[wet].max():.3f} m, "
          f"last batch size {m.scheme.queue_addition_size}")
    m.close()
