"""Is a single-domain run reproducible bit for bit?  (fp32 / fp64, rain on dry terrain, FAST; tools only)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hipims-ocl_amd")]
os.environ.setdefault("HIPIMS_MI_NO_TORCH", "1")
import hipims_mi as hp
from hipims_mi import synthetic as syn
cols, rows, steps = 300, 157, 90
for precision in ("f32", "f64"):
    real = np.float32 if precision == "f32" else np.float64
    st, bed, man, rain = syn.s_rain_rows(cols, rows, 0, rows, dx=2.0, dtype=real)
    for kernel in (hp.KERNEL_AUTO, hp.KERNEL_BASIC):
        outs = []
        for rep in range(6):
            d = hp.Domain(cols, rows, dx=2.0, precision=precision, kernel=kernel)
            d.upload(st, bed, man)
            d.add_gridded(hp.GRIDDED_RAIN_INTENSITY, rain["grids"], rain["resolution"], rain["off_x"], rain["off_y"], rain["interval"])
            d.set_target_time(1e9)
            d.update_timestep()
            for n in (1, 2, steps - 3):
                d.step_batch(n)
            outs.append(d.download()); d.close()
        diffs = [int((outs[0].view(np.uint8) != o.view(np.uint8)).any(axis=-1).sum()) if False else int((outs[0] != o).sum()) for o in outs[1:]]
        where = np.argwhere((outs[0] != outs[[i for i, v in enumerate(diffs) if v][0] + 1]).any(axis=-1))[:6].tolist() if any(diffs) else []
        print(precision, "kernel", kernel, "values differing from run 0:", diffs, where, flush=True)
