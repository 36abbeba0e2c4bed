"""Worker of tests/test_gpu_two_step.py (HP_TWO_STEP is read once per process): runs a scenario on the FAST engine and writes the
final state, the time-control scalars and the number of iteration pairs the engine ran to an .npz.
usage: two_step_worker.py <scenario> <precision> <out.npz>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hipims-ocl_amd")]
import hipims_mi as hp  # noqa: E402
from hipims_mi import synthetic as syn  # noqa: E402

scenario, precision, out = sys.argv[1:4]
real = np.float64 if precision == "f64" else np.float32
kw = {}
if scenario == "dam":                       # all wet, walls: the benchmark's shape
    cols, rows = 1030, 700
    st, bed, man = syn.s_dam(cols, rows, dtype=real)
    plan = [("run", 40), ("run", 7), ("download",), ("run", 64), ("run", 1), ("run", 30)]
elif scenario == "rough":                   # wet/dry rough terrain with a Manning array: untouched cells, stopping conditions, friction
    cols, rows = 500, 333
    st, bed, man = syn.s_rough(cols, rows, dtype=real, manning=None)
    plan = [("run", 50), ("run", 33), ("download",), ("run", 100), ("run", 2), ("run", 61)]
elif scenario == "damdry":                  # dam break onto a dry bed + a sync point inside the run (clipped and suspended iterations)
    cols, rows = 600, 120
    st, bed, man = syn.s_dam(cols, rows, dtype=real, wet_right=False)
    plan = [("target", 1.0), ("run", 64), ("run", 64), ("target", 2.5), ("update",), ("run", 90), ("save",), ("run", 20), ("restore",), ("run", 20)]
elif scenario == "fixed":                   # fixed timestep (no reduction at all)
    cols, rows = 400, 300
    st, bed, man = syn.s_rough(cols, rows, dtype=real, manning=0.03)
    kw = dict(dynamic_dt=False, dt_fixed=0.01)
    plan = [("run", 41), ("run", 40)]
elif scenario == "rainlater":               # pairs first, then a rain boundary arrives: the first single iteration behind the pairs is K1 with FUSED
    cols, rows = 500, 333                   # boundaries AND the FILL flag (the buffer it writes is two states old); dry land keeps untouched cells
    st, bed, man = syn.s_rough(cols, rows, dtype=real, manning=0.03)
    plan = [("run", 24), ("rain",), ("run", 9), ("run", 30), ("download",), ("run", 5)]
else:
    raise SystemExit(scenario)
dom = hp.Domain(cols, rows, precision=precision, math_mode=hp.MATH_FAST, **kw)
dom.upload(st, bed, man)
dom.set_target_time(1e9)
for step in plan:
    if step[0] == "run":
        dom.step_batch(step[1])
    elif step[0] == "download":
        dom.download()
    elif step[0] == "target":
        dom.set_target_time(step[1])
    elif step[0] == "update":
        dom.update_timestep()
    elif step[0] == "save":
        dom.state_save()
    elif step[0] == "restore":
        dom.state_restore()
    elif step[0] == "rain":
        dom.add_uniform(hp.UNIFORM_RAIN_INTENSITY, np.array([[0.0, 90.0], [5.0, 30.0], [10.0, 0.0]]), 5.0, 10.0)
final = dom.download()
sc = dom.read_scalars()
counts = dom.launch_counts()
np.savez(out, state=final, t=sc["time"], dt=sc["timestep"], ok=sc["batch_successful"], skipped=sc["batch_skipped"],
         iterations=sc["iterations"], launches=counts[0])
dom.close()
print(f"{scenario} {precision}: t = {sc['time']!r}, iterations {sc['iterations']}, flux launches {counts[0]}")
