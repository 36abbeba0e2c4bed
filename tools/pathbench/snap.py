"""GPU box: snapshots of real states for tools/pathbench (what the general paths are fed with).
Writes /tmp/pathbench_<name>.bin = int32 cols, rows + float64 state[rows][cols][4] + bed[rows][cols]."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hipims-ocl_amd")]
os.environ.setdefault("HIPIMS_MI_NO_TORCH", "1")
import hipims_mi as hp
from hipims_mi import synthetic as syn

N = 512
def snap(name, st, bed, man, steps, scheme=hp.SCHEME_GODUNOV, rain=None, dx=1.0):
    d = hp.Domain(N, N, dx=dx, scheme=scheme)
    d.upload(st, bed, man)
    if rain is not None:
        d.add_gridded(hp.GRIDDED_RAIN_INTENSITY, rain["grids"], rain["resolution"], rain["off_x"], rain["off_y"], rain["interval"])
    d.set_target_time(1e9)
    d.step_batch(steps)
    out = d.download()
    sc = d.read_scalars()
    depth = np.maximum(0, out[..., 0] - bed)
    with open(f"/tmp/pathbench_{name}.bin", "wb") as f:
        np.array([N, N], np.int32).tofile(f)
        out.astype(np.float64).tofile(f)
        bed.astype(np.float64).tofile(f)
        np.array([sc["timestep"]], np.float64).tofile(f)
    print(f"{name}: t = {sc['time']:.3f} s dt = {sc['timestep']:.4g}, wet {np.mean(depth > 1e-10):.3f}, mean depth of wet {depth[depth > 1e-10].mean():.4g} m, "
          f"max |q| {np.abs(out[..., 2:]).max():.3g}", flush=True)
    d.close()

st, bed, man, rain = syn.s_rain(N, N, dx=2.0, dtype=np.float64)
snap("srain", st, bed, man, 1500, rain=rain, dx=2.0)
st, bed, man = syn.s_rough(N, N, manning=0.03)
snap("srough", st, bed, man, 150)
snap("srough_muscl", st, bed, man, 150, scheme=hp.SCHEME_MUSCL_HANCOCK)
st, bed, man = syn.s_dam(N, N)
snap("sdam", st, bed, man, 400)
