"""K1 / K2 / K6 launch time at 4096^2 on DRY land (S-RAIN's terrain without rain) and on a lake at rest."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hipims-ocl_amd")]
import hipims_mi as hp
from hipims_mi import synthetic as syn
N = 4096
def run(name, scheme, st, bed, man, dx=1.0, steps=200, precision="f64"):
    d = hp.Domain(N, N, dx=dx, scheme=scheme, precision=precision)
    d.upload(st, bed, man); d.set_target_time(1e9)
    d.step_batch(40); d.sync()
    t0 = time.perf_counter(); d.step_batch(steps); d.sync(); dt = (time.perf_counter() - t0) / steps
    print(f"{name:40s} {dt*1e3:.4f} ms/step  {N*N/dt/1e6:9.0f} Mcs/s", flush=True)
    d.close()
for prec, real in (("f64", np.float64), ("f32", np.float32)):
    st, bed, man, rain = syn.s_rain_rows(N, N, 0, N, dx=2.0, dtype=real)
    for sname, scheme in (("godunov", hp.SCHEME_GODUNOV), ("muscl", hp.SCHEME_MUSCL_HANCOCK), ("inertial", hp.SCHEME_INERTIAL)):
        run(f"{sname} {prec} dry land", scheme, st, bed, man, dx=2.0, precision=prec)
