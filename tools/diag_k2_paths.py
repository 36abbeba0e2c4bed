"""K1 / K2 launch time at 4096^2 on three kinds of water: a still lake (every row quiet), the S-DAM dam break (one column
strip of tiles carries the front) and S-ROUGH (every tile on the general path).  Shows how much of a launch is the
slowest tiles rather than the average tile."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hipims-ocl_amd")]
import hipims_mi as hp
from hipims_mi import synthetic as syn

N = int(os.environ.get("N", 4096))
def run(name, scheme, st, bed, man, steps=200):
    d = hp.Domain(N, N, scheme=scheme)
    d.upload(st, bed, man); d.set_target_time(1e9)
    d.step_batch(60); d.sync()
    t0 = time.perf_counter(); d.step_batch(steps); d.sync(); dt = (time.perf_counter() - t0) / steps
    print(f"{name:34s} {dt*1e3:.4f} ms/step  {N*N/dt/1e6:9.0f} Mcs/s", flush=True)
    d.close()

for sname, scheme in (("godunov", hp.SCHEME_GODUNOV), ("muscl", hp.SCHEME_MUSCL_HANCOCK)):
    st, bed, man = syn.s_dam(N, N, levels=(10.0, 10.0))
    run(f"{sname} still lake", scheme, st, bed, man)
    st, bed, man = syn.s_dam(N, N)
    run(f"{sname} S-DAM", scheme, st, bed, man)
    st, bed, man = syn.s_dam(N, N); st = np.ascontiguousarray(st.transpose(1, 0, 2)); st[..., [2, 3]] = st[..., [3, 2]]
    run(f"{sname} S-DAM transposed (front along x)", scheme, st, bed, man)
    st, bed, man = syn.s_rough(N, N, manning=0.03)
    run(f"{sname} S-ROUGH", scheme, st, bed, man)
