import os, sys, time
sys.path.insert(0, "hipims-ocl_amd"); os.environ.setdefault("HIPIMS_MI_NO_TORCH", "1")
import numpy as np, hipims_mi as hp
from hipims_mi import synthetic as syn
for name, kw in (("s-dam", {}), ("s-dam-dry", dict(wet_right=False))):
    for scheme in (hp.SCHEME_GODUNOV, hp.SCHEME_MUSCL_HANCOCK):
        st, bed, man = syn.s_dam(4096, 4096, **kw)
        d = hp.Domain(4096, 4096, scheme=scheme); d.upload(st, bed, man); d.set_target_time(1e9)
        d.step_batch(50); d.sync(); t0 = time.perf_counter(); d.step_batch(300); d.sync(); el = time.perf_counter() - t0
        print(name, scheme, round(4096 * 4096 * 300 / el / 1e6), "Mcell-steps/s", flush=True); d.close()
