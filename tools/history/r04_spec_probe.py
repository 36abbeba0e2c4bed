import os, sys, time
sys.path[:0] = ["/root/repo", "/root/repo/hipims-ocl_amd"]
os.environ["HIPIMS_MI_NO_TORCH"] = "1"
import numpy as np, hipims_mi as hp
from hipims_mi import synthetic as syn
logs = []
hp.set_log_sink(lambda l, t: logs.append(t))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
st, bed, man = syn.s_dam(n, n)
d = hp.Domain(n, n, math_mode=hp.MATH_STRICT)
d.upload(st, bed, man); d.set_target_time(1e9)
for b in (20, 200, 200, 200):
    t0 = time.perf_counter(); d.step_batch(b); d.sync(); el = time.perf_counter() - t0
    print(f"batch {b}: {el / b * 1e3:.4f} ms/iteration, replays so far {sum('re-run' in l for l in logs)}")
