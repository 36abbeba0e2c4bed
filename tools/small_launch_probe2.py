import os, sys, time
sys.path.insert(0, "hipims-ocl_amd"); os.environ["HIPIMS_MI_NO_TORCH"] = "1"
import hipims_mi as hp
from hipims_mi import synthetic as syn
for scheme, name in ((hp.SCHEME_MUSCL_HANCOCK, "muscl"), (hp.SCHEME_INERTIAL, "inertial")):
    for cols, rows, steps in ((342, 195, 20000), (1024, 1024, 4000), (4096, 514, 2000), (4096, 4096, 300)):
        st, bed, man = syn.s_dam(cols, rows, levels=(2.0, 1.6) if name == "inertial" else (10.0, 1.0))
        d = hp.Domain(cols, rows, scheme=scheme); d.upload(st, bed, man); d.set_target_time(1e9)
        d.step_batch(100); d.sync()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); d.step_batch(steps); d.sync(); best = min(best, (time.perf_counter() - t0) / steps * 1e6)
        print("%-8s %5d x %4d: %7.2f us/iteration" % (name, cols, rows, best), "(HP_LAUNCH_TAIL=0)" if os.environ.get("HP_LAUNCH_TAIL") == "0" else "", flush=True)
        d.close()
