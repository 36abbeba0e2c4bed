import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hipims-ocl_amd")]
import hipims_mi as hp
from hipims_mi import synthetic as syn
# (a) rows identical (S-DAM): only column mix-ups show; (b) columns identical (transposed dam): only row mix-ups show
for name, mk in (("dam-x", lambda: syn.s_dam(128, 64)), ("dam-y", None)):
    if mk:
        st, bed, man = mk()
    else:
        st, bed, man = syn.s_dam(64, 128)
        st = np.ascontiguousarray(np.transpose(st, (1, 0, 2))[..., [0, 1, 3, 2]]); bed = np.ascontiguousarray(bed.T); man = np.ascontiguousarray(man.T)
    rows, cols = bed.shape
    outs = []
    for kernel in (hp.KERNEL_AUTO, hp.KERNEL_BASIC):
        d = hp.Domain(cols, rows, kernel=kernel); d.upload(st, bed, man); d.set_target_time(1e9); d.step_batch(1); outs.append(d.download()); d.close()
    diff = np.abs(outs[0] - outs[1]).max(axis=2)
    print(name, "max diff", diff.max(), "bad", (diff > 1e-9).sum())
    if diff.max() > 1e-9:
        ys, xs = np.nonzero(diff > 1e-9)
        print("  rows", np.unique(ys)[:20], "cols", np.unique(xs)[:20])
        for (y, x) in list(zip(ys, xs))[:4]:
            print("  cell", (x, y), "auto", outs[0][y, x], "basic", outs[1][y, x])
