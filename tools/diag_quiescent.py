import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hipims-ocl_amd")]
import hipims_mi as hp
from hipims_mi import synthetic as syn
st, bed, man = syn.s_dam(512, 256)
d = hp.Domain(512, 256, scheme=hp.SCHEME_MUSCL_HANCOCK); d.upload(st, bed, man); d.set_target_time(1e9); d.step_batch(40)
o = d.download()
for name, sl in (("left 20:120", np.s_[20:200, 20:120]), ("right 400:490", np.s_[20:200, 400:490])):
    blk = o[sl]
    print(name, "z unique", np.unique(blk[..., 0])[:5], "qx absmax", np.abs(blk[..., 2]).max(), "qy absmax", np.abs(blk[..., 3]).max(), "zmax unique", np.unique(blk[..., 1])[:3])
print("t", d.read_scalars()["time"])
