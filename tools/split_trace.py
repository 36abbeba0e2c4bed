"""rocprofv3 --kernel-trace target: 60 split-launch steps of a 16384x1026 strip (no communication)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hipims-ocl_amd"))
os.environ.setdefault("HIPIMS_MI_NO_TORCH", "1")
import hipims_mi as hp
from hipims_mi import synthetic as syn
cols, rows = 16384, 1026
st, bed, man = syn.s_dam(cols, rows)
d = hp.Domain(cols, rows, global_rows=rows + 2048, row_offset=1024)
d.upload(st, bed, man); d.set_target_time(1e9)
d.set_halo_overlap(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
for _ in range(60):
    d.step_begin(); d.step_end()
d.sync(); d.close()
