"""Worker of tests/test_gpu_strips.py::test_cxx_strip_loop_with_ranks_that_are_processes: ONE RANK of a strip run whose ranks
are separate processes sharing the box's one GPU.  The ghost rows and the wave-speed maxima travel the way they do between
GPUs in production -- written by this rank's advance kernel into the neighbours' state buffers and mailboxes, here through
IPC mappings of another process's memory (hp_strip_peer_*, level 2) -- and the collective library (the tests' double in
its process mode: all-reduce only) is left with the start-of-batch handshake.
usage: strip_procs_worker.py <rank> <world> <directory> <scheme> <f64|f32> <rain 0|1> <exchange period>"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hipims-ocl_amd")]
os.environ["HIPIMS_MI_NO_TORCH"] = "1"
import numpy as np  # noqa: E402

import hipims_mi as hp  # noqa: E402
from hipims_mi import strips, synthetic as syn  # noqa: E402

rank, world, where, scheme, precision, rain_on, period = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5], int(sys.argv[6]), int(sys.argv[7])
cols, rows, steps = (int(v) for v in os.environ.get("STRIP_WORKER_GRID", "300,157,90").split(","))
real = np.float64 if precision == "f64" else np.float32
g = strips.ghost_rows(scheme) * period
if rain_on:
    st, bed, man, rain = syn.s_rain_rows(cols, rows, 0, rows, dx=2.0, dtype=real)
    dx = 2.0
else:
    st, bed, man = syn.s_rough(cols, rows, dtype=real)
    rain, dx = None, 1.0
own_lo, own_hi, lo, hi = strips.partition(rows, world, g)[rank]


def gather(name, blob):
    """Every rank's blob in rank order: through files (the host's own means)."""
    with open(os.path.join(where, f"{name}.{rank}.tmp"), "wb") as f:
        f.write(blob)
    os.rename(os.path.join(where, f"{name}.{rank}.tmp"), os.path.join(where, f"{name}.{rank}"))
    out, deadline = [], time.time() + 120
    for r in range(world):
        path = os.path.join(where, f"{name}.{r}")
        while not os.path.exists(path):
            if time.time() > deadline:
                print("FAILED: nothing from rank", r, flush=True); sys.exit(3)
            time.sleep(0.01)
        out.append(open(path, "rb").read())
    return out


lib = hp.load_library()
hp._check(lib, lib.hp_comm_load(os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so").encode()), "hp_comm_load")
dom = hp.Domain(cols, hi - lo, dx=dx, scheme=scheme, precision=precision, global_rows=rows, row_offset=lo, ghost_rows=g if period > 1 else 0)
dom.upload(st[lo:hi], bed[lo:hi], man[lo:hi])
if rain is not None:
    dom.add_gridded(hp.GRIDDED_RAIN_INTENSITY, rain["grids"], rain["resolution"], rain["off_x"], rain["off_y"], rain["interval"])
dom.strip_comm_init(gather("id", hp.comm_unique_id())[0], rank, world)
level = dom.strip_peer_connect(gather("ticket", dom.strip_peer_ticket()), rank)
info = dom.strip_info()
if level != 2 or not info["peer_halo"] or not info["peer_max"]:
    print("FAILED: protocol level", level, info, flush=True); sys.exit(4)
dom.set_target_time(1e9)
dom.strip_update_timestep()
for n in (1, 2, steps - 3):
    dom.strip_step_batch(n)
dom.sync()
sc = dom.read_scalars()
np.save(os.path.join(where, f"owned.{rank}.npy"), dom.download()[own_lo - lo:own_hi - lo])
print("rank", rank, "of", world, "level", level, "time %.17g dt %.17g" % (sc["time"], sc["timestep"]), "flux launches", dom.launch_counts()[0],
      "iterations", sc["iterations"], flush=True)
dom.strip_comm_destroy()
dom.close()
