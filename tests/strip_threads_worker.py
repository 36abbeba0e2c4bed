"""Worker of tests/test_gpu_strips.py::test_cxx_strip_loop_with_several_ranks: runs hp_strip_step_batch -- the library's own
per-iteration strip loop -- with WORLD ranks that are threads of this process sharing the one GPU, over the in-process
test double of the collective library (tests/fake_rccl), and compares the gathered strips with the single domain
bit for bit.   usage: strip_threads_worker.py <world> <scheme 0|1|2> <f64|f32> <overlap 0|1> <rain 0|1> [exchange period 1|2]
[boundary on rank k only: -2 = no cell boundary] [2 = mailboxes and ghost rows written by the strips themselves (default), 1 = mailboxes for the maximum only, 0 = everything through the library]"""
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hipims-ocl_amd")]
os.environ["HIPIMS_MI_NO_TORCH"] = "1"
# every rank's streams on hardware queues of their own: a rank's advance kernel WAITS for the other ranks' kernels, which
# must not sit behind it in a shared queue (one process per GPU in production; here up to 8 ranks x 2 streams share one)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import numpy as np  # noqa: E402

import hipims_mi as hp  # noqa: E402
from hipims_mi import strips, synthetic as syn  # noqa: E402

world, scheme, precision, overlap, rain_on = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
period = int(sys.argv[6]) if len(sys.argv) > 6 else 1
cell_rank = int(sys.argv[7]) if len(sys.argv) > 7 else -2          # >= 0: a cell boundary that only THAT rank is told about
peer_max = int(sys.argv[8]) if len(sys.argv) > 8 else 2
if peer_max == 1:
    os.environ["HP_PEER_DIRECT"] = "0"
variant = sys.argv[9] if len(sys.argv) > 9 else ""                  # "fixed": TIMESTEP_FIXED; "basic": the one-work-item-per-cell kernel; "noq1": quirk Q1 off; "strict": STRICT arithmetic
extra = {"fixed": dict(dynamic_dt=False, dt_fixed=0.004, dt_initial=0.004), "basic": dict(kernel=hp.KERNEL_BASIC),
         "noq1": dict(quirks=hp.QUIRKS_REFERENCE & ~hp.QUIRK_CFL_READS_PRIMARY), "strict": dict(math_mode=hp.MATH_STRICT), "": {}}[variant]
cols, rows, steps = (int(v) for v in os.environ.get("STRIP_WORKER_GRID", "300,157,90").split(","))   # (soak runs: a bigger grid, more iterations)
real = np.float64 if precision == "f64" else np.float32
g = strips.ghost_rows(scheme) * period                             # ghost rows stored per interior side
if rain_on:
    st, bed, man, rain = syn.s_rain_rows(cols, rows, 0, rows, dx=2.0, dtype=real)
    dx = 2.0
else:
    st, bed, man = syn.s_rough(cols, rows, dtype=real)
    rain, dx = None, 1.0


parts = strips.partition(rows, world, g)
cell_ids = None
if cell_rank >= 0:                                                 # five cells in the middle of that rank's owned rows: imposed depth
    lo_, hi_ = parts[cell_rank][0], parts[cell_rank][1]
    cell_ids = np.array([((lo_ + hi_) // 2) * cols + x for x in range(100, 105)], np.uint64)
    cell_series = np.array([[0.0, 0.8, 0.0, 0.0], [1000.0, 0.8, 0.0, 0.0], [2000.0, 0.8, 0.0, 0.0]])


def attach(dom, rank=None):
    if rain is not None:
        dom.add_gridded(hp.GRIDDED_RAIN_INTENSITY, rain["grids"], rain["resolution"], rain["off_x"], rain["off_y"], rain["interval"])
    if cell_ids is not None and (rank is None or rank == cell_rank):     # the other ranks are NOT told (participation)
        dom.add_cell(hp.DEPTH_IS_DEPTH, hp.DISCHARGE_IGNORE, cell_ids, cell_series, 1000.0, 2000.0)


single = hp.Domain(cols, rows, dx=dx, scheme=scheme, precision=precision, **extra)
single.upload(st, bed, man); attach(single); single.set_target_time(1e9)
single.update_timestep()
batches = [int(v) for v in os.environ["STRIP_WORKER_BATCHES"].split(",")] if "STRIP_WORKER_BATCHES" in os.environ else [1, 2, steps - 3]
steps = sum(batches)
wander = [int(v) for v in os.environ["STRIP_WORKER_WANDER"].split(",")] if "STRIP_WORKER_WANDER" in os.environ else []   # iterations between a save and a restore
# a host write into ONE strip between the first two batches (CDomainLink::pushToBuffer's kind of write, hp_domain_upload_rows):
# the level of a few cells of a row in the middle of that rank's owned rows is raised; the other ranks are not told
poke_rank = int(os.environ.get("STRIP_WORKER_POKE", "-1"))
poke_row = (parts[poke_rank][0] + parts[poke_rank][1]) // 2 if poke_rank >= 0 else -1


def poke(dom, local_row):
    row = dom.download(row0=local_row, nrows=1)
    row[0, 10:20, 0] += 0.0625
    row[0, 10:20, 1] = np.maximum(row[0, 10:20, 1], row[0, 10:20, 0])
    dom.upload_rows(row, local_row)


update_between = "STRIP_WORKER_UPDATE" in os.environ          # hp_update_timestep / hp_strip_update_timestep between the first two batches
if poke_rank >= 0 or update_between:
    single.step_batch(batches[0])
    if poke_rank >= 0:
        poke(single, poke_row)
    if update_between:
        single.update_timestep()
    single.step_batch(steps - batches[0])
else:
    single.step_batch(steps)
want, want_sc = single.download(), single.read_scalars()
single.close()

lib = hp.load_library()
hp._check(lib, lib.hp_comm_load(os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so").encode()), "hp_comm_load")
uid = hp.comm_unique_id()
got, scal, errors = [None] * world, [None] * world, []
launches = [None] * world
start = threading.Barrier(world)
tickets, peers_active = [None] * world, [None] * world


def rank_main(r):
    try:
        own_lo, own_hi, lo, hi = parts[r]
        dom = hp.Domain(cols, hi - lo, dx=dx, scheme=scheme, precision=precision, global_rows=rows, row_offset=lo,
                        ghost_rows=g if period > 1 else 0, **extra)
        dom.upload(st[lo:hi], bed[lo:hi], man[lo:hi]); attach(dom, r)
        dom.set_halo_overlap(bool(overlap))
        dom.strip_comm_init(uid, r, world)
        info = dom.strip_info()                           # an explicit choice survives comm_init
        assert info["halo_overlap"] == bool(overlap) and info["comm_rank"] == r and info["ghost_rows"] == g, info
        dom.set_target_time(1e9)
        if peer_max:
            tickets[r] = dom.strip_peer_ticket()
            start.wait()                                  # the tickets travel "by the host's own means"
            peers_active[r] = dom.strip_peer_connect(tickets, r)
            info = dom.strip_info()
            assert peers_active[r] == peer_max and info["peer_max"] and info["peer_halo"] == (peer_max == 2), (peers_active[r], info)
        start.wait()
        dom.strip_update_timestep()                       # tst_Reduce + all-reduce + tst_UpdateTimestep, as after any upload
        for i, n in enumerate(batches):                   # odd and even batch lengths: both ping-pong phases at batch ends
            dom.strip_step_batch(n)
            if i == 0 and r == poke_rank:
                poke(dom, poke_row - lo)
            if i == 0 and update_between:
                dom.strip_update_timestep()
            if i == 0 and wander:                         # a device checkpoint on every rank: save, run on, come back (bench.py's pre-warm does this)
                dom.state_save()
                for w in wander:
                    dom.strip_step_batch(w)
                dom.state_restore()
        dom.sync()
        got[r] = dom.download()[own_lo - lo:own_hi - lo]
        scal[r] = dom.read_scalars()
        launches[r] = dom.launch_counts()[0]
        dom.strip_comm_destroy()
        dom.close()
    except Exception as e:                                # noqa: BLE001
        errors.append((r, repr(e)))
        try:
            start.abort()
        except Exception:                                 # noqa: BLE001
            pass


threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
for t in threads:
    t.start()
for t in threads:
    t.join(300)
if errors or any(t.is_alive() for t in threads):
    print("FAILED", errors, [t.is_alive() for t in threads], flush=True); os._exit(2)
out = np.concatenate(got, axis=0)
same = np.array_equal(out.view(np.uint8), want.view(np.uint8))
if not same:
    bad = np.argwhere((out != want).any(axis=-1))
    worst = np.abs(out.astype(np.float64) - want.astype(np.float64)).max(axis=-1)
    print("largest difference %.3e at (row, col) %s; first differing cells %s" % (worst.max(), np.unravel_index(worst.argmax(), worst.shape), bad[:6].tolist()), flush=True)
    print("differing cells:", len(bad), "rows", sorted(set(bad[:, 0].tolist()))[:20], "cols", sorted(set(bad[:, 1].tolist()))[:12],
          "strip edges", [pp[:2] for pp in parts], flush=True)
times = {(s["time"], s["timestep"]) for s in scal}
print("flux launches per rank", launches, "iterations", steps + sum(wander), flush=True)
print("ranks", world, "scheme", scheme, precision, "overlap", overlap, "rain", rain_on, "period", period, "cell boundary on rank", cell_rank, "variant", variant or "-", "peer-written maximum", peers_active, "bit-identical", same, "times", times,
      "single", (want_sc["time"], want_sc["timestep"]), flush=True)
os._exit(0 if same and times == {(want_sc["time"], want_sc["timestep"])} else 1)
