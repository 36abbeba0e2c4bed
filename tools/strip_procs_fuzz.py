#!/usr/bin/env python3
"""Random strip configurations with ranks that are PROCESSES (tests/strip_procs_worker.py: ghost rows and maxima through IPC
mappings of the other processes' memory, the collective double in its shared-file mode for the handshake only), each compared
bit for bit with the single domain.   usage: strip_procs_fuzz.py <first seed> <count>"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hipims-ocl_amd")]
os.environ["HIPIMS_MI_NO_TORCH"] = "1"
import numpy as np
import hipims_mi as hp
from hipims_mi import strips, synthetic as syn

first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    world = int(rng.integers(2, 5))
    scheme = int(rng.choice([0, 0, 1, 2]))
    precision = str(rng.choice(["f64", "f64", "f32"]))
    period = int(rng.integers(1, 3))
    rain = int(rng.integers(0, 2)) if scheme == 0 else 0
    g = (2 if scheme == 1 else 1) * period
    cols = int(rng.integers(70, 1200))
    rows = int(rng.integers(world * (3 * g + 4), world * (3 * g + 4) + 300))
    steps = int(rng.integers(12, 120))
    real = np.float64 if precision == "f64" else np.float32
    if rain:
        st, bed, man, rn = syn.s_rain_rows(cols, rows, 0, rows, dx=2.0, dtype=real)
        dx = 2.0
    else:
        st, bed, man = syn.s_rough(cols, rows, dtype=real)
        rn, dx = None, 1.0
    with tempfile.TemporaryDirectory() as tmp:
        shm = os.path.join(tmp, "allreduce.shm")
        open(shm, "wb").write(bytes(4096))
        env = dict(os.environ, FAKE_RCCL_SHM=shm, HP_PEER_TEST_MS="20000", STRIP_WORKER_GRID=f"{cols},{rows},{steps}")
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "strip_procs_worker.py"), str(r), str(world), tmp, str(scheme), precision, str(rain), str(period)],
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(world)]
        single = hp.Domain(cols, rows, dx=dx, scheme=scheme, precision=precision)
        single.upload(st, bed, man)
        if rn is not None:
            single.add_gridded(hp.GRIDDED_RAIN_INTENSITY, rn["grids"], rn["resolution"], rn["off_x"], rn["off_y"], rn["interval"])
        single.set_target_time(1e9); single.update_timestep(); single.step_batch(steps)
        want, sc = single.download(), single.read_scalars()
        single.close()
        outs = [p.communicate(timeout=300)[0] for p in procs]
        ok = all(p.returncode == 0 for p in procs)
        if ok:
            got = np.concatenate([np.load(os.path.join(tmp, f"owned.{r}.npy")) for r in range(world)], axis=0)
            stamp = "time %.17g dt %.17g" % (sc["time"], sc["timestep"])
            ok = np.array_equal(got.view(np.uint8), want.view(np.uint8)) and all(stamp in o for o in outs)
    bad += not ok
    print("seed", seed, "ok" if ok else "FAILED", "world", world, "scheme", scheme, precision, "period", period, "rain", rain, "grid", (cols, rows, steps), "" if ok else "\n".join(outs)[-500:], flush=True)
print("failed:", bad, "of", count)
sys.exit(1 if bad else 0)
