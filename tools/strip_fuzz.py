#!/usr/bin/env python3
"""Random strip configurations through tests/strip_threads_worker.py (2-4 thread ranks on the one GPU, the library's own strip
loop), each compared bit for bit with the single domain: world, scheme, precision, iterations per exchange, rain, halo overlap,
transport level (0 library, 1 mailboxes, 2 mailboxes + pushed ghost rows), variant (fixed dt, cross-check kernel, Q1 off), a cell
boundary that only one rank knows, and grid shape are drawn from a seeded generator.   usage: strip_fuzz.py <first seed> <count>"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    world = int(rng.integers(2, 5))
    scheme = int(rng.choice([0, 0, 1, 2]))
    precision = str(rng.choice(["f64", "f64", "f32"]))
    period = int(rng.integers(1, 3))
    rain = int(rng.integers(0, 2)) if scheme == 0 else 0
    overlap = int(rng.integers(0, 2))
    level = int(rng.choice([2, 2, 1, 0]))
    variant = str(rng.choice(["", "", "", "fixed", "basic", "noq1"])) if scheme == 0 else ""
    g = (2 if scheme == 1 else 1) * period
    cols = int(rng.integers(70, 900))
    rows = int(rng.integers(world * (3 * g + 4), world * (3 * g + 4) + 300))
    steps = int(rng.integers(12, 120))
    cell_rank = int(rng.integers(0, world)) if (scheme == 0 and cols > 110 and rng.random() < 0.3) else -2    # a cell boundary only ONE rank is told about
    if variant == "" and rng.random() < 0.25:              # (drawn last: the seeds that found something keep their configurations)
        variant = "strict"
    env = dict(os.environ, STRIP_WORKER_GRID=f"{cols},{rows},{steps}")
    extra = ""
    if rng.random() < 0.35:                                # a checkpoint after a random first batch, a random walk away from it, and back
        first_batch = int(rng.integers(1, steps - 2))
        env["STRIP_WORKER_BATCHES"] = f"{first_batch},{steps - first_batch}"
        env["STRIP_WORKER_WANDER"] = ",".join(str(int(v)) for v in rng.integers(1, 12, int(rng.integers(1, 3))))
        extra = f"checkpoint after {first_batch}, away for {env['STRIP_WORKER_WANDER']}"
    elif rng.random() < 0.3 and cols > 40:                 # a host write into one strip's rows between two batches
        first_batch = int(rng.integers(1, steps - 2))
        env["STRIP_WORKER_BATCHES"] = f"{first_batch},{steps - first_batch}"
        env["STRIP_WORKER_POKE"] = str(int(rng.integers(0, world)))
        extra = f"rows of rank {env['STRIP_WORKER_POKE']} written by the host after {first_batch}"
        if rng.random() < 0.5:
            env["STRIP_WORKER_UPDATE"] = "1"
            extra += ", then an update of the timestep"
    elif rng.random() < 0.25:
        first_batch = int(rng.integers(1, steps - 2))
        env["STRIP_WORKER_BATCHES"] = f"{first_batch},{steps - first_batch}"
        env["STRIP_WORKER_UPDATE"] = "1"
        extra = f"an update of the timestep after {first_batch}"
    cmd = [sys.executable, os.path.join(ROOT, "tests", "strip_threads_worker.py"), str(world), str(scheme), precision, str(overlap), str(rain),
           str(period), str(cell_rank), str(level), variant]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    ok = r.returncode == 0 and "bit-identical True" in r.stdout
    bad += not ok
    print("seed", seed, "ok" if ok else "FAILED", "world", world, "scheme", scheme, precision, "period", period, "rain", rain, "overlap", overlap,
          "level", level, "variant", variant or "-", "cell boundary on rank", cell_rank, "grid", (cols, rows, steps), extra, "" if ok else (r.stdout + r.stderr)[-600:], flush=True)
print("failed:", bad, "of", count)
sys.exit(1 if bad else 0)
