"""round 5: the pair kernel's tilings after the new tie rule (eight one-round shapes), and K1's single-iteration tiling on the strong-scaling strip
once more.   usage: python tools/r05fq.py"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-manning-leg", "--no-moving-leg", "--no-strict-leg",
     "--repeats", "3", "--steps", "400", "--warmup", "40"]


def run(cols, rows, **env):
    r = subprocess.run(B + ["--cols", str(cols), "--rows", str(rows)], capture_output=True, text=True, env=dict(os.environ, HP_PRINT_TILING="1", **env))
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    m = re.search(r"pair kernel (\d+) rows x (\d+) bands, (\w+)", r.stderr)
    k = re.search(r"K1/K6 (\d+) rows x (\d+) bands", r.stderr)
    return d["ms_per_step"] * 1e3, d["roofline"]["iterations_per_launch"], (m.group(1), m.group(2), m.group(3)) if m else None, (k.group(1), k.group(2)) if k else None


for cols, rows in ((4096, 514), (8192, 258), (2048, 1026), (3072, 514), (4096, 386), (4096, 642), (6144, 386), (1448, 1448)):
    us, ipl, til, _ = run(cols, rows)
    us1, _, _, k1 = run(cols, rows, HP_TWO_STEP="0")
    print(f"{cols}x{rows}: default {us:.2f} us (it/launch {ipl}; pair kernel {til[1]} bands x {til[0]} rows, {til[2]})   singles {us1:.2f} us (K1 {k1[1]} bands x {k1[0]} rows)", flush=True)
print("K1 single iterations on 4096x514:")
for nb in (14, 15, 16):
    line = [f"  bands {nb}:"]
    for r in (12, 13, 14, 15, 16, 17):
        us, _, _, _ = run(4096, 514, HP_TWO_STEP="0", HP_NBANDS=str(nb), HP_MARCH_RSEG=str(r))
        line.append(f"r={r}: {us:.2f}")
    print("  ".join(line), flush=True)
