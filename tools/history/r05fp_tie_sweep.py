"""round 5: the pair kernel's one-round tiling search ties between tile heights (same rows per CU); which of the tied heights is
fastest, over several one-round shapes?   usage: python tools/r05fp_tie_sweep.py"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-manning-leg", "--no-moving-leg", "--no-strict-leg",
     "--repeats", "3", "--steps", "400", "--warmup", "40"]


def run(cols, rows, **env):
    r = subprocess.run(B + ["--cols", str(cols), "--rows", str(rows)], capture_output=True, text=True, env=dict(os.environ, **env))
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    m = re.search(r"pair kernel (\d+) rows x (\d+) bands, (\w+)", r.stderr)
    return d["ms_per_step"] * 1e3, d["roofline"]["iterations_per_launch"], (int(m.group(1)), int(m.group(2)), m.group(3)) if m else None


for cols, rows in ((4096, 514), (8192, 258), (2048, 1026), (3072, 514), (4096, 386), (4096, 642), (6144, 386), (1448, 1448)):
    us, ipl, til = run(cols, rows, HP_PRINT_TILING="1", HP_TWO_STEP="1")
    r0, nb, _ = til
    line = [f"{cols}x{rows}: search {nb} bands x {r0} rows {us:.2f} us |"]
    for r in range(max(4, r0 - 3), r0 + 2):
        us_r, _, _ = run(cols, rows, HP_TWO_STEP="1", HP_NBANDS=str(nb), HP_MARCH2_RSEG=str(r))
        line.append(f"r={r}: {us_r:.2f}")
    us1, _, _ = run(cols, rows, HP_TWO_STEP="0")
    line.append(f"| singles {us1:.2f}")
    print("  ".join(line), flush=True)
