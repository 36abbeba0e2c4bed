#!/usr/bin/env python3
"""Round 4: what a one-round launch (the 4096 x 514 strip of the 8-GPU strong-scaling split) gains from block counts that the
8-band TileMap cannot offer.  Runs itself once per (HP_NBANDS, HP_MARCH_RSEG) in a child process (both are read once).
usage: r04_band_sweep.py [cols rows]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.join(ROOT, "hipims-ocl_amd"))
    os.environ["HIPIMS_MI_NO_TORCH"] = "1"
    import hipims_mi as hp
    from hipims_mi import synthetic as syn
    cols, rows, scheme, prec = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    import numpy as np
    st, bed, man = syn.s_dam(cols, rows, dtype=np.float64 if prec == "f64" else np.float32)
    d = hp.Domain(cols, rows, scheme=scheme, precision=prec)
    d.upload(st, bed, man); d.set_target_time(1e9); d.step_batch(100); d.sync()
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter(); d.step_batch(1500); d.sync(); best = min(best, (time.perf_counter() - t0) / 1500 * 1e6)
    print("%.2f" % best)
    sys.exit(0)
cols, rows = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 514)
scheme = int(sys.argv[3]) if len(sys.argv) > 3 else 0
prec = sys.argv[4] if len(sys.argv) > 4 else "f64"
g = 2 if scheme == 1 else 1
groups = (((cols - 2 * g) + (60 if scheme == 1 else 62) - 1) // (60 if scheme == 1 else 62) + 3) // 4
knob = {0: "HP_MARCH_RSEG", 1: "HP_MUSCL_RSEG", 2: "HP_INERTIAL_RSEG"}[scheme]
print(f"{cols} x {rows} scheme {scheme} {prec}: {groups} groups of 4 wavefronts; us per iteration (blocks)")
def run(nb, rseg):
    env = dict(os.environ)
    if nb: env["HP_NBANDS"] = str(nb)
    if rseg: env[knob] = str(rseg)
    out = subprocess.run([sys.executable, __file__, "child", str(cols), str(rows), str(scheme), prec], capture_output=True, text=True, env=env)
    return out.stdout.strip() or out.stderr[-200:]
def run_default(search):
    env = dict(os.environ, HP_PRINT_TILING="1", HP_TILING_SEARCH=str(search))
    out = subprocess.run([sys.executable, __file__, "child", str(cols), str(rows), str(scheme), prec], capture_output=True, text=True, env=env)
    return out.stdout.strip() + "   " + " ".join(l for l in out.stderr.splitlines() if "tiling" in l)
print("default tiling (search on) :", run_default(1))
print("default tiling (search off):", run_default(0))
if os.environ.get("SWEEP", "1") == "0":
    sys.exit(0)
upd = rows - 2 * g
NBS = [int(v) for v in os.environ.get("SWEEP_NB", "8,5,6,10,12,15,30").split(",")]
RSEGS = [int(v) for v in os.environ.get("SWEEP_RSEG", "9,11,13,15,16,17,18,19,22,26").split(",")]
for nb in NBS:
    line = []
    for rseg in RSEGS:
        band = (upd + nb - 1) // nb
        if rseg > band: continue
        blocks = nb * groups * ((band + rseg - 1) // rseg)
        line.append(f"rseg {rseg}: {run(nb, rseg)} ({blocks})")
    print(f"nbands {nb:2d} | " + " | ".join(line), flush=True)
