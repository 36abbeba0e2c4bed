#!/usr/bin/env python3
"""VALU instructions per phase of a FAST fp64 row step, counted in the compiled ISA (static counts; branches that a
wave-uniform fast path skips are listed separately where the source has them).  Runs here (no GPU)."""
import os, re, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-S",
                      "--cuda-device-only", "-w", "-o", "-", os.path.join(here, "budget.hip")], capture_output=True, text=True)
if asm.returncode: sys.exit(asm.stderr)
cur, counts = None, {}
for line in asm.stdout.split("\n"):
    m = re.match(r"^_Z\d+(phase_\w+?)PKdPd:", line)
    if m: cur = m.group(1); counts[cur] = dict(valu=0, trans=0, dpp=0, salu=0, vmem=0, branch=0)
    elif cur:
        ins = re.match(r"\s+([a-z_0-9]+)", line)
        if not ins: continue
        op = ins.group(1)
        if op == "s_endpgm": cur = None; continue
        c = counts[cur]
        if op.startswith("v_"):
            c["valu"] += 1
            if re.match(r"v_(rcp|rsq|sqrt|exp|log)_", op): c["trans"] += 1
        elif op.startswith("s_cbranch"): c["branch"] += 1
        elif op.startswith("s_"): c["salu"] += 1
        elif op.startswith(("global_", "buffer_", "flat_")): c["vmem"] += 1
base = counts.get("phase_empty", dict(valu=0))["valu"]
print(f"{'phase':16s} {'VALU':>6s} {'(trans)':>8s} {'SALU':>6s} {'branches':>9s}   (VALU net of {base} address/IO instructions per kernel not subtracted)")
for k, c in counts.items():
    print(f"{k:16s} {c['valu']:6d} {c['trans']:8d} {c['salu']:6d} {c['branch']:9d}")
