#!/usr/bin/env python3
"""Basic-block listing of one kernel in an ISA dump: per block VALU / LDS / VMEM / SALU counts and where it branches.
usage: blocks.py <file.s> <substring of the kernel symbol>"""
import re, sys
src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and ":" in l)
blocks, cur = [], None
for i in range(start + 1, len(lines)):
    l = lines[i]
    if l.startswith("\t.end_amdhsa_kernel") or l.startswith(".Lfunc_end"): break
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m or cur is None:
        cur = dict(name=m.group(1) if m else "entry", valu=0, dpp=0, lds=0, vmem=0, salu=0, trans=0, term=[], line=i + 1)
        blocks.append(cur)
        if m: continue
    ins = re.match(r"\s+([a-z_0-9]+)\s*(.*)", l)
    if not ins: continue
    op, rest = ins.group(1), ins.group(2)
    if op.startswith("v_"):
        cur["valu"] += 1
        if "dpp" in op or "row_" in rest or "wave_" in rest: cur["dpp"] += 1
        if re.match(r"v_(rcp|rsq|sqrt|exp|log)_", op): cur["trans"] += 1
    elif op.startswith("ds_"): cur["lds"] += 1
    elif op.startswith(("buffer_", "global_", "flat_", "scratch_")): cur["vmem"] += 1
    elif op.startswith("s_cbranch") or op == "s_branch":
        cur["term"].append(op.replace("s_cbranch_", "") + "->" + rest.split()[0])
        cur = dict(name="  +" + cur["name"].strip(" +"), valu=0, dpp=0, lds=0, vmem=0, salu=0, trans=0, term=[], line=i + 2)
        blocks.append(cur)
    elif op.startswith("s_"): cur["salu"] += 1
tot = sum(b["valu"] for b in blocks)
print(f"{len(blocks)} blocks, {tot} VALU static")
for b in blocks:
    print(f"{b['name']:12s} L{b['line']:<7d} valu {b['valu']:4d} dpp {b['dpp']:3d} trans {b['trans']:2d} lds {b['lds']:3d} vmem {b['vmem']:2d} salu {b['salu']:3d}  {' '.join(b['term'])}")
