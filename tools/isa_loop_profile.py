#!/usr/bin/env python3
"""Instruction mix of a kernel's hot loop in compiler assembly (-S): what the register allocator's spills cost where it matters.

    python tools/isa_loop_profile.py engine.s godunov_march2ILb0ELi1ELb0ELi1EdEE

The compiler annotates every basic block with the loop it belongs to ("Loop Header: BBn_m"); the blocks of the loop with the most
vector instructions are summed: VALU, SALU, scratch loads / stores (spilled vector registers), v_readlane / v_writelane (spilled
scalar registers), buffer / global / LDS instructions, branches.  Totals per kernel (kernel-resource-usage) do not say where the
spill code sits; this does."""
import collections
import re
import sys


def profile(path, key):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and ":" in l)
    per_loop = collections.defaultdict(collections.Counter)
    cur = None
    for l in lines[start + 1:]:
        if l.startswith("\t.end_amdhsa_kernel") or l.startswith(".Lfunc_end"):
            break
        if re.match(r"^(\.LBB\d+_\d+):", l) or l.startswith("; %bb."):
            h = re.search(r"Header[=:] ?(BB\d+_\d+)", l)
            cur = h.group(1) if h else None
            continue
        m = re.match(r"\s+([a-z_0-9]+)", l)
        if not m or cur is None:
            continue
        op = m.group(1)
        c = per_loop[cur]
        if op.startswith(("v_readlane", "v_writelane")):
            c["sgpr_spill_moves"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
        elif op.startswith("scratch_load"):
            c["scratch_load"] += 1
        elif op.startswith("scratch_store"):
            c["scratch_store"] += 1
        elif op.startswith(("s_cbranch", "s_branch")):
            c["branch"] += 1
        elif op.startswith("s_waitcnt"):
            c["waitcnt"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
        elif op.startswith(("buffer_", "global_", "flat_")):
            c["vmem"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
    return per_loop


if __name__ == "__main__":
    loops = profile(sys.argv[1], sys.argv[2])
    for name, c in sorted(loops.items(), key=lambda kv: -kv[1]["valu"])[:int(sys.argv[3]) if len(sys.argv) > 3 else 2]:
        print(name, dict(c))
