#!/usr/bin/env python3
"""Parse the compiler's kernel-resource-usage remarks (hipims-ocl_amd/lib/resource_usage.txt, written by the library's build) into
one record per kernel: demangled name, VGPRs, SGPRs, spilled VGPRs / SGPRs, scratch bytes per lane, waves per SIMD, LDS bytes.

    python tools/resource_usage.py [resource_usage.txt]            # the table
    python tools/resource_usage.py [resource_usage.txt] --budget   # the table's "kernels" object
    python tools/resource_usage.py --write-budget                   # rewrite csrc/resource_budget.json from this build (after the GPU suite passed on it)

tests/test_resource_usage.py imports parse().  (The reference logs the same figures for every kernel it creates:
src/OpenCL/Executors/COCLKernel.cpp:284-328.)"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = os.path.join(ROOT, "hipims-ocl_amd", "lib", "resource_usage.txt")
FIELDS = {"VGPRs": "vgprs", "TotalSGPRs": "sgprs", "VGPRs Spill": "vgpr_spill", "SGPRs Spill": "sgpr_spill",
          "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "waves", "LDS Size [bytes/block]": "lds"}


def parse(path=DEFAULT):
    recs, cur = [], None
    for line in open(path):
        body = line.strip()
        if body.startswith("Function Name:"):
            cur = {"mangled": body.split(": ", 1)[1]}
            recs.append(cur)
        elif cur is not None and ":" in body:
            k, v = body.split(":", 1)
            if k.strip() in FIELDS:
                v = v.strip()
                cur[FIELDS[k.strip()]] = int(v) if re.fullmatch(r"-?\d+", v) else None      # (symbolic: a kernel that calls through a pointer)
    names = subprocess.run(["c++filt"] + [r["mangled"] for r in recs], capture_output=True, text=True, check=True).stdout.splitlines()
    out = {}
    for r, n in zip(recs, names):
        n = re.sub(r"^void ", "", re.sub(r"\(.*", "", n))
        if not n.startswith("hp::"):
            continue                                                       # (the two store_scalar / copy_scalar helpers of the host file)
        r["name"] = n
        out[n] = r
    return out


def is_strict(name):
    """STRICT instantiations: the first template argument of the flux kernels / the reductions."""
    m = re.match(r"hp::(godunov_march2|godunov_march|muscl_march|inertial_march|godunov_basic|cfl_reduce|cfl_edge_ring)<(true|false)", name)
    return bool(m and m.group(2) == "true")


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    recs = parse(args[0] if args else DEFAULT)
    if "--budget" in sys.argv or "--write-budget" in sys.argv:
        kernels = {n: {"scratch": r["scratch"], "vgpr_spill": r["vgpr_spill"]} for n, r in sorted(recs.items())
                   if (r["scratch"] or 0) > 0 or (r["vgpr_spill"] or 0) > 0}
        if "--write-budget" in sys.argv:          # (only AFTER the bit-exact GPU suite has passed on this build: see the file's own text)
            path = os.path.join(ROOT, "hipims-ocl_amd", "csrc", "resource_budget.json")
            budget = json.load(open(path))
            budget["kernels"] = kernels
            budget["strict_kernels_allowed_to_spill_vector_registers"] = [n for n, r in sorted(recs.items()) if is_strict(n) and (r["vgpr_spill"] or 0) > 0]
            json.dump(budget, open(path, "w"), indent=1)
            print(f"{path}: {len(kernels)} kernels with scratch or spills, {len(budget['strict_kernels_allowed_to_spill_vector_registers'])} STRICT ones among them")
        else:
            print(json.dumps(kernels, indent=1))
    else:
        for n, r in recs.items():
            print(f"{n:72s} VGPR {r['vgprs']!s:>4} SGPR {r['sgprs']!s:>4} scratch {r['scratch']!s:>4} B  spilled V {r['vgpr_spill']!s:>3} S {r['sgpr_spill']!s:>3}"
                  f"  waves {r['waves']}  LDS {r['lds']}")
