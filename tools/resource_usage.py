#!/usr/bin/env python3
"""Compact per-kernel resource table (VGPRs, SGPRs, scratch, LDS, occupancy) from hipcc's
-Rpass-analysis=kernel-resource-usage; runs here (cross-compile, no GPU).  usage: tools/resource_usage.py [filter]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "hipims-ocl_amd", "csrc", "hp_engine.hip")
flags = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math".split()
out = subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"],
                     capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s+\[-Rpass", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
flt = sys.argv[1] if len(sys.argv) > 1 else ""
print(f"{'kernel':70s} VGPR AGPR SGPR scratch   LDS occ  sgpr-spill vgpr-spill")
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("void hp::", "")
    if flt in n:
        print(f"{n:70s} {r.get('VGPRs','?'):>4} {r.get('AGPRs','?'):>4} {r.get('TotalSGPRs','?'):>4} {r.get('ScratchSize [bytes/lane]','?'):>7} {r.get('LDS Size [bytes/block]','?'):>5} {r.get('Occupancy [waves/SIMD]','?'):>3} {r.get('SGPRs Spill','?'):>10} {r.get('VGPRs Spill','?'):>10}")
