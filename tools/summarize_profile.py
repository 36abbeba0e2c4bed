#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/prof/*) into the tracked summaries under profiles/.

    python tools/summarize_profile.py gpurun_out/prof profiles/r01_godunov4096 "<note>"

Reads  <prof>/kt/**/_kernel_stats.csv          (rocprofv3 --kernel-trace --stats)
       <prof>/pmc_*/**/_counter_collection.csv  (one --pmc pass each; FETCH_SIZE and WRITE_SIZE in separate passes)
Writes <out>_kernel_stats.csv (verbatim copy), <out>_pmc.json (per-kernel per-launch averages + derived HBM bytes).
FETCH_SIZE is doubled before use (gfx950 tallies 128-B requests as 64 B for wide coalesced reads;
/opt/skills/guides/MI355X_MICROARCH.md, HBM section); WRITE_SIZE is used as read.  Both are reported in KiB.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def main():
    prof, out = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    ks = glob.glob(os.path.join(prof, "kt", "**", "*_kernel_stats.csv"), recursive=True)
    if ks:
        shutil.copy(ks[0], out + "_kernel_stats.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(prof, "pmc_*", "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    summary = {"note": note, "kernels": {}}
    for k, v in agg.items():
        e = {c: sum(x) / len(x) for c, x in v.items()}
        e["launches_sampled"] = max(len(x) for x in v.values())
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_read_bytes_corrected"] = e["FETCH_SIZE"] * 1024 * 2
            e["hbm_write_bytes"] = e["WRITE_SIZE"] * 1024
            e["hbm_bytes_per_launch"] = e["hbm_read_bytes_corrected"] + e["hbm_write_bytes"]
        summary["kernels"][k] = e
    if ks:
        stats = {}
        for r in csv.DictReader(open(ks[0])):
            stats[r["Name"].split("(")[0]] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                              "pct": float(r["Percentage"])}
        summary["kernel_trace"] = stats
    json.dump(summary, open(out + "_pmc.json", "w"), indent=1)
    print(json.dumps(summary, indent=1)[:3000])


if __name__ == "__main__":
    main()
