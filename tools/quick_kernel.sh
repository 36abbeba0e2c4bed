#!/bin/bash
# GPU box: quick look at one kernel flavour -- bench line + VALU / LDS instruction counts per launch (one PMC pass)
# usage: tools/quick_kernel.sh <tag> [bench args...]      (HIPIMS_MI_LIB selects a library variant)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1; shift
OUT=gpurun_out/quick_$TAG; rm -rf $OUT; mkdir -p $OUT
python3 bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --steps 200 --warmup 20 $* > $OUT/bench.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d $OUT/pmc -- python3 bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --steps 20 --warmup 5 --prewarm-s 0.1 --repeats 1 $* > $OUT/pmc.log 2>&1
python3 - $OUT "$TAG $*" <<'PY'
import csv, glob, json, sys, collections
out, tag = sys.argv[1:3]
b = json.loads([l for l in open(out + "/bench.log") if l.startswith("{")][-1])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"{tag:40s} step {b['ms_per_step']:.4f} ms  kernel {b['roofline']['avg_launch_ms']:.4f} ms  frac {b['roofline']['frac']:.3f}")
for k, v in agg.items():
    if "march" in k or "bdy" in k:
        print("    %-50s %s" % (k[:50], "  ".join(f"{c}={sum(x)/len(x)/1e6:.2f}M" for c, x in sorted(v.items()))))
PY
