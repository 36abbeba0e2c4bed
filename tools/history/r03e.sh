#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; mkdir -p $OUT
for w in "s-rough muscl" "s-rain godunov" "s-rough godunov"; do set -- $w
  ARGS="--workload $1 --scheme $2 --no-cpu-baseline --no-manning-leg --no-moving-leg --steps 20 --warmup 5 --prewarm-s 0.1 --repeats 1"
  D=$OUT/$1_$2; mkdir -p $D
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $D/a -- python3 bench.py $ARGS > $D/a.log 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS_F64 GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d $D/b -- python3 bench.py $ARGS > $D/b.log 2>&1
  rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INST_LEVEL_VMEM --output-format csv -d $D/c -- python3 bench.py $ARGS > $D/c.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/*_*")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "march" in k:
            print(d.split("/")[-1], k)
            for c, x in sorted(v.items()): print(f"    {c:28s} {sum(x)/len(x)/1e6:12.3f} M")
PY
