#!/bin/bash
# usage: tools/fetch_probe.sh <tag> [env assignments...] -- measures FETCH_SIZE / WRITE_SIZE / duration of the flux kernel
cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1; shift
for a in "$@"; do export "$a"; done
OUT=gpurun_out/probe_$TAG; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/w.log 2>&1
python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/b.log 2>&1
python3 - <<PY
import csv,glob,json
def avg(d,c):
    f=glob.glob("$OUT/"+d+"/**/*_counter_collection.csv",recursive=True)[0]
    v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "march" in r["Kernel_Name"] and r["Counter_Name"]==c]
    return sum(v)/len(v)
b=json.loads(open("$OUT/b.log").read().strip().splitlines()[-1])
fe,wr=avg("f","FETCH_SIZE")*1024*2,avg("w","WRITE_SIZE")*1024
ms=b["roofline"]["avg_launch_ms"]
print("$TAG", "read_MB",round(fe/1e6),"write_MB",round(wr/1e6),"launch_ms",round(ms,4),"value",round(b["value"]),"HBM_TBps",round((fe+wr)/ms/1e9,2))
PY
