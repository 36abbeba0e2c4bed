#!/bin/bash
# A/B on one box, workload lines of DESIGN 6 that the default line does not carry: the library in the tree against
# lib/variants/libhipims_mi_prev.so.  usage: tools/r06_ab_lines.sh <out> [rounds]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; mkdir -p $OUT
Q="--no-cpu-baseline --no-manning-leg --no-moving-leg --no-strict-leg --no-config-legs --steps 200 --warmup 20"
line() { name=$1; shift; l=$(timeout 900 python3 bench.py $Q "$@" 2>/dev/null | grep '^{' | tail -1); python3 - "$name" "$l" <<'PY'
import json, sys
n, l = sys.argv[1:3]
b = json.loads(l); r = b["roofline"]
print(f"{n:64s} {b['ms_per_step']:.4f} ms/step  frac {r['frac']:.3f}  it/launch {r['iterations_per_launch']}")
PY
}
PREV=$PWD/hipims-ocl_amd/lib/variants/libhipims_mi_prev.so
for i in $(seq 1 ${2:-2}); do
  for which in new prev; do
    if [ $which = prev ]; then export HIPIMS_MI_LIB=$PREV; else unset HIPIMS_MI_LIB; fi
    line "$which S-DAM 4096^2 f64"
    line "$which S-ROUGH 4096^2 f64" --workload s-rough
    line "$which S-RAIN 4096^2 f64 (pairs with the rain)" --workload s-rain --warmup 50
    HP_PAIR_EXACT=1 line "$which S-DAM 4096^2 f64, every pair exact"
    HP_PAIR_EXACT=1 line "$which S-ROUGH 4096^2 f64, every pair exact" --workload s-rough
    line "$which S-DAM 4096x514 strip" --cols 4096 --rows 514
  done
done | tee $OUT/summary.txt
