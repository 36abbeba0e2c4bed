#!/bin/bash
# A/B on one box, the exact mode's lines: the library in the tree against lib/variants/libhipims_mi_prev.so.  usage: tools/r06_ab_strict.sh <out> [rounds]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; mkdir -p $OUT
Q="--no-cpu-baseline --no-manning-leg --no-moving-leg --no-strict-leg --no-config-legs --steps 200 --math strict"
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
    line "$which STRICT S-DAM 4096^2 f64" --warmup 30
    line "$which STRICT S-DAM 4096^2 f64 developed flood" --warmup 30 --evolve-steps 1500
    HP_TWO_STEP=1 HP_PAIR_TUNE=0 line "$which STRICT S-ROUGH 4096^2 f64, pairs forced" --workload s-rough --warmup 30
  done
done | tee $OUT/summary.txt
