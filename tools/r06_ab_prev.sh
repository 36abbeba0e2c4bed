#!/bin/bash
# A/B on one box: the library in the tree against lib/variants/libhipims_mi_prev.so, the default line's legs.  usage: tools/r06_ab_prev.sh <out> [rounds]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; mkdir -p $OUT
ARGS="--no-cpu-baseline --no-manning-leg --no-strict-leg"
for i in $(seq 1 ${2:-2}); do
  python3 bench.py $ARGS > $OUT/new_$i.json 2> $OUT/new_$i.err
  HIPIMS_MI_LIB=$PWD/hipims-ocl_amd/lib/variants/libhipims_mi_prev.so python3 bench.py $ARGS > $OUT/prev_$i.json 2> $OUT/prev_$i.err
done
for f in $OUT/*.json; do python3 -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1])
c5=d.get('c5_fp32_rain',{}); c3=d.get('c3_muscl',{}); mv=d.get('moving_water',{})
print('%-14s S-DAM %.4f ms frac %.3f | S-ROUGH %.4f ms %.3f | C5 %.4f ms %.3f | C3 %.4f ms %.3f dev %.4f' % ('$f'.split('/')[-1], d['ms_per_step'], d['roofline']['frac'], mv.get('ms_per_step',0), mv.get('frac',0), c5.get('ms_per_step',0), c5.get('frac',0), c3.get('ms_per_step',0), c3.get('frac',0), c3.get('developed_flood',{}).get('ms_per_step',0)))
"; done | tee $OUT/summary.txt
