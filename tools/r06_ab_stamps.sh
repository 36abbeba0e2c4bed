#!/bin/bash
# Round 6 A/B on one box: the default (pairs with area boundaries; quirk Q3's stamps only where a boundary removes water) | every pair
# exact (HP_PAIR_EXACT=1: godunov_march2's HZ instantiation -- what the bookkeeping costs the march) | area-boundary domains kept on
# single iterations (HP_PAIR_BDY=0: what pairs buy S-RAIN).  usage: tools/r06_ab_stamps.sh <outdir under gpurun_out> [rounds]
# (profiles/r06c..f_stamps_ab.txt are earlier forms of the stamps, built as a second library: a branch per row, masked stores)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; mkdir -p $OUT
ARGS="--no-cpu-baseline --no-manning-leg --no-strict-leg"
for i in $(seq 1 ${2:-2}); do
  python3 bench.py $ARGS > $OUT/shipped_$i.json 2> $OUT/shipped_$i.err
  HP_PAIR_EXACT=1 python3 bench.py $ARGS > $OUT/exact_$i.json 2> $OUT/exact_$i.err
  HP_PAIR_BDY=0 python3 bench.py $ARGS > $OUT/nobdypairs_$i.json 2> $OUT/nobdypairs_$i.err
done
python3 - <<'PY'
import json,glob,os,sys
out=sys.argv[1] if len(sys.argv)>1 else None
PY
for f in $OUT/*.json; do python3 -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
c5=d.get('c5_fp32_rain',{}); c3=d.get('c3_muscl',{}); mv=d.get('moving_water',{})
print('%-22s S-DAM %.4f ms frac %.3f | S-ROUGH %.4f ms %.3f | C5 %.4f ms %.3f ipl %s | C3 %.4f ms %.3f' % (os.path.basename('$f') if False else '$f'.split('/')[-1], d['ms_per_step'], d['roofline']['frac'], mv.get('ms_per_step',0), mv.get('frac',0), c5.get('ms_per_step',0), c5.get('frac',0), c5.get('iterations_per_launch'), c3.get('ms_per_step',0), c3.get('frac',0)))
"; done | tee $OUT/summary.txt
