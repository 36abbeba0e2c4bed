#!/bin/bash
# fp32 S-RAIN: pairs with the rain (HP_TWO_STEP=1 forces) against single iterations with the rain fused (HP_PAIR_BDY=0), by size -- where
# the default's threshold for fp32 pairs with area boundaries belongs (hp_engine.hip: pairs_possible_common).  usage: <out>
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; mkdir -p $OUT
Q="--no-cpu-baseline --no-manning-leg --no-moving-leg --no-strict-leg --no-config-legs --steps 200 --warmup 50 --workload s-rain --precision f32"
line() { name=$1; shift; l=$(timeout 900 python3 bench.py $Q "$@" 2>/dev/null | grep '^{' | tail -1); python3 - "$name" "$l" <<'PY'
import json, sys
n, l = sys.argv[1:3]
b = json.loads(l); r = b["roofline"]
print(f"{n:54s} {b['ms_per_step']:.4f} ms/step  frac {r['frac']:.3f}  it/launch {r['iterations_per_launch']}")
PY
}
for i in 1 2; do
for shape in "2048 2048" "3072 3072" "4096 4096" "4096 6144" "6144 6144" "8192 8192"; do
  set -- $shape
  HP_TWO_STEP=1 line "f32 S-RAIN $1 x $2 pairs" --cols $1 --rows $2
  HP_PAIR_BDY=0 line "f32 S-RAIN $1 x $2 single iterations" --cols $1 --rows $2
done; done | tee $OUT/summary.txt
