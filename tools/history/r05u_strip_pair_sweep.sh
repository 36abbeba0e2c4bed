#!/bin/bash
# round 5: can iteration pairs pay on the strong-scaling strip (4096 x 514, one round of blocks)?  (bands, tile rows) sweep of godunov_march2
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --no-strict-leg --repeats 3 --cols 4096 --rows 514 --steps 400 --warmup 40"
L() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s %.2f us per iteration  frac %.3f  it/launch %d' % ('$1', d['ms_per_step']*1e3, d['roofline']['frac'], d['roofline']['iterations_per_launch']))"; }
{
HP_TWO_STEP=0 $B | L "single iterations (default tiling)"
for nb in 8 10 12 14 15 16; do for r in 10 11 12 13 14 15 16 17 18 20 22; do
  HP_TWO_STEP=1 HP_NBANDS=$nb HP_MARCH2_RSEG=$r $B | L "pairs bands=$nb rseg=$r"
done; done
HP_TWO_STEP=0 $B | L "single iterations (default tiling)"
} 2>&1 | tee gpurun_out/r05u_strip_pair_sweep.txt
