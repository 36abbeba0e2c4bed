#!/bin/bash
# round 5, after the arithmetic cuts: (bands, tile rows) of the pair kernel on the strong-scaling strip once more
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --no-strict-leg --repeats 3 --cols 4096 --rows 514 --steps 400 --warmup 40"
L() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s %.2f us per iteration  frac %.3f  it/launch %d' % ('$1', d['ms_per_step']*1e3, d['roofline']['frac'], d['roofline']['iterations_per_launch']))"; }
{
HP_PRINT_TILING=1 $B 2>&1 >/dev/null | grep "pair kernel" | head -1
$B | L "pairs (default tiling)"
HP_TWO_STEP=0 $B | L "single iterations (default tiling)"
for nb in 12 13 14 15 16 18 21; do for r in 12 13 14 15 16 17 18 19; do
  HP_TWO_STEP=1 HP_NBANDS=$nb HP_MARCH2_RSEG=$r $B | L "pairs bands=$nb rseg=$r"
done; done
$B | L "pairs (default tiling)"
} 2>&1 | tee gpurun_out/r05fo_strip_sweep.txt
