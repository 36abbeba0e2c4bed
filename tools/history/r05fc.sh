#!/bin/bash
# round 5: K1's FILL flag (a single iteration after pairs stores the untouched cells itself instead of a device copy in front of it):
# the pair tests, a soak, and the driver's own command line with and without (HP_FILL_AFTER_PAIRS=0 = the copy)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r05fc}
{
timeout 1500 python -m pytest tests/test_gpu_two_step.py tests/test_gpu_strips.py tests/test_gpu_fallback_paths.py -m gpu -q -x -n 4 2>&1 | tail -3
tools/r05_two_step_soak.sh 7000 300 ${TAG}_soak
L() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%-34s %9.1f Mcell-steps/s  %.4f ms/step  frac %.3f  launches %s' % ('$1', d['value'], d['ms_per_step'], r['frac'], r.get('timed_region_launches')))"; }
B="python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --no-strict-leg"
for rep in 1 2 3; do
  HP_FILL_AFTER_PAIRS=0 $B --gpus 1 --steps 20 --warmup 5 | L "steps 20 warmup 5, copy"
  $B --gpus 1 --steps 20 --warmup 5 | L "steps 20 warmup 5, fill"
done
HP_FILL_AFTER_PAIRS=0 $B --gpus 1 --steps 21 --warmup 4 | L "steps 21 warmup 4, copy"
$B --gpus 1 --steps 21 --warmup 4 | L "steps 21 warmup 4, fill"
$B | L "default"
} 2>&1 | tee gpurun_out/${TAG}.txt
