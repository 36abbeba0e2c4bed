#!/bin/bash
# round 5: tile height of K1 on config C5's strip (8192 x 1026 fp32, S-RAIN) and on the fp64 S-RAIN strip, after the round's kernel changes
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r05fd}
B="python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --no-strict-leg --repeats 2"
L() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-38s %-22s %9.1f Mcell-steps/s  %.4f ms/step  frac %.3f' % ('$1', '$2', d['value'], d['ms_per_step'], d['roofline']['frac']))"; }
{
C5="--cols 8192 --rows 1026 --steps 200 --workload s-rain --precision f32"
HP_PRINT_TILING=1 $B $C5 2>&1 >/dev/null | grep tiling | head -2
$B $C5 | L "C5 strip S-RAIN f32" default
HP_TILING_SEARCH_F32=1 $B $C5 | L "C5 strip S-RAIN f32" "searched"
for r in 8 10 12 14 16 20 24 32; do HP_MARCH_RSEG=$r $B $C5 | L "C5 strip S-RAIN f32" "rseg $r"; done
for nb in 4 6 10 12 16; do HP_NBANDS=$nb $B $C5 | L "C5 strip S-RAIN f32" "nbands $nb"; done
$B $C5 | L "C5 strip S-RAIN f32" default
C5D="--cols 8192 --rows 1026 --steps 200 --precision f32"
$B $C5D | L "C5 strip S-DAM f32" default
for r in 10 12 14 16 20; do HP_MARCH_RSEG=$r HP_TWO_STEP=0 $B $C5D | L "C5 strip S-DAM f32 singles" "rseg $r"; done
} 2>&1 | tee gpurun_out/${TAG}.txt
