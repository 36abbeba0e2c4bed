#!/bin/bash
# round 5: K1's main loop with rotating registers (six rows a turn) against the copying two-row loop, on single-iteration lines
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r05fw}; LIBS="${2:-tools/experiments/libs/libhipims_mi_final2.so cur}"
B="python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --no-strict-leg --repeats 2"
L() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-36s %-10s %9.1f Mcell-steps/s  %.4f ms/step  frac %.3f' % ('$1', '$2', d['value'], d['ms_per_step'], d['roofline']['frac']))"; }
run() { name=$1; shift
  for rep in 1 2; do for lib in $LIBS; do
    if [ "$lib" = cur ]; then $B "$@" | L "$name" cur; else HIPIMS_MI_LIB=$PWD/$lib $B "$@" | L "$name" $(basename $lib .so | sed s/libhipims_mi_//); fi
  done; done; }
export HP_TWO_STEP=0
{
run "S-DAM 4096^2 f64 singles"
run "S-ROUGH 4096^2 f64 singles" --workload s-rough
run "S-RAIN 4096^2 f64" --workload s-rain
run "S-RAIN 8192^2 f32 (C5)" --cols 8192 --rows 8192 --steps 100 --workload s-rain --precision f32
run "4096x514 strip f64 singles" --cols 4096 --rows 514 --steps 400
run "8192x1026 strip f32 S-RAIN" --cols 8192 --rows 1026 --steps 200 --workload s-rain --precision f32
} 2>&1 | tee gpurun_out/${TAG}_ab.txt
