#!/bin/bash
# round 5: three builds (m1: four cuts; m2: + merged clamps + lazy dry tests; cur: lazy dry tests in fp64 only) on the lines that disagreed
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r05fm}; LIBS="${2:-tools/experiments/libs/libhipims_mi_m1.so tools/experiments/libs/libhipims_mi_m2.so cur}"
B="python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --no-strict-leg --repeats 2"
L() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-34s %-22s %9.1f Mcell-steps/s  %.4f ms/step  frac %.3f' % ('$1', '$2', d['value'], d['ms_per_step'], d['roofline']['frac']))"; }
run() { name=$1; shift
  for rep in 1 2; do for lib in $LIBS; do
    if [ "$lib" = cur ]; then $B "$@" | L "$name" cur; else HIPIMS_MI_LIB=$PWD/$lib $B "$@" | L "$name" $(basename $lib .so | sed s/libhipims_mi_//); fi
  done; done; }
{
run "S-DAM 4096^2 godunov f32" --precision f32
run "S-RAIN 4096^2 godunov f32" --workload s-rain --precision f32
run "S-RAIN 8192^2 godunov f32 (C5)" --cols 8192 --rows 8192 --steps 100 --workload s-rain --precision f32
run "S-RAIN 4096^2 godunov f64" --workload s-rain
run "S-DAM 4096^2 godunov f64"
run "S-DAM developed muscl f64" --scheme muscl --evolve-steps 1500
run "S-ROUGH 4096^2 muscl f64" --workload s-rough --scheme muscl
} 2>&1 | tee gpurun_out/${TAG}_ab.txt
