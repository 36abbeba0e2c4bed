#!/bin/bash
# round 5: fp32 faces in packed arithmetic (face_solve_fast_xy) against the build before; fp64 lines as controls; then the fp32-heavy tests
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r05fu}; LIBS="${2:-tools/experiments/libs/libhipims_mi_final1.so cur}"
B="python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --no-strict-leg --repeats 2"
L() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-36s %-10s %9.1f Mcell-steps/s  %.4f ms/step  frac %.3f' % ('$1', '$2', d['value'], d['ms_per_step'], d['roofline']['frac']))"; }
run() { name=$1; shift
  for rep in 1 2; do for lib in $LIBS; do
    if [ "$lib" = cur ]; then $B "$@" | L "$name" cur; else HIPIMS_MI_LIB=$PWD/$lib $B "$@" | L "$name" $(basename $lib .so | sed s/libhipims_mi_//); fi
  done; done; }
{
run "S-DAM 4096^2 godunov f32" --precision f32
run "S-DAM 4096^2 godunov f32 singles" --precision f32 --workload s-dam --evolve-steps 1
run "S-RAIN 4096^2 godunov f32" --workload s-rain --precision f32
run "S-RAIN 8192^2 godunov f32 (C5)" --cols 8192 --rows 8192 --steps 100 --workload s-rain --precision f32
run "S-RAIN 8192x1026 strip f32 (C5)" --cols 8192 --rows 1026 --steps 200 --workload s-rain --precision f32
run "S-DAM 8192x1026 strip f32" --cols 8192 --rows 1026 --steps 200 --precision f32
run "S-DAM 4096^2 godunov f64" 
run "S-ROUGH 4096^2 godunov f64" --workload s-rough
run "S-DAM 4096x514 strip f64" --cols 4096 --rows 514
} 2>&1 | tee gpurun_out/${TAG}_ab.txt
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -6 | tee gpurun_out/${TAG}_pytest.txt
