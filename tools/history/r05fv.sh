#!/bin/bash
# round 5: the friction term's cube root with one third-order step against two Newton steps, on the lines where friction acts
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r05fv}; LIBS="${2:-tools/experiments/libs/libhipims_mi_final1.so cur}"
B="python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --no-strict-leg --repeats 2"
L() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-36s %-10s %9.1f Mcell-steps/s  %.4f ms/step  frac %.3f' % ('$1', '$2', d['value'], d['ms_per_step'], d['roofline']['frac']))"; }
run() { name=$1; shift
  for rep in 1 2; do for lib in $LIBS; do
    if [ "$lib" = cur ]; then $B "$@" | L "$name" cur; else HIPIMS_MI_LIB=$PWD/$lib $B "$@" | L "$name" $(basename $lib .so | sed s/libhipims_mi_//); fi
  done; done; }
{
run "S-ROUGH 4096^2 godunov f64" --workload s-rough
run "S-RAIN 4096^2 godunov f64" --workload s-rain
run "S-DAM developed godunov f64" --evolve-steps 1500
run "S-ROUGH 4096^2 muscl f64" --workload s-rough --scheme muscl
run "S-DAM 4096^2 godunov f64"
} 2>&1 | tee gpurun_out/${TAG}_ab.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_c1_full.py tests/test_gpu_c5.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -5 | tee gpurun_out/${TAG}_pytest.txt
