#!/bin/bash
# round 5: the fused (rain) kernel's two loop copies against the single loop, same box, interleaved.
# usage: tools/r05fa.sh <tag> "<lib paths relative to the repo>" [pytest]      ("cur" = the library in hipims-ocl_amd/lib)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1; LIBS="$2"; PYT=${3:-}
B="python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --no-strict-leg --repeats 2"
L() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-38s %-30s %9.1f Mcell-steps/s  %.4f ms/step  frac %.3f' % ('$1', '$2', d['value'], d['ms_per_step'], d['roofline']['frac']))"; }
run() { name=$1; shift
  for rep in 1 2; do
  for lib in $LIBS; do
    if [ "$lib" = cur ]; then $B "$@" | L "$name" cur; else HIPIMS_MI_LIB=$PWD/$lib $B "$@" | L "$name" $(basename $lib .so); fi
  done; done; }
{
run "S-RAIN 4096^2 godunov f64" --workload s-rain
run "S-RAIN 4096^2 godunov f32" --workload s-rain --precision f32
run "S-RAIN 8192^2 godunov f32 (C5)" --cols 8192 --rows 8192 --steps 100 --workload s-rain --precision f32
run "S-RAIN 8192x1026 strip f32 (C5)" --cols 8192 --rows 1026 --steps 200 --workload s-rain --precision f32
run "S-RAIN 4096x514 strip f64" --cols 4096 --rows 514 --steps 200 --workload s-rain
run "S-DAM 4096^2 godunov f64 (control)"
} 2>&1 | tee gpurun_out/${TAG}_ab.txt
if [ -n "$PYT" ]; then
  timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 | tee gpurun_out/${TAG}_pytest.txt
fi
