#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; mkdir -p $OUT
for rep in 1 2 3; do for lib in libhipims_mi.so libhipims_mi_w6.so; do
  for args in "--precision f32" "--precision f32 --workload s-rain" "--precision f32 --workload s-rain --cols 8192 --rows 8192 --steps 100" "--precision f32 --workload s-rough"; do
    HIPIMS_MI_LIB=$PWD/hipims-ocl_amd/lib/$lib python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --repeats 2 $args | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib [$args]', round(d['roofline']['avg_launch_ms'],4), round(d['value']))"
  done; done; done 2>&1 | tee $OUT/ab.txt
