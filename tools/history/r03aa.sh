#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; mkdir -p $OUT
for rep in 1 2 3; do for lib in libhipims_mi.so libhipims_mi_l2pf.so; do
  for args in "--scheme muscl" "--scheme muscl --evolve-steps 1500" "--scheme muscl --workload s-rough" "--scheme muscl --cols 8192 --rows 8192 --steps 100"; do
    HIPIMS_MI_LIB=$PWD/hipims-ocl_amd/lib/$lib python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --repeats 2 $args | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib [$args]', round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],3))"
  done; done; done 2>&1 | tee $OUT/ab.txt
