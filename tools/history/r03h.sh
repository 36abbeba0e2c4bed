#!/bin/bash
# A/B on ONE box: library variants x tile heights, interleaved repeats
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; mkdir -p $OUT
LIBS="${LIBS:-libhipims_mi.so libhipims_mi_d3.so}"
for rep in 1 2 3; do
 for lib in $LIBS; do
  for r in 18 19; do
   HP_MARCH_RSEG=$r HIPIMS_MI_LIB=$PWD/hipims-ocl_amd/lib/$lib python bench.py --no-cpu-baseline --repeats 2 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib rseg $r default', round(d['roofline']['avg_launch_ms'],4), 'manning', round(d['roofline_manning_array']['avg_launch_ms'],4))"
  done
  for w in s-rain s-rough; do
   HIPIMS_MI_LIB=$PWD/hipims-ocl_amd/lib/$lib python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --repeats 2 --workload $w | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib $w', round(d['roofline']['avg_launch_ms'],4))"
  done
 done
done 2>&1 | tee $OUT/ab.txt
