#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; mkdir -p $OUT
for rep in 1 2 3; do for lib in libhipims_mi.so libhipims_mi_ntls.so libhipims_mi_nts.so libhipims_mi_ntl.so; do
  HIPIMS_MI_LIB=$PWD/hipims-ocl_amd/lib/$lib python bench.py --no-cpu-baseline --repeats 2 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib default', round(d['roofline']['avg_launch_ms'],4), 'manning', round(d['roofline_manning_array']['avg_launch_ms'],4))"
done; done 2>&1 | tee $OUT/ab.txt
for lib in libhipims_mi.so libhipims_mi_ntls.so libhipims_mi_nts.so libhipims_mi_ntl.so; do
  HIPIMS_MI_LIB=$PWD/hipims-ocl_amd/lib/$lib python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --repeats 2 --scheme muscl | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib muscl', round(d['roofline']['avg_launch_ms'],4))"
done 2>&1 | tee -a $OUT/ab.txt
