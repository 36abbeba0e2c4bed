#!/bin/bash
# A/B: fp32 FAST with refined reciprocal / square root (build flag HP_XP_F32_REFINE) -- accuracy (survey, fp32 part) and speed
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; mkdir -p $OUT
for lib in libhipims_mi.so libhipims_mi_f32r.so; do
  echo "== $lib"; HIPIMS_MI_LIB=$PWD/hipims-ocl_amd/lib/$lib timeout 1500 python tools/fast_deviation_survey.py 3000 2>&1 | grep -A9 "^f32" | cut -c1-330
done > $OUT/survey.txt 2>&1
for rep in 1 2; do for lib in libhipims_mi.so libhipims_mi_f32r.so; do
  for args in "--precision f32" "--precision f32 --workload s-rain" "--precision f32 --workload s-rain --cols 8192 --rows 8192 --steps 100" "--precision f32 --scheme muscl --workload s-rough"; do
    HIPIMS_MI_LIB=$PWD/hipims-ocl_amd/lib/$lib python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --repeats 2 $args | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib [$args]', round(d['roofline']['avg_launch_ms'],4), round(d['value']))"
  done; done; done > $OUT/ab.txt 2>&1
cat $OUT/survey.txt; sort $OUT/ab.txt
