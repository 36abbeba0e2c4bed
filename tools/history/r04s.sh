#!/bin/bash
# round 4: the workload lines for a library variant.  usage: tools/r04s.sh <tag> [short]   (HIPIMS_MI_LIB selects the library)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1; SHORT=${2:-}
B="python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --no-strict-leg --repeats 2"
L() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s %9.1f Mcell-steps/s  %.4f ms/step  frac %.3f' % ('$1', d['value'], d['ms_per_step'], d['roofline']['frac']))"; }
{
$B | L "S-DAM 4096^2 godunov f64 [20,220)"
$B --evolve-steps 1500 | L "S-DAM 4096^2 godunov f64 [1520,1720)"
$B --scheme muscl | L "S-DAM 4096^2 muscl f64 [20,220)"
$B --scheme muscl --evolve-steps 1500 | L "S-DAM 4096^2 muscl f64 [1520,1720)"
$B --workload s-rain | L "S-RAIN 4096^2 godunov f64, rain fused"
$B --workload s-rough | L "S-ROUGH 4096^2 godunov f64"
$B --workload s-rough --scheme muscl | L "S-ROUGH 4096^2 muscl f64"
$B --math strict | L "S-DAM 4096^2 godunov f64 STRICT"
$B --math strict --workload s-rain | L "S-RAIN 4096^2 godunov f64 STRICT"
$B --math strict --scheme muscl --workload s-rough | L "S-ROUGH 4096^2 muscl f64 STRICT"
if [ -z "$SHORT" ]; then
$B --scheme inertial | L "S-DAM 4096^2 (2|1.6 m) inertial f64"
$B --precision f32 | L "S-DAM 4096^2 godunov f32"
$B --scheme muscl --precision f32 | L "S-DAM 4096^2 muscl f32"
$B --workload s-rain --evolve-steps 1500 | L "S-RAIN 4096^2 godunov f64 [1520,1720)"
$B --workload s-rain --precision f32 | L "S-RAIN 4096^2 godunov f32"
$B --math strict --scheme muscl --evolve-steps 1500 | L "S-DAM 4096^2 muscl f64 STRICT [1520,1720)"
$B --cols 8192 --rows 8192 --steps 100 | L "S-DAM 8192^2 godunov f64"
$B --cols 8192 --rows 8192 --steps 100 --scheme muscl | L "S-DAM 8192^2 muscl f64"
$B --cols 8192 --rows 8192 --steps 100 --workload s-rain --precision f32 | L "S-RAIN 8192^2 godunov f32 (C5)"
$B --cols 16384 --rows 8192 --steps 100 | L "S-DAM 16384x8192 godunov f64 (C4 whole)"
$B --cols 16384 --rows 1026 --steps 100 | L "S-DAM 16384x1026 godunov f64 (C4 strip)"
fi
} 2>&1 | tee gpurun_out/${TAG}_workloads.txt
