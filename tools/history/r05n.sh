#!/bin/bash
# round 5: the two-iterations kernel -- tile height, waves per SIMD, and which workloads it helps (same box, interleaved)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r05n}
B="python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --no-strict-leg --repeats 2"
L() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s %-26s %9.1f Mcell-steps/s  %.4f ms/step  frac %.3f  it/launch %d' % ('$1', '$2', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['iterations_per_launch']))"; }
{
for r in 12 18 24 32 48; do HP_TWO_STEP=1 HP_MARCH2_RSEG=$r $B | L "S-DAM 4096^2 f64" "pairs rseg=$r"; done
HP_TWO_STEP=1 HP_MARCH2_RSEG=24 HIPIMS_MI_LIB=$PWD/tools/experiments/libs/libhipims_mi_k1bw3.so $B | L "S-DAM 4096^2 f64" "pairs rseg=24 3 waves"
HP_TWO_STEP=1 HP_MARCH2_RSEG=18 HIPIMS_MI_LIB=$PWD/tools/experiments/libs/libhipims_mi_k1bw3.so $B | L "S-DAM 4096^2 f64" "pairs rseg=18 3 waves"
HP_TWO_STEP=0 $B | L "S-DAM 4096^2 f64" "single"
for w in "--workload s-rough" "--evolve-steps 1500" "--precision f32" "--cols 8192 --rows 8192 --steps 100" "--cols 2048 --rows 2048" "--cols 1024 --rows 1024" "--cols 16384 --rows 1026 --steps 100" "--cols 4096 --rows 514"; do
  HP_TWO_STEP=0 $B $w | L "$w" "single"
  HP_TWO_STEP=1 $B $w | L "$w" "pairs rseg=24"
  HP_TWO_STEP=1 HP_MARCH2_RSEG=12 $B $w | L "$w" "pairs rseg=12"
done
} 2>&1 | tee gpurun_out/${TAG}_two_step.txt
python -m pytest tests/test_gpu_two_step.py -m gpu -q 2>&1 | tail -3 | tee -a gpurun_out/${TAG}_two_step.txt
