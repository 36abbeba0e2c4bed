#!/bin/bash
# round 4: fp32 tile height (HP_MARCH_RSEG / HP_MUSCL_RSEG) on whole-domain launches, S-DAM and the general-path workloads
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --no-strict-leg --repeats 2 --precision f32"
P() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-52s rseg %-3s %.4f ms  frac %.3f' % ('$1', '$2', d['ms_per_step'], d['roofline']['frac']))"; }
for a in "" "--workload s-rain" "--workload s-rough" "--workload s-rain --cols 8192 --rows 8192 --steps 100" "--workload s-rain --cols 8192 --rows 1026"; do
  for r in 8 12 16 20 24 32; do HP_MARCH_RSEG=$r $B $a 2>/dev/null | P "K1 f32 $a" $r; done
done
for a in "--scheme muscl" "--scheme muscl --workload s-rough"; do
  for r in 8 12 16 24 32; do HP_MUSCL_RSEG=$r $B $a 2>/dev/null | P "K2 f32 $a" $r; done
done
B64="python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --no-strict-leg --repeats 2"
for a in "" "--workload s-rain" "--workload s-rough"; do
  for r in 10 12 14 16 20; do HP_MARCH_RSEG=$r $B64 $a 2>/dev/null | P "K1 f64 $a" $r; done
done
