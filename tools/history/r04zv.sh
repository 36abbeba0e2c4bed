#!/bin/bash
# round 4: alternating sweep direction again, on the new tile heights
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --no-strict-leg --repeats 3"
P() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-58s alternate=%s  %.4f ms  frac %.3f' % ('$1', '$2', d['ms_per_step'], d['roofline']['frac']))"; }
for a in "--precision f32" "--precision f32 --workload s-rain" "--precision f32 --workload s-rough" "--precision f32 --workload s-rain --cols 8192 --rows 1026" "--precision f32 --workload s-rain --cols 8192 --rows 8192 --steps 100" "--precision f32 --scheme muscl" "--workload s-rain" "--workload s-rough" "--scheme muscl --workload s-rough" "--math strict --workload s-rain"; do
  for v in 0 1 0 1; do HP_SWEEP_ALTERNATE=$v $B $a 2>/dev/null | P "$a" $v; done
done
