#!/bin/bash
# round 4: alternating sweep direction (HP_SWEEP_ALTERNATE) off / on, same box, same library
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --no-strict-leg --repeats 3"
for a in "" "--evolve-steps 1500" "--scheme muscl" "--scheme muscl --workload s-rough" "--workload s-rain" "--workload s-rough" "--precision f32" "--scheme inertial" "--math strict" "--cols 8192 --rows 8192 --steps 100" "--cols 16384 --rows 1026 --steps 100" "--cols 2048 --rows 2048"; do
  for v in 0 1 0 1; do
    HP_SWEEP_ALTERNATE=$v $B $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s alternate=%s  %.4f ms  frac %.3f' % ('$a', '$v', d['ms_per_step'], d['roofline']['frac']))"
  done
done
