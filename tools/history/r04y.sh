#!/bin/bash
# round 4: STRICT lines, library A (HIPIMS_MI_LIB given as $1) against the current one, same box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --no-strict-leg --repeats 2 --math strict"
P() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s %-10s %.4f ms  frac %.3f' % ('$1', '$2', d['ms_per_step'], d['roofline']['frac']))"; }
for a in "" "--workload s-rain" "--workload s-rough" "--evolve-steps 1500" "--workload s-rain --precision f32" "--scheme muscl --workload s-rough"; do
  for r in 1 2; do
    HIPIMS_MI_LIB=$1 $B $a 2>/dev/null | P "$a" "A"
    $B $a 2>/dev/null | P "$a" "current"
  done
done
