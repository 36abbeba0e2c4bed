#!/bin/bash
# round 4: K2 lines, two libraries on the same box.  usage: tools/r04t.sh <libA> <libB>
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --no-strict-leg --repeats 3 --scheme muscl"
for a in "" "--evolve-steps 1500" "--workload s-rough" "--precision f32" "--workload s-rough --precision f32" "--workload s-rain"; do
  for lib in "$1" "$2" "$1" "$2"; do
    HIPIMS_MI_LIB=$lib $B $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s %-50s %.4f ms  frac %.3f' % ('$a', '$lib'[-50:], d['ms_per_step'], d['roofline']['frac']))"
  done
done
