#!/bin/bash
# same-box A/B of two libraries on the FAST fp64 K1 lines.  usage: tools/r04zx.sh <libA> <libB>
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --no-strict-leg --repeats 3"
P() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s %-44s %.4f ms  frac %.3f' % ('$1', '$2'[-44:], d['ms_per_step'], d['roofline']['frac']))"; }
for a in "" "--workload s-rain" "--workload s-rough" "--evolve-steps 1500" "--scheme inertial"; do
  for lib in "$1" "$2" "$1" "$2"; do HIPIMS_MI_LIB=$lib $B $a 2>/dev/null | P "$a" "$lib"; done
done
