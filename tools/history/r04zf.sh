#!/bin/bash
# same-box A/B of libraries on the fp32 K1 lines.  usage: tools/r04zf.sh <lib> <lib> ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --no-strict-leg --repeats 3 --precision f32"
P() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-52s %-36s %.4f ms  frac %.3f' % ('$1', '$2'[-36:], d['ms_per_step'], d['roofline']['frac']))"; }
for a in "" "--workload s-rain" "--workload s-rough" "--workload s-rain --cols 8192 --rows 8192 --steps 100" "--workload s-rain --cols 8192 --rows 1026" "--scheme inertial"; do
  for rep in 1 2; do for lib in "$@"; do HIPIMS_MI_LIB=$lib $B $a 2>/dev/null | P "$a" "$lib"; done; done
done
