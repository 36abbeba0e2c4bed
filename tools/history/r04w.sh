#!/bin/bash
# round 4: K1 still-water skip and the alternating sweep, same box: A = library before both, B = current library with
# HP_SWEEP_ALTERNATE=0, C = current library as shipped
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OLD=$PWD/tools/experiments/rol/libhipims_mi.so
B="python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --no-strict-leg --repeats 3"
P() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s %-28s %.4f ms  frac %.3f' % ('$1', '$2', d['ms_per_step'], d['roofline']['frac']))"; }
for a in "" "--evolve-steps 1500" "--workload s-rain" "--workload s-rough" "--math strict" "--precision f32" "--scheme muscl" "--scheme muscl --evolve-steps 1500" "--scheme inertial" "--cols 8192 --rows 8192 --steps 100" "--cols 16384 --rows 1026 --steps 100" "--cols 16384 --rows 8192 --steps 60"; do
  for r in 1 2; do
    HIPIMS_MI_LIB=$OLD $B $a 2>/dev/null | P "$a" "A before"
    HP_SWEEP_ALTERNATE=0 $B $a 2>/dev/null | P "$a" "B still-skip"
    $B $a 2>/dev/null | P "$a" "C still-skip + alternate"
  done
done
