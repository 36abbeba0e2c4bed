#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --no-strict-leg --repeats 3"
P() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-58s alternate=%s  %.4f ms  frac %.3f' % ('$1', '$2', d['ms_per_step'], d['roofline']['frac']))"; }
for a in "" "--evolve-steps 1500" "--scheme muscl" "--scheme inertial" "--math strict" "--cols 8192 --rows 8192 --steps 100" "--cols 16384 --rows 1026 --steps 100" "--cols 4096 --rows 514" "--cols 2048 --rows 2048" "--cols 16384 --rows 8192 --steps 60"; do
  for v in 0 1 0 1; do HP_SWEEP_ALTERNATE=$v $B $a 2>/dev/null | P "$a" $v; done
done
