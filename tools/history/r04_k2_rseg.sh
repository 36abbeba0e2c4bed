#!/bin/bash
# round 4: K2 tile height on live water (S-ROUGH: every tile live), the developed flood and the bench window
line() { python bench.py --scheme muscl --no-cpu-baseline --no-manning-leg --no-moving-leg --no-strict-leg --repeats 2 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.4f ms/step  frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))"; }
for r in 12 16 20 24 32 48; do
  export HP_MUSCL_RSEG=$r
  echo "rseg $r: s-rough $(line --workload s-rough) | developed $(line --workload s-dam --evolve-steps 1500) | window $(line --workload s-dam)"
done
