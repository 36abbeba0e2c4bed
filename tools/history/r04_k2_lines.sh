#!/bin/bash
# round 4: K2 (MUSCL-Hancock, FAST fp64) on the three regimes + fp32, one box
line() { python bench.py --scheme muscl --no-cpu-baseline --no-manning-leg --no-moving-leg --no-strict-leg --repeats 2 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%8.4f ms/step  frac %.3f' % (d['ms_per_step'], d['roofline']['frac']))"; }
echo "s-rough $(line --workload s-rough) | developed $(line --workload s-dam --evolve-steps 1500) | window $(line --workload s-dam) | fp32 window $(line --workload s-dam --precision f32) | fp32 s-rough $(line --workload s-rough --precision f32)"
