#!/bin/bash
# round 4: the exact mode's bench lines (what the reference's bits cost): S-DAM, S-RAIN, MUSCL developed flood, MUSCL S-ROUGH, S-RAIN fp32 8192^2
line() { python bench.py --math strict --no-cpu-baseline --no-manning-leg --no-moving-leg --repeats 2 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-90s %8.4f ms/step  frac %.3f' % (d['config']['workload'][:90], d['ms_per_step'], d['roofline']['frac']))"; }
line --workload s-dam
line --workload s-rain
line --workload s-rough
line --scheme muscl --workload s-dam --evolve-steps 1500
line --scheme muscl --workload s-rough
line --workload s-rain --precision f32 --cols 8192 --rows 8192 --steps 100
