#!/bin/bash
# interleaved repeats of the tile height for K2 (HP_MUSCL_RSEG) or K6 (HP_INERTIAL_RSEG): tools/rseg_fine_sweep2.sh muscl|inertial "<values>"
S=$1; VAR=HP_MUSCL_RSEG; [ "$S" = "inertial" ] && VAR=HP_INERTIAL_RSEG
for rep in 1 2 3; do for r in $2; do
env $VAR=$r python bench.py --scheme $S --steps 300 --warmup 30 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('RSEG $S $r', round(d['value']), round(d['roofline']['avg_launch_ms'],4))"
done; done
