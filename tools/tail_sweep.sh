#!/bin/bash
# guided-tile sweep on one box: HP_TAIL_PCT x HP_TAIL_RSEG (x HP_MARCH_RSEG) for a scheme
SCHEME=${1:-godunov}; shift
for rs in ${RSEGS:-16}; do for pct in ${PCTS:-0 10 20 30 40}; do for tr in ${TAILS:-2 4 8}; do
HP_MARCH_RSEG=$rs HP_MUSCL_RSEG=${MRSEG:-32} HP_INERTIAL_RSEG=${IRSEG:-32} HP_TAIL_PCT=$pct HP_TAIL_RSEG=$tr python bench.py --scheme $SCHEME --steps 300 --warmup 30 --no-cpu-baseline "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TAIL $SCHEME rseg=$rs pct=$pct tail=$tr', round(d['value']), round(d['roofline']['avg_launch_ms'],4))"
[ "$pct" = "0" ] && break
done; done; done
