#!/bin/bash
# interleaved repeats of the K1 tile height on one box (noise is ~1 % within a box)
for rep in 1 2 3; do for r in ${RSEGS:-12 13 14 15 16 17 18 19 20}; do
HP_MARCH_RSEG=$r python bench.py --steps 400 --warmup 40 --no-cpu-baseline "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('RSEG $r', round(d['value']), round(d['roofline']['avg_launch_ms'],4))"
done; done
