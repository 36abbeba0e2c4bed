#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; mkdir -p $OUT
for rep in 1 2; do for r in 18 24 36 54; do
  HP_MARCH_RSEG=$r python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --repeats 2 --cols 16384 --rows 8192 --steps 40 --warmup 10 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('16384x8192 rseg $r', round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],4))"
done; done 2>&1 | tee $OUT/tall_tiles.txt
