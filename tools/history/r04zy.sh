#!/bin/bash
# round 4: K1 fp64 4096^2 S-DAM, fine sweep of the tile height (interleaved, two passes)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --no-strict-leg --repeats 3"
P() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s rseg %-8s %.4f ms  frac %.3f' % ('$1', '$2', d['ms_per_step'], d['roofline']['frac']))"; }
for pass in 1 2; do
  $B $1 2>/dev/null | P "$1" default
  for r in 9 10 11 13 14 15 17 18 19 21 22 23 25 27; do HP_MARCH_RSEG=$r $B $1 2>/dev/null | P "$1" $r; done
done
