#!/bin/bash
# round 4: fp32 lines with the tile-height rule (default), against forced 32-row tiles (what round 3 shipped)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --no-strict-leg --repeats 2 --precision f32"
P() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s %-10s %.4f ms  frac %.3f' % ('$1', '$2', d['ms_per_step'], d['roofline']['frac']))"; }
for a in "" "--workload s-rain" "--workload s-rough" "--workload s-rain --cols 8192 --rows 8192 --steps 100" "--workload s-rain --cols 8192 --rows 1026" "--workload s-rain --cols 8192 --rows 2050" "--scheme inertial" "--cols 2048 --rows 2048" "--cols 1024 --rows 1024"; do
  HP_MARCH_RSEG=32 HP_INERTIAL_RSEG=32 $B $a 2>/dev/null | P "f32 $a" "32 rows"
  HP_PRINT_TILING=1 $B $a 2>/tmp/til.txt | P "f32 $a" "rule"; grep -m1 tiling /tmp/til.txt
done
for a in "--scheme muscl" "--scheme muscl --workload s-rough" "--scheme muscl --cols 8192 --rows 8192 --steps 100" "--scheme muscl --cols 8192 --rows 1028"; do
  HP_MUSCL_RSEG=32 $B $a 2>/dev/null | P "f32 $a" "32 rows"
  HP_PRINT_TILING=1 $B $a 2>/tmp/til.txt | P "f32 $a" "rule"; grep -m1 tiling /tmp/til.txt
done
