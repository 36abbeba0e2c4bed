#!/bin/bash
# round 4: the tile heights the library now picks (fp64 rule by rounds), against the classic 16 (K1) / 12 (K2) rows
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --no-strict-leg --repeats 3"
P() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-58s %-8s %.4f ms  frac %.3f' % ('$1', '$2', d['ms_per_step'], d['roofline']['frac']))"; }
for a in "" "--cols 2048 --rows 2048" "--cols 1024 --rows 1024" "--cols 4096 --rows 514" "--cols 4096 --rows 1026" "--cols 4096 --rows 2050" "--cols 8192 --rows 1026" "--cols 16384 --rows 1026 --steps 100" "--cols 4096 --rows 1026 --workload s-rough" "--cols 2048 --rows 2048 --math strict"; do
  HP_MARCH_RSEG=16 $B $a 2>/dev/null | P "f64 $a" "16 rows"
  HP_PRINT_TILING=1 $B $a 2>/tmp/til.txt | P "f64 $a" "default"; grep -m1 tiling /tmp/til.txt
done
for a in "--scheme muscl" "--scheme muscl --cols 2048 --rows 2048" "--scheme muscl --cols 4096 --rows 1028" "--scheme muscl --cols 8192 --rows 1028" "--scheme muscl --cols 4096 --rows 516"; do
  HP_MUSCL_RSEG=12 $B $a 2>/dev/null | P "f64 $a" "12 rows"
  HP_PRINT_TILING=1 $B $a 2>/tmp/til.txt | P "f64 $a" "default"; grep -m1 tiling /tmp/til.txt
done
