#!/bin/bash
# round 4: fp64 tile height on grids between one round and six (HP_MARCH_RSEG / HP_MUSCL_RSEG forced; "default" = what the library picks)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --no-strict-leg --repeats 2"
P() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-58s %-8s %.4f ms  frac %.3f' % ('$1', '$2', d['ms_per_step'], d['roofline']['frac']))"; }
for a in "--cols 2048 --rows 2048" "--cols 1024 --rows 1024" "--cols 4096 --rows 1026" "--cols 4096 --rows 2050" "--cols 8192 --rows 1026" "--cols 16384 --rows 1026 --steps 100" "--cols 2048 --rows 2048 --workload s-rain" "--cols 4096 --rows 1026 --workload s-rough"; do
  HP_PRINT_TILING=1 $B $a 2>/tmp/til.txt | P "f64 $a" "default"; grep -m1 tiling /tmp/til.txt
  for r in 8 10 12 14 16 20; do HP_MARCH_RSEG=$r $B $a 2>/dev/null | P "f64 $a" $r; done
done
for a in "--scheme muscl --cols 2048 --rows 2048" "--scheme muscl --cols 4096 --rows 1028" "--scheme muscl --cols 8192 --rows 1028"; do
  $B $a 2>/dev/null | P "f64 $a" "default"
  for r in 8 10 12 16 20; do HP_MUSCL_RSEG=$r $B $a 2>/dev/null | P "f64 $a" $r; done
done
