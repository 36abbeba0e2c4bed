#!/bin/bash
# GPU box: the larger grids of the configs on one GPU (C5's 8192^2 fp32 rain; 8192^2 and C4's whole 16384 x 8192 in fp64)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-big}
for args in "--cols 8192 --rows 8192 --scheme godunov" "--cols 8192 --rows 8192 --scheme muscl" "--cols 8192 --rows 8192 --workload s-rain --precision f32" "--cols 8192 --rows 8192 --workload s-rain" "--cols 16384 --rows 8192 --scheme godunov" "--cols 16384 --rows 1026 --scheme godunov"; do
  line=$(timeout 900 python3 bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --steps 100 --warmup 10 $args 2>&1 | grep '^{' | tail -1)
  python3 - "$args" "$line" <<'PY'
import json, sys
a, l = sys.argv[1:3]
b = json.loads(l)
print(f"{a:60s} {b['value']:9.0f} Mcs/s  step {b['ms_per_step']:.4f} ms  kernel {b['roofline']['avg_launch_ms']:.4f} ms  frac {b['roofline']['frac']:.3f}")
PY
done | tee gpurun_out/${TAG}_bench_big.txt
