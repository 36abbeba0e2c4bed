#!/bin/bash
# GPU box: bench lines of the main kernels, early (steps [20,220) from t=0) and developed (after 1500 more steps) flood
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-lines}
for args in "--scheme godunov" "--scheme godunov --evolve-steps 1500" "--scheme muscl" "--scheme muscl --evolve-steps 1500" "--scheme inertial" "--scheme godunov --precision f32" "--scheme muscl --precision f32" "--workload s-rain" "--workload s-rain --precision f32"; do
  line=$(timeout 600 python3 bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --steps 200 --warmup 20 $args 2>&1 | grep '^{' | tail -1)
  python3 - "$args" "$line" <<'PY'
import json, sys
a, l = sys.argv[1:3]
b = json.loads(l)
print(f"{a:45s} {b['value']:9.0f} Mcs/s  step {b['ms_per_step']:.4f} ms  kernel {b['roofline']['avg_launch_ms']:.4f} ms  frac {b['roofline']['frac']:.3f}  repeats {[round(x,4) for x in b['repeats_ms_per_step']]}")
PY
done | tee gpurun_out/${TAG}_bench.txt
