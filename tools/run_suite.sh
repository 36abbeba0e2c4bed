#!/bin/bash
# GPU box: full GPU test suite, then the bench lines of the main kernels (one box, one call)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-run}
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest.log 2>&1; grep -E "passed|failed" gpurun_out/${TAG}_pytest.log | tail -2
for args in "--scheme godunov" "--scheme muscl" "--scheme inertial" "--scheme godunov --precision f32" "--scheme muscl --precision f32" "--workload s-rain" "--workload s-rain --precision f32"; do
  line=$(timeout 300 python3 bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --steps 200 --warmup 20 $args 2>&1 | grep '^{' | tail -1)
  python3 - "$args" "$line" <<'PY'
import json, sys
a, l = sys.argv[1:3]
b = json.loads(l)
print(f"{a:45s} {b['value']:9.0f} Mcs/s  kernel {b['roofline']['avg_launch_ms']:.4f} ms  frac {b['roofline']['frac']:.3f}")
PY
done | tee gpurun_out/${TAG}_bench.txt
