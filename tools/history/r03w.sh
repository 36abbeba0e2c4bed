#!/bin/bash
# GPU box: the round's record -- GPU suite, profiles (kernel trace + PMC) of the three bench workloads, the workload tables
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=$1
python -m pytest tests -m gpu -q > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${T}_pytest.log
grep -E "passed|failed|FAILED|rc=" gpurun_out/${T}_pytest.log | head
bash tools/profile_bench.sh ${T}_godunov4096
bash tools/profile_bench.sh ${T}_muscl4096 --scheme muscl
bash tools/profile_bench.sh ${T}_srain4096 --workload s-rain
bash tools/profile_bench.sh ${T}_srough_muscl4096 --workload s-rough --scheme muscl
bash tools/bench_lines.sh ${T}
for args in "--workload s-rough --scheme godunov" "--workload s-rough --scheme muscl" "--workload s-rain --evolve-steps 1500"; do
  line=$(timeout 600 python3 bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --steps 200 --warmup 20 $args 2>&1 | grep '^{' | tail -1)
  python3 -c "
import json, sys
b = json.loads(sys.argv[2]); print(f\"{sys.argv[1]:45s} {b['value']:9.0f} Mcs/s  step {b['ms_per_step']:.4f} ms  kernel {b['roofline']['avg_launch_ms']:.4f} ms  frac {b['roofline']['frac']:.3f}\")" "$args" "$line"
done | tee -a gpurun_out/${T}_bench.txt
bash tools/bench_lines_big.sh ${T}
python bench.py > gpurun_out/${T}_default_bench_line.json 2>gpurun_out/${T}_default_bench.err
tail -c 1500 gpurun_out/${T}_default_bench_line.json
