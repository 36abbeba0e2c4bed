#!/bin/bash
# round 4, closing run with the final binary: kernel trace + PMC passes of the default bench command (Godunov, MUSCL), the workload
# lines, the default bench line, the GPU suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r04u}
tools/profile_bench.sh ${TAG}_prof_godunov > gpurun_out/${TAG}_profile_godunov.log 2>&1
python tools/summarize_profile.py gpurun_out/${TAG}_prof_godunov gpurun_out/${TAG}_godunov4096 "default bench command, final round-4 binary" >> gpurun_out/${TAG}_profile_godunov.log 2>&1
tools/profile_bench.sh ${TAG}_prof_muscl --scheme muscl > gpurun_out/${TAG}_profile_muscl.log 2>&1
python tools/summarize_profile.py gpurun_out/${TAG}_prof_muscl gpurun_out/${TAG}_muscl4096 "default bench command --scheme muscl, final round-4 binary" >> gpurun_out/${TAG}_profile_muscl.log 2>&1
rm -rf gpurun_out/${TAG}_prof_godunov gpurun_out/${TAG}_prof_muscl
tools/r04s.sh ${TAG} > /dev/null 2>&1
python bench.py > gpurun_out/${TAG}_default_bench_line.json 2> gpurun_out/${TAG}_default_bench_line.err
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|Error" | tail -3 | tee gpurun_out/${TAG}_pytest.txt
