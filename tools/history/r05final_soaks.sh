#!/bin/bash
# round 5, the final binary: STRICT engine vs oracle over 20 000 fuzz cases, pairs forced on vs off over 400 configurations, strips
# (600 configurations by default, 300 with pairs forced), 64 big shapes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r05final}
HIPIMS_MI_FUZZ_CASES=20000 timeout 1500 python -m pytest tests/test_gpu_fuzz_strict.py -m gpu -q -n 8 2>&1 | tail -2 > gpurun_out/${TAG}_fuzz_soak_20000.txt
tools/r05_two_step_soak.sh 21000 400 ${TAG}_two_step_soak > /dev/null 2>&1
timeout 900 python tools/strip_fuzz.py 23000 600 2>&1 | tail -2 > gpurun_out/${TAG}_strip_fuzz_600.txt
HP_TWO_STEP=1 timeout 900 python tools/strip_fuzz.py 24000 300 2>&1 | tail -2 > gpurun_out/${TAG}_strip_fuzz_pairs_forced_300.txt
timeout 600 python tools/big_shape_fuzz.py 25000 64 2>&1 | tail -2 > gpurun_out/${TAG}_big_shape_fuzz.txt
for f in fuzz_soak_20000 two_step_soak strip_fuzz_600 strip_fuzz_pairs_forced_300 big_shape_fuzz; do echo "== $f"; cat gpurun_out/${TAG}_$f.txt; done
