#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; mkdir -p $OUT
python -m pytest tests/test_gpu_strips.py tests/test_gpu_host_cpp.py -m gpu -q -x > $OUT/pytest_strips.log 2>&1; echo "rc=$?" >> $OUT/pytest_strips.log
grep -E "passed|failed|FAILED|rc=|Error" $OUT/pytest_strips.log | head -20
for rep in 1 2; do for P in 1 2; do PERIOD=$P python tools/strong_probe.py 4096 514 2>&1 | grep -E "strip|hp_"; done; done | tee $OUT/strong_probe.txt
PERIOD=1 python tools/strong_probe.py 16384 1026 2>&1 | grep -E "strip|hp_" | tee -a $OUT/strong_probe.txt
PERIOD=2 python tools/strong_probe.py 16384 1026 2>&1 | grep -E "strip|hp_" | tee -a $OUT/strong_probe.txt
