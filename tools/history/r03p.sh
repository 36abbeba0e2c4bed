#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; mkdir -p $OUT
for w in 4 2 3; do for i in 1 2 3 4 5 6; do python tests/strip_threads_worker.py $w 0 f32 1 1 1 -2 2>&1 | tail -2 | cut -c1-200; done; done | tee $OUT/flaky.txt
for i in 1 2 3 4; do python tests/strip_threads_worker.py 4 0 f32 1 1 2 -2 2>&1 | tail -2 | cut -c1-200; done | tee -a $OUT/flaky.txt
python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "passed|failed|FAILED|rc=" $OUT/pytest.log | head
python bench.py --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],4), 'manning', round(d['roofline_manning_array']['avg_launch_ms'],4))"
for P in 1 2; do PERIOD=$P python tools/strong_probe.py 4096 514 2>&1 | grep -E "strip|hp_"; done | tee $OUT/strong_probe.txt
