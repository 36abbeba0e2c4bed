#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; mkdir -p $OUT
python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
grep -E "passed|failed|FAILED|rc=" $OUT/pytest.log | head -20
B="python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg"
for rep in 1 2; do for f in 1 0; do
  HP_FUSE_BDY=$f $B --workload s-rain | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fuse=$f s-rain f64 4096', 'step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['avg_launch_ms'],4), 'value', round(d['value']))"
  HP_FUSE_BDY=$f $B --workload s-rain --evolve-steps 1500 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fuse=$f s-rain f64 4096 developed', 'step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['avg_launch_ms'],4), 'value', round(d['value']))"
  HP_FUSE_BDY=$f $B --workload s-rain --precision f32 --cols 8192 --rows 8192 --steps 100 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fuse=$f s-rain f32 8192', 'step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['avg_launch_ms'],4), 'value', round(d['value']))"
done; done 2>&1 | tee $OUT/fuse_ab.txt
