#!/bin/bash
# GPU box: GPU tests after the shared cube root / prefetch fix, A/B of the serpentine tile order, PMC of the general paths.
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r03b; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
B="python bench.py --no-cpu-baseline"
for rep in 1 2; do
for s in 0 1; do
  HP_SERPENTINE=$s $B | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('serp=$s default', round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],4), 'manning', round(d['roofline_manning_array']['avg_launch_ms'],4), round(d['roofline_manning_array']['frac'],4))"
  HP_SERPENTINE=$s $B --no-manning-leg --no-moving-leg --scheme muscl | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('serp=$s muscl', round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],4))"
  HP_SERPENTINE=$s $B --no-manning-leg --no-moving-leg --workload s-rain | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('serp=$s srain', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],4))"
done; done 2>&1 | tee $OUT/serpentine_ab.txt
for w in "s-rough godunov" "s-rough muscl" "s-rain godunov" "s-dam muscl"; do set -- $w
  bash tools/quick_kernel.sh r03b_$1_$2 --workload $1 --scheme $2 2>&1 | tee -a $OUT/quick.txt
done
