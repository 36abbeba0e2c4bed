#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; mkdir -p $OUT
python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
python bench.py --no-cpu-baseline > $OUT/bench_default.json 2>$OUT/bench_default.err
python -c "
import json; d=json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1])
print('default', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), round(d['roofline']['frac'],4), 'manning', round(d['roofline_manning_array']['avg_launch_ms'],4), round(d['roofline_manning_array']['frac'],4))" | tee $OUT/bench_lines.txt
B="python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg"
for w in "s-dam godunov --evolve-steps 1500" "s-rough godunov" "s-rain godunov" "s-dam muscl" "s-dam godunov --precision f32" "s-dam godunov --cols 8192 --rows 8192 --steps 100" "s-dam godunov --cols 16384 --rows 8192 --steps 60" "s-dam inertial"; do set -- $w
  $B --workload $1 --scheme $2 $3 $4 $5 $6 $7 $8 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', 'step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],4), 'value', round(d['value']))"
done 2>&1 | tee -a $OUT/bench_lines.txt
for r in 14 15 16 17 18 19 20 22 24; do
  HP_MARCH_RSEG=$r $B --repeats 2 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rseg $r', round(d['roofline']['avg_launch_ms'],4), round(d['ms_per_step'],4))"
done 2>&1 | tee $OUT/rseg_sweep.txt
