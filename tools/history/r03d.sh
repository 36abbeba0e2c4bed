#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; mkdir -p $OUT
python tools/pathbench/snap.py > $OUT/snap.txt 2>&1
for b in base new; do [ -x tools/pathbench/pathbench_$b ] && tools/pathbench/pathbench_$b --iters 1000 /tmp/pathbench_sdam.bin /tmp/pathbench_srough.bin /tmp/pathbench_srain.bin > $OUT/pathbench_$b.txt 2>&1; done
paste -d'|' $OUT/pathbench_base.txt $OUT/pathbench_new.txt | cut -c1-70,97-165 | tee $OUT/pathbench_ab.txt
B="python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg"
for w in "s-dam godunov" "s-rough godunov" "s-rain godunov" "s-dam muscl" "s-rough muscl"; do set -- $w
  $B --workload $1 --scheme $2 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', 'step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],4))"
done 2>&1 | tee $OUT/bench_lines.txt
$B --scheme muscl --evolve-steps 1500 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('s-dam muscl developed', 'step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],4))" | tee -a $OUT/bench_lines.txt
$B --workload s-rain --precision f32 --cols 8192 --rows 8192 --steps 100 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('s-rain f32 8192^2', 'step', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],4))" | tee -a $OUT/bench_lines.txt
python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
