#!/bin/bash
# GPU box: round-3 starting point -- GPU test suite, the fixed membench, and the workloads VERDICT r02 names
# (K2 on S-ROUGH / developed flood, K1 on S-RAIN, the default line with and without a Manning array).
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r03a; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
tools/membench/membench 4096 > $OUT/membench_4096.txt 2>&1
tools/membench/membench 16384 > $OUT/membench_16384.txt 2>&1
python tools/diag_k2_paths.py > $OUT/diag_k2_paths.txt 2>&1
python bench.py --no-cpu-baseline > $OUT/bench_default.json 2>$OUT/bench_default.err
python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --scheme muscl > $OUT/bench_muscl.json 2>&1
python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --scheme muscl --evolve-steps 1500 > $OUT/bench_muscl_dev.json 2>&1
python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --workload s-rain > $OUT/bench_srain64.json 2>&1
python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --workload s-rain --precision f32 --cols 8192 --rows 8192 --steps 100 > $OUT/bench_srain32_8k.json 2>&1
head -c 600 $OUT/bench_default.json; echo
grep -h "unrolled x8 copy (plain loads) + bed" $OUT/membench_4096.txt | tail -3
cat $OUT/diag_k2_paths.txt
