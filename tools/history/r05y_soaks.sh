#!/bin/bash
# round 5, final binary: STRICT engine vs oracle over 20 000 fuzz cases, 64 big shapes, 600 strip configurations; the default line again
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r05y}
HIPIMS_MI_FUZZ_CASES=20000 timeout 1500 python -m pytest tests/test_gpu_fuzz_strict.py -m gpu -q 2>&1 | tail -2 > gpurun_out/${TAG}_fuzz_soak_20000.txt
timeout 600 python tools/big_shape_fuzz.py 9000 64 2>&1 | tail -3 > gpurun_out/${TAG}_big_shape_fuzz.txt
timeout 900 python tools/strip_fuzz.py 7000 600 2>&1 | tail -3 > gpurun_out/${TAG}_strip_fuzz_600.txt
python bench.py > gpurun_out/${TAG}_default_bench_line.json 2> gpurun_out/${TAG}_default_bench_line.err
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_driver_style_bench_line.json 2>> gpurun_out/${TAG}_default_bench_line.err
B="python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --no-strict-leg --repeats 2"
L() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s %9.1f Mcell-steps/s  %.4f ms/step  frac %.3f' % ('$1', d['value'], d['ms_per_step'], d['roofline']['frac']))"; }
{ $B --scheme muscl | L "S-DAM 4096^2 muscl f64"; $B --scheme muscl --evolve-steps 1500 | L "S-DAM developed muscl f64"; $B --scheme muscl --workload s-rough | L "S-ROUGH 4096^2 muscl f64"; } > gpurun_out/${TAG}_muscl_lines.txt 2>&1
cat gpurun_out/${TAG}_fuzz_soak_20000.txt gpurun_out/${TAG}_big_shape_fuzz.txt gpurun_out/${TAG}_strip_fuzz_600.txt gpurun_out/${TAG}_muscl_lines.txt
