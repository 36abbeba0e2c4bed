#!/bin/bash
# One box, every workload/scheme/precision the bench knows: JSON lines into gpurun_out/<name>.jsonl
OUT=gpurun_out/${1:-workloads}.jsonl; mkdir -p gpurun_out; : > $OUT
run() { python bench.py --steps 300 --warmup 30 --no-cpu-baseline "$@" | tail -1 >> $OUT; }
run --scheme godunov
run --scheme godunov --math strict
run --scheme muscl
run --scheme muscl --math strict
run --scheme inertial
run --scheme godunov --precision f32
run --scheme muscl --precision f32
run --scheme inertial --precision f32
run --workload s-rain
run --workload s-rain --precision f32
run --workload s-rain --precision f32 --cols 8192 --rows 8192
run --scheme godunov --cols 16384 --rows 8192
run --scheme godunov --kernel basic --steps 40 --warmup 5
python -c "
import json,sys
for l in open('$OUT'):
    d=json.loads(l); r=d['roofline']
    print('%-110s %8.0f Mcell-steps/s  %.4f ms/launch  frac %.3f' % (d['config']['workload'][:110], d['value'], r['avg_launch_ms'], r['frac']))
"
