#!/bin/bash
# time per cell of K1 against the grid shape (one box)
for shape in "2048 2048" "4096 2048" "2048 4096" "4096 4096" "8192 4096" "4096 8192" "8192 8192" "16384 4096" "4096 16384" "16384 8192"; do
set -- $shape
HP_TAIL_PCT=${HP_TAIL_PCT:-0} python bench.py --cols $1 --rows $2 --steps 200 --warmup 20 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ms=d['roofline']['avg_launch_ms']; c=$1*$2; print('SHAPE %5dx%-5d %8.4f ms  %6.2f ps/cell  %6.0f Mcell-steps/s' % ($1,$2,ms,ms*1e9/c,d['value']))"
done
