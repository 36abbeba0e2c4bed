#!/bin/bash
# round 5: the pair kernel's one-round tiling search on grids of about one round of blocks -- default (search) against single iterations
cd "${GRAFT_REPO_ROOT:-/root/repo}"
B="python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --no-strict-leg --repeats 3 --steps 400 --warmup 40"
L() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-24s %-22s %.2f us per iteration  frac %.3f  it/launch %d' % ('$1', '$2', d['ms_per_step']*1e3, d['roofline']['frac'], d['roofline']['iterations_per_launch']))"; }
{
for shape in "4096 514" "1448 1448" "2048 1024" "8192 514" "4096 770" "4096 1026" "3000 700" "1024 2050"; do
  set -- $shape
  HP_PRINT_TILING=1 python -c "
import sys; sys.path.insert(0,'hipims-ocl_amd')
import hipims_mi as hp
d=hp.Domain($1,$2); d.close()" 2>&1 | grep "pair kernel"
  for pass in 1 2; do
    HP_TWO_STEP=0 $B --cols $1 --rows $2 | L "$1 x $2" "single iterations"
    $B --cols $1 --rows $2 | L "$1 x $2" "default"
  done
done
} 2>&1 | tee gpurun_out/r05v_one_round_pairs.txt
HP_TWO_STEP=1 timeout 900 python tools/strip_fuzz.py 7000 300 2>&1 | tail -2 | tee gpurun_out/r05v_strip_fuzz_pairs_forced_300.txt
