#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; mkdir -p $OUT
for rep in 1 2; do for f in 1 0; do for shape in "4096 514" "4096 1026" "4096 2050" "2048 2048" "1024 1024" "8192 514"; do
  HP_RSEG_REFINE=$f python tools/strong_probe.py $shape 2>&1 | grep -E "hp_step_batch|split" | sed "s/^/refine=$f $shape /"
done; done; done 2>&1 | tee $OUT/refine_ab.txt
