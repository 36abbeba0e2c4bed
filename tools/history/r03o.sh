#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; mkdir -p $OUT
echo "== old tree (before the strip-loop changes)" | tee $OUT/flaky.txt
for i in 1 2 3 4 5 6 7 8; do (cd tools/experiments/oldtree && python tests/strip_threads_worker.py 4 0 f32 1 1 2>&1 | tail -1 | cut -c1-120); done | tee -a $OUT/flaky.txt
echo "== new tree" | tee -a $OUT/flaky.txt
for i in 1 2 3 4 5 6; do python tests/strip_threads_worker.py 4 0 f32 1 1 1 -2 2>&1 | tail -2 | cut -c1-260; done | tee -a $OUT/flaky.txt
echo "== new tree, 2 and 3 ranks" | tee -a $OUT/flaky.txt
for w in 2 3; do for i in 1 2 3 4; do python tests/strip_threads_worker.py $w 0 f32 1 1 1 -2 2>&1 | tail -2 | cut -c1-260; done; done | tee -a $OUT/flaky.txt
