#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; mkdir -p $OUT
for f in 1 0; do for i in 1 2 3 4 5 6 7 8; do
  HP_FUSE_BDY=$f python tests/strip_threads_worker.py 4 0 f32 1 1 1 -2 2>&1 | tail -1 | cut -c1-140 | sed "s/^/fuse=$f /"
done; done | tee $OUT/flaky.txt
for i in 1 2 3 4; do HP_FUSE_BDY=1 python tests/strip_threads_worker.py 4 0 f32 0 1 1 -2 2>&1 | tail -1 | cut -c1-140 | sed "s/^/fuse=1 overlap=0 /"; done | tee -a $OUT/flaky.txt
for i in 1 2 3 4; do HP_FUSE_BDY=1 python tests/strip_threads_worker.py 4 0 f64 1 1 1 -2 2>&1 | tail -1 | cut -c1-140 | sed "s/^/fuse=1 f64 /"; done | tee -a $OUT/flaky.txt
