#!/bin/bash
# round 5: which build breaks tests/spec_worker.py (speculative STRICT batches with the tail block)?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=${1:-6}
for lib in tools/experiments/libs/libhipims_mi_head.so tools/experiments/libs/libhipims_mi_fill.so hipims-ocl_amd/lib/libhipims_mi.so; do
  ok=0; bad=0
  for i in $(seq 1 $N); do
    if HIPIMS_MI_LIB=$PWD/$lib HP_STRICT_SPECULATE=1 timeout 120 python tests/spec_worker.py default > /tmp/spec_out.txt 2>&1; then ok=$((ok+1)); else bad=$((bad+1)); fi
  done
  echo "$lib: ok $ok failed $bad"
done 2>&1 | tee gpurun_out/r05fh.txt
