#!/bin/bash
# round 4: fp32 strips -- tile height x bands sweep on the shapes of an 8-GPU split (C5: 8192^2 fp32 over 8, 4, 2 GPUs; 4096^2)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for shape in "8192 1026" "8192 2050" "8192 4098" "4096 514" "4096 1026" "4096 2050"; do
  SWEEP_NB=8,9,16,18 SWEEP_RSEG=12,16,20,22,24,26,28,32 python tools/r04_band_sweep.py $shape 0 f32
done
