#!/bin/bash
# round 4: the one-round tiling search against the 8-band tiling on a range of shapes (default on / off, no sweep)
export SWEEP=0
for shape in "4096 514 0 f64" "4096 258 0 f64" "2048 514 0 f64" "1024 1024 0 f64" "512 512 0 f64" "2048 2048 0 f64" "8192 1026 0 f32" "4096 514 1 f64" "4096 516 1 f64" "1024 1024 1 f64" "4096 514 2 f64" "342 195 0 f64" "4096 1026 0 f64" "8192 514 0 f64"; do
  python tools/r04_band_sweep.py $shape
done
