#!/bin/bash
# Round 6, the final binary, ONE box: the GPU suite, the smoke test, the driver's command and the default command of bench.py, the workload
# lines of DESIGN 6, the strip shapes.  usage: tools/r06_final.sh <tag>   -> gpurun_out/<tag>/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r06z}; OUT=gpurun_out/$TAG; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_style_bench_line.json 2> $OUT/driver_style.err
python bench.py > $OUT/default_bench_line.json 2> $OUT/default.err
Q="--no-cpu-baseline --no-manning-leg --no-moving-leg --no-strict-leg --no-config-legs --steps 200 --warmup 20"
line() { name=$1; shift; l=$(timeout 900 python3 bench.py $Q "$@" 2>/dev/null | grep '^{' | tail -1); python3 - "$name" "$l" <<'PY'
import json, sys
n, l = sys.argv[1:3]
b = json.loads(l); r = b["roofline"]
print(f"{n:58s} {b['value']:9.1f} Mcell-steps/s  {b['ms_per_step']:.4f} ms/step  frac {r['frac']:.3f}  it/launch {r['iterations_per_launch']}")
PY
}
{
line "S-DAM 4096^2 godunov f64 [20,220)"
line "S-DAM 4096^2 godunov f64 [1520,1720)" --evolve-steps 1500
line "S-DAM 4096^2 muscl f64 [20,220)" --scheme muscl
line "S-DAM 4096^2 muscl f64 [1520,1720)" --scheme muscl --evolve-steps 1500
line "S-RAIN 4096^2 godunov f64 (pairs, rain in registers)" --workload s-rain --warmup 50
line "S-ROUGH 4096^2 godunov f64" --workload s-rough
line "S-ROUGH 4096^2 muscl f64" --workload s-rough --scheme muscl
line "S-DAM 4096^2 godunov f64 STRICT (pairs by measurement)" --math strict --warmup 30
line "S-RAIN 4096^2 godunov f64 STRICT" --math strict --workload s-rain --warmup 50
line "S-ROUGH 4096^2 godunov f64 STRICT" --math strict --workload s-rough --warmup 30
line "S-ROUGH 4096^2 muscl f64 STRICT" --math strict --workload s-rough --scheme muscl
line "S-DAM 4096^2 (2|1.6 m) inertial f64" --scheme inertial
line "S-DAM 4096^2 godunov f32" --precision f32
line "S-DAM 4096^2 muscl f32" --scheme muscl --precision f32
line "S-RAIN 4096^2 godunov f32" --workload s-rain --precision f32 --warmup 50
line "S-DAM 8192^2 godunov f64" --cols 8192 --rows 8192 --steps 100
line "S-RAIN 8192^2 godunov f32 (C5, pairs)" --workload s-rain --precision f32 --cols 8192 --rows 8192 --warmup 50 --steps 100
line "S-DAM 16384x8192 godunov f64 (C4 whole)" --cols 16384 --rows 8192 --steps 60 --repeats 2
line "S-DAM 16384x1026 godunov f64 (C4 strip)" --cols 16384 --rows 1026
line "S-DAM 4096x514 godunov f64 strip, pairs" --cols 4096 --rows 514
HP_TWO_STEP=0 line "S-DAM 4096x514 godunov f64 strip, single iterations" --cols 4096 --rows 514
line "S-RAIN 8192x1026 godunov f32 strip (C5 share)" --workload s-rain --precision f32 --cols 8192 --rows 1026 --warmup 50
HP_PAIR_EXACT=1 line "S-DAM 4096^2 godunov f64, every pair exact" 
HP_PAIR_EXACT=1 line "S-ROUGH 4096^2 godunov f64, every pair exact" --workload s-rough
} | tee $OUT/workloads.txt
# the strong-scaling strip: alone, and cut in two with thread ranks over the strips' own transport (pairs: two reaches of ghost rows)
{ echo "== tools/strong_probe.py (one rank's share of 4096^2 / 8, batch call and strip loop)"; timeout 600 python tools/strong_probe.py 2>&1 | grep -v amdgpu.ids | tail -12;
  echo "== PERIOD=2 tools/strong_probe.py"; PERIOD=2 timeout 600 python tools/strong_probe.py 2>&1 | grep -v amdgpu.ids | tail -12;
  echo "== tools/strong_probe_pair.py"; timeout 600 python tools/strong_probe_pair.py 2>&1 | grep -v amdgpu.ids | tail -12; } > $OUT/strong_probes.txt 2>&1
tail -5 $OUT/strong_probes.txt
