#!/bin/bash
# round 5, closing run with the final binary: kernel trace + PMC passes of the default bench command (Godunov: the pair kernel) and of
# MUSCL-Hancock, of K1 / K2 on S-ROUGH and of K1 on S-RAIN (fp64 4096^2, fp32 8192^2); the workload lines; the default line; the GPU suite
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r05z}
OUT=gpurun_out
P() { name=$1; note=$2; shift 2
  tools/profile_bench.sh ${TAG}_prof_$name "$@" > $OUT/${TAG}_profile_$name.log 2>&1
  python tools/summarize_profile.py $OUT/${TAG}_prof_$name $OUT/${TAG}_$name "$note" >> $OUT/${TAG}_profile_$name.log 2>&1
  rm -rf $OUT/${TAG}_prof_$name; }
P godunov4096 "default bench command (S-DAM 4096^2 fp64 FAST: godunov_march2, two iterations per launch), final round-5 binary"
P muscl4096 "default bench command --scheme muscl, final round-5 binary" --scheme muscl
P godunov_srough "K1 FAST fp64 on S-ROUGH 4096^2 (pairs), final round-5 binary" --workload s-rough
P muscl_srough "K2 FAST fp64 on S-ROUGH 4096^2, final round-5 binary" --workload s-rough --scheme muscl
P godunov_srain "K1 FAST fp64 on S-RAIN 4096^2 (rain fused; single iterations), final round-5 binary" --workload s-rain
HP_TWO_STEP=0 P godunov_single4096 "default bench command with HP_TWO_STEP=0 (single iterations: godunov_march), final round-5 binary"
tools/r04s.sh ${TAG} > /dev/null 2>&1
python bench.py > $OUT/${TAG}_default_bench_line.json 2> $OUT/${TAG}_default_bench_line.err
rm -f $OUT/parity_numbers.jsonl
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|Error" | tail -3 | tee $OUT/${TAG}_pytest.txt
cp $OUT/parity_numbers.jsonl $OUT/${TAG}_parity_numbers.jsonl
