#!/bin/bash
# round 5, opening run with round 4's binary: the evidence VERDICT r04 found a round stale -- kernel trace + PMC passes of K1 on the
# MOVING-water workloads (S-ROUGH, S-RAIN fp64 4096^2; S-RAIN fp32 8192^2 = config C5's shape), the FAST price list, the short lines
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r05a}
OUT=gpurun_out
P() { name=$1; note=$2; shift 2
  tools/profile_bench.sh ${TAG}_prof_$name "$@" > $OUT/${TAG}_profile_$name.log 2>&1
  python tools/summarize_profile.py $OUT/${TAG}_prof_$name $OUT/${TAG}_$name "$note" >> $OUT/${TAG}_profile_$name.log 2>&1
  rm -rf $OUT/${TAG}_prof_$name; }
P k1_srough "K1 FAST fp64 on S-ROUGH 4096^2 (bench.py --workload s-rough), round-4 binary" --workload s-rough
P k1_srain "K1 FAST fp64 on S-RAIN 4096^2 (bench.py --workload s-rain), round-4 binary" --workload s-rain
P k1_srain_f32_8192 "K1 FAST fp32 on S-RAIN 8192^2 (C5's shape), round-4 binary" --workload s-rain --precision f32 --cols 8192 --rows 8192 --steps 100
python tools/pathbench/snap.py > $OUT/${TAG}_snap.txt 2>&1
tools/pathbench/pathbench_fast --iters 1000 /tmp/pathbench_sdam.bin /tmp/pathbench_srough.bin /tmp/pathbench_srain.bin > $OUT/${TAG}_pathbench_fast.txt 2>&1
tools/r04s.sh ${TAG} short > /dev/null 2>&1
