#!/bin/bash
# round 5, the final binary: kernel trace + PMC of the default bench command (godunov_march2 on S-DAM 4096^2), of K1b on S-ROUGH
# (the moving_water leg's kernel) and of the 4096 x 514 strip
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r05final}; OUT=gpurun_out
prof() { name=$1; note=$2; shift 2
  tools/profile_bench.sh ${TAG}_prof_$name "$@" > $OUT/${TAG}_profile_$name.log 2>&1
  python tools/summarize_profile.py $OUT/${TAG}_prof_$name $OUT/${TAG}_$name "$note" >> $OUT/${TAG}_profile_$name.log 2>&1
  rm -rf $OUT/${TAG}_prof_$name; tail -3 $OUT/${TAG}_profile_$name.log | cut -c1-300; }
prof godunov4096 "godunov_march2 FAST fp64 on S-DAM 4096^2 (the default bench command), final round-5 binary"
prof godunov_srough "godunov_march2 FAST fp64 on S-ROUGH 4096^2 (the moving_water leg), final round-5 binary" --workload s-rough
prof godunov_strip "godunov_march2 FAST fp64 on the 4096 x 514 strip, final round-5 binary" --cols 4096 --rows 514
