#!/bin/bash
# round 5: tests/spec_worker.py in a loop -- the speculative STRICT batches' test failed twice in some twenty suite runs (first compare,
# Godunov with rain): how often, under which switches, and what the mismatch looks like
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r05fg}; N=${2:-40}
loop() { name=$1; shift; ok=0; bad=0
  for i in $(seq 1 $N); do
    if env "$@" timeout 120 python tests/spec_worker.py default > /tmp/spec_out.txt 2>&1; then ok=$((ok+1)); else bad=$((bad+1)); grep -h MISMATCH /tmp/spec_out.txt | cut -c1-900; fi
  done
  echo "$name: ok $ok failed $bad"; }
{
loop "speculate"                 HP_STRICT_SPECULATE=1
loop "speculate, no tail block"  HP_STRICT_SPECULATE=1 HP_LAUNCH_TAIL=0
loop "no speculation"            HP_STRICT_SPECULATE=0
} 2>&1 | tee gpurun_out/${TAG}.txt
