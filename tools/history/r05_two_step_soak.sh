#!/bin/bash
# round 5: pairs forced on against pairs off over N random configurations (tests/two_step_fuzz_worker.py), every observable hashed
# usage: tools/r05_two_step_soak.sh <first seed> <count> <tag>
cd "${GRAFT_REPO_ROOT:-/root/repo}"
FIRST=${1:-1000}; N=${2:-300}; TAG=${3:-r05_two_step_soak}
HP_TWO_STEP=0 python tests/two_step_fuzz_worker.py $FIRST $N 2>/dev/null | grep '^seed' | sed 's/  # .*//' > /tmp/soak_single.txt
HP_TWO_STEP=1 python tests/two_step_fuzz_worker.py $FIRST $N 2>/dev/null | grep '^seed' > /tmp/soak_pairs_raw.txt
sed 's/  # .*//' /tmp/soak_pairs_raw.txt > /tmp/soak_pairs.txt
{
echo "two-step soak: seeds $FIRST .. $((FIRST + N - 1)): $(wc -l < /tmp/soak_single.txt) single lines, $(wc -l < /tmp/soak_pairs.txt) pair lines"
if diff -q /tmp/soak_single.txt /tmp/soak_pairs.txt > /dev/null; then echo "ALL $(wc -l < /tmp/soak_pairs.txt) configurations bit-identical (pairs forced on vs off)"; else echo "MISMATCHES:"; diff /tmp/soak_single.txt /tmp/soak_pairs.txt | head -20; fi
echo "configurations that ended in non-finite states (a fixed timestep beyond the CFL limit; not compared): $(grep -c non-finite /tmp/soak_pairs.txt)"
awk '{n+=$NF; it+=$8} END {print "iterations", it, "flux launches with pairs", n}' /tmp/soak_pairs_raw.txt
} | tee gpurun_out/${TAG}.txt
