#!/bin/bash
# The exact mode's tuner under stress: every seed of tests/two_step_fuzz_worker.py in STRICT without boundaries, pairs possible on every grid
# and the tuner ON (HP_TWO_STEP=2) with a sample due every 12 iterations -- nearly every batch samples: three pairs, the repair of the other
# buffer, six single iterations, then whatever won -- against single iterations only (HP_TWO_STEP=0): every observable hashed, the lines
# must be identical.  usage: tools/r06_tuner_soak.sh <tag> [first seed] [count]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r06tuner}; FIRST=${2:-41000}; N=${3:-200}; OUT=gpurun_out/$TAG; mkdir -p $OUT
FUZZ_STRICT_TUNER=1 HP_TWO_STEP=0 python tests/two_step_fuzz_worker.py $FIRST $N 2>/dev/null | grep '^seed' | sed 's/  # .*//' > /tmp/tuner_single.txt
FUZZ_STRICT_TUNER=1 HP_TWO_STEP=2 HP_PAIR_TUNE_PERIOD=12 python tests/two_step_fuzz_worker.py $FIRST $N 2>/dev/null | grep '^seed' > /tmp/tuner_raw.txt
sed 's/  # .*//' /tmp/tuner_raw.txt > /tmp/tuner_tuned.txt
{ echo "tuner soak: seeds $FIRST .. $((FIRST + N - 1)): $(wc -l < /tmp/tuner_single.txt) single lines, $(wc -l < /tmp/tuner_tuned.txt) tuned lines"
  if diff -q /tmp/tuner_single.txt /tmp/tuner_tuned.txt > /dev/null; then echo "ALL $(wc -l < /tmp/tuner_tuned.txt) configurations bit-identical (tuner sampling every 12 iterations vs single iterations only)"; else echo "MISMATCHES: $(diff /tmp/tuner_single.txt /tmp/tuner_tuned.txt | grep -c '^<')"; diff /tmp/tuner_single.txt /tmp/tuner_tuned.txt | head -8; fi
  python3 - <<'PY'
import re
it = la = 0
for l in open("/tmp/tuner_raw.txt"):
    it += int(re.search(r"iterations (\d+)", l).group(1)); la += int(re.search(r"launches (\d+)", l).group(1))
print(f"iterations {it}, flux launches with the tuner {la} ({la / max(it, 1):.2f} per iteration: 1.00 = single iterations only, 0.50 = pairs only)")
PY
} | tee $OUT/tuner_soak.txt
