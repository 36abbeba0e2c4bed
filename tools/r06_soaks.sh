#!/bin/bash
# Round 6, the final binary: pairs (exact flavour forced: HP_PAIR_EXACT=1) against single iterations over N random configurations with
# area boundaries and STRICT cases (tests/two_step_fuzz_worker.py, every observable hashed); the same in the DEFAULT flavour on the
# configurations without boundaries; STRICT engine vs oracle fuzz; strips (default + pairs forced); big shapes.
# usage: tools/r06_soaks.sh <tag> [two-step count]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r06soak}; N=${2:-400}; OUT=gpurun_out/$TAG; mkdir -p $OUT
soak() { name=$1; first=$2; shift 2
  env HP_TWO_STEP=0 "$@" python tests/two_step_fuzz_worker.py $first $N 2>/dev/null | grep '^seed' | sed 's/  # .*//' > /tmp/soak_single.txt
  env HP_TWO_STEP=1 "$@" python tests/two_step_fuzz_worker.py $first $N 2>/dev/null | grep '^seed' > /tmp/soak_pairs_raw.txt
  sed 's/  # .*//' /tmp/soak_pairs_raw.txt > /tmp/soak_pairs.txt
  { echo "two-step soak ($name): seeds $first .. $((first + N - 1)): $(wc -l < /tmp/soak_single.txt) single lines, $(wc -l < /tmp/soak_pairs.txt) pair lines; with boundaries $(grep -c ' bdy ' /tmp/soak_pairs.txt), STRICT $(grep -c ' strict' /tmp/soak_pairs.txt)"
    if diff -q /tmp/soak_single.txt /tmp/soak_pairs.txt > /dev/null; then echo "ALL $(wc -l < /tmp/soak_pairs.txt) configurations bit-identical (pairs forced on vs off)"; else echo "MISMATCHES: $(diff /tmp/soak_single.txt /tmp/soak_pairs.txt | grep -c '^<')"; diff /tmp/soak_single.txt /tmp/soak_pairs.txt | head -12; fi
    echo "non-finite (not compared): $(grep -c non-finite /tmp/soak_pairs.txt)"
    awk '{n+=$NF} END {print "flux launches with pairs", n}' /tmp/soak_pairs_raw.txt
    awk '{n+=$NF} END {print "flux launches in single iterations", n}' <(env HP_TWO_STEP=0 true; sed 's/.*launches //' /tmp/soak_pairs_raw.txt | head -0; cat /dev/null) 2>/dev/null
  } | tee $OUT/two_step_soak_$name.txt; }
soak exact 31000 HP_PAIR_EXACT=1
soak default 31000
HIPIMS_MI_FUZZ_CASES=20000 timeout 1500 python -m pytest tests/test_gpu_fuzz_strict.py -m gpu -q -n 8 2>&1 | tail -2 > $OUT/fuzz_soak_20000.txt
timeout 900 python tools/strip_fuzz.py 33000 600 2>&1 | tail -2 > $OUT/strip_fuzz_600.txt
HP_TWO_STEP=1 timeout 900 python tools/strip_fuzz.py 34000 300 2>&1 | tail -2 > $OUT/strip_fuzz_pairs_forced_300.txt
timeout 600 python tools/big_shape_fuzz.py 35000 64 2>&1 | tail -2 > $OUT/big_shape_fuzz.txt
for f in fuzz_soak_20000 strip_fuzz_600 strip_fuzz_pairs_forced_300 big_shape_fuzz; do echo "== $f"; cat $OUT/$f.txt; done
