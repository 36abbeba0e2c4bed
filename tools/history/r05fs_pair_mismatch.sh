#!/bin/bash
# round 5: two seeds of the pairs-vs-singles soak differ on the final binary -- which switch / which build, and where
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p /tmp/dump
hashes() { # <label> env...
  label=$1; shift
  for seed in 21182 21346; do
    a=$(env "$@" HP_TWO_STEP=0 python tests/two_step_fuzz_worker.py $seed 1 2>/dev/null | grep '^seed' | sed 's/  # .*//' | awk '{print $NF}')
    b=$(env "$@" HP_TWO_STEP=1 python tests/two_step_fuzz_worker.py $seed 1 2>/dev/null | grep '^seed' | sed 's/  # .*//' | awk '{print $NF}')
    if [ "$a" = "$b" ]; then echo "$label seed $seed: equal"; else echo "$label seed $seed: DIFFER"; fi
  done; }
{
hashes "final build"
hashes "final build, copy instead of FILL" HP_FILL_AFTER_PAIRS=0
hashes "final build, pair tiling 8 bands x 12 rows" HP_TILING_SEARCH=0 HP_MARCH2_RSEG=12
hashes "build m1 (four cuts, before the merged clamp)" HIPIMS_MI_LIB=$PWD/tools/experiments/libs/libhipims_mi_m1.so
hashes "build fill" HIPIMS_MI_LIB=$PWD/tools/experiments/libs/libhipims_mi_fill.so
hashes "build head" HIPIMS_MI_LIB=$PWD/tools/experiments/libs/libhipims_mi_head.so
for seed in 21182 21346; do
  FUZZ_DUMP=/tmp/dump HP_TWO_STEP=0 python tests/two_step_fuzz_worker.py $seed 1 > /dev/null 2>&1
  FUZZ_DUMP=/tmp/dump HP_TWO_STEP=1 python tests/two_step_fuzz_worker.py $seed 1 > /dev/null 2>&1
  python - $seed <<'PY'
import sys, numpy as np
seed = sys.argv[1]
a = np.load(f"/tmp/dump/seed{seed}_0.npz"); b = np.load(f"/tmp/dump/seed{seed}_1.npz")
print("seed", seed, "ops", a["ops"].tolist(), "iterations before each op", a["its"].tolist(), b["its"].tolist())
for k in range(len(a["ops"])):
    sa, sb = a["states"][k], b["states"][k]
    bad = np.argwhere(sa != sb)
    if len(bad):
        ys, xs, fs = bad[:, 0], bad[:, 1], bad[:, 2]
        print(f"  first difference in front of op #{k} (op {a['ops'][k]}; after op {a['ops'][k-1]}): {len(bad)} entries, rows {ys.min()}..{ys.max()}, cols {xs.min()}..{xs.max()}, fields {np.unique(fs).tolist()}, "
              f"max |diff| {np.nanmax(np.abs(sa - sb)):.3e}; dt {a['dts'][k]!r} / {b['dts'][k]!r}; t {a['times'][k]!r} / {b['times'][k]!r}")
        for y, x, f in bad[:6]:
            print(f"     [{y},{x}] f{f}: singles {sa[y, x, f]!r} pairs {sb[y, x, f]!r}   z-zb? state {sa[y, x].tolist()}")
        print("     distinct rows", np.unique(ys)[:20].tolist(), " distinct cols", np.unique(xs)[:20].tolist())
        break
PY
done
} 2>&1 | tee gpurun_out/r05fs_pair_mismatch.txt
