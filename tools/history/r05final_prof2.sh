#!/bin/bash
# round 5, the final binary: kernel trace + PMC of K1 (single iterations, rain fused) on S-RAIN 4096^2 fp64 and on C5's shape (8192^2 fp32)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r05zzz}; OUT=gpurun_out
prof() { name=$1; note=$2; shift 2
  tools/profile_bench.sh ${TAG}_prof_$name "$@" > $OUT/${TAG}_profile_$name.log 2>&1
  python tools/summarize_profile.py $OUT/${TAG}_prof_$name $OUT/${TAG}_$name "$note" >> $OUT/${TAG}_profile_$name.log 2>&1
  rm -rf $OUT/${TAG}_prof_$name; }
prof godunov_srain "K1 FAST fp64 on S-RAIN 4096^2 (rain fused), final round-5 binary" --workload s-rain
prof godunov_srain_f32_8192 "K1 FAST fp32 on S-RAIN 8192^2 (config C5's shape), final round-5 binary" --workload s-rain --precision f32 --cols 8192 --rows 8192
python - <<'PY'
import json, csv
for name in ("godunov_srain", "godunov_srain_f32_8192"):
    d = json.load(open(f"gpurun_out/r05zzz_{name}_pmc.json"))
    for k, v in d["kernels"].items():
        if "godunov_march<" in k and v.get("launches_sampled", 0) >= 5:
            print(name, k[:60], "VALU %.1fM" % (v.get("SQ_INSTS_VALU", 0) / 1e6), "bytes %.4f GB" % (v.get("hbm_bytes_per_launch", 0) / 1e9), "n", v.get("launches_sampled"))
    for r in csv.DictReader(open(f"gpurun_out/r05zzz_{name}_kernel_stats.csv")):
        if "godunov_march<" in r.get("Name", "") and int(r["Calls"]) > 50:
            print("   trace", r["Name"][:60], r["Calls"], "avg %.1f us" % (float(r["AverageNs"]) / 1e3))
PY
