#!/bin/bash
# round 5, the last run with the final binary: the GPU suite, the default and the driver-style bench lines, K2's trace + PMC on S-ROUGH, the workload lines
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-r05zz}
OUT=gpurun_out
rm -f $OUT/parity_numbers.jsonl
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|Error" | tail -3 | tee $OUT/${TAG}_pytest.txt
cp $OUT/parity_numbers.jsonl $OUT/${TAG}_parity_numbers.jsonl
python bench.py > $OUT/${TAG}_default_bench_line.json 2> $OUT/${TAG}_default_bench_line.err
python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_driver_style_bench_line.json 2>> $OUT/${TAG}_default_bench_line.err
tools/profile_bench.sh ${TAG}_prof_muscl_srough --workload s-rough --scheme muscl > $OUT/${TAG}_profile_muscl_srough.log 2>&1
python tools/summarize_profile.py $OUT/${TAG}_prof_muscl_srough $OUT/${TAG}_muscl_srough "K2 FAST fp64 on S-ROUGH 4096^2, final round-5 binary" >> $OUT/${TAG}_profile_muscl_srough.log 2>&1
rm -rf $OUT/${TAG}_prof_muscl_srough
tools/r04s.sh ${TAG} > /dev/null 2>&1
python - $OUT/${TAG}_default_bench_line.json $OUT/${TAG}_driver_style_bench_line.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f)); r = d["roofline"]
    print(f.split("/")[-1], "value %.0f  ms/step %.4f  frac %.3f  it/launch %s  launches %s  event-sampled frac %.3f  traffic %s GB/s" % (
        d["value"], d["ms_per_step"], r["frac"], r["iterations_per_launch"], r["timed_region_launches"], r["frac_event_sampled"], r.get("traffic")))
    print("    moving_water %.3f (%.4f ms)  strict %.3f  manning array %.3f  cpu_baseline %.1f" % (
        d["moving_water"]["frac"], d["moving_water"]["ms_per_step"], d["strict"]["frac"], d["roofline_manning_array"]["frac"], d["cpu_baseline"]["value"]))
PY
