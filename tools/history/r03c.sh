#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r03c; mkdir -p $OUT
python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -8 $OUT/pytest.log
python tools/pathbench/snap.py 2>&1 | tee $OUT/snap.txt
tools/pathbench/pathbench /tmp/pathbench_sdam.bin /tmp/pathbench_srough.bin /tmp/pathbench_srough_muscl.bin /tmp/pathbench_srain.bin 2>&1 | tee $OUT/pathbench.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 20 --warmup 5 --prewarm-s 0.1 --repeats 1 --no-cpu-baseline --no-manning-leg --no-moving-leg > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 20 --warmup 5 --prewarm-s 0.1 --repeats 1 --no-cpu-baseline --no-manning-leg --no-moving-leg > $OUT/pmc_write.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for what in ("fetch", "write"):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/r03c/pmc_{what}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "march" in k[0]: print(what, k, len(v), sum(v) / len(v))
PY
