#!/bin/bash
# usage: tools/knob_sweep.sh <out.jsonl> "<bench args>" "ENV1=a ENV2=b" "ENV1=c" ...   (one bench line per env set, same box)
OUT=$1; ARGS=$2; shift 2
: > "$OUT"
for envset in "$@"; do
  line=$(env $envset timeout 300 python3 bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg $ARGS 2>&1 | grep '^{' | tail -1)
  echo "{\"env\": \"$envset\", \"bench\": ${line:-null}}" >> "$OUT"
done
python3 - "$OUT" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    b = d["bench"]
    print(f"{d['env']:50s}", "FAILED" if not b else f"{b['value']:9.0f} Mcs/s  {b['ms_per_step']:.4f} ms/step  kernel {b['roofline']['avg_launch_ms']:.4f} ms  frac {b['roofline']['frac']:.3f}")
PY
