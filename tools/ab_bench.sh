#!/bin/bash
# A/B timing of library builds on ONE GPU box (box-to-box variation is +-3 %): tools/ab_bench.sh "<libs>" "<bench arg sets separated by ;>"
# libs are file names under hipims-ocl_amd/lib/; each (lib, args) pair is run REPS times, interleaved.
LIBS="$1"; IFS=';' read -ra SETS <<< "$2"; REPS=${REPS:-3}
for r in $(seq $REPS); do for set in "${SETS[@]}"; do for lib in $LIBS; do
HIPIMS_MI_LIB=$PWD/hipims-ocl_amd/lib/$lib python bench.py $set --steps 300 --warmup 30 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('AB', '$lib', '[$set]', round(d['value']), round(d['roofline']['avg_launch_ms'],4))"
done; done; done
