#!/bin/bash
# A/B: how often the tail block polls (s_sleep 1 vs 16 between polls), K1 / K2 / K6 at 4096^2 with the tail block forced on for all three
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2; do for lib in libhipims_mi.so libhipims_mi_sleep16.so; do for args in "" "--scheme muscl" "--scheme inertial" "--workload s-rain"; do
  HP_TAIL_MAX_BLOCKS_K2K6=60000 HIPIMS_MI_LIB=$PWD/hipims-ocl_amd/lib/$lib python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --repeats 2 $args | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib [$args]', 'step %.4f' % d['ms_per_step'], 'kernel %.4f' % d['roofline']['avg_launch_ms'], round(d['value']))"
done; done; done 2>&1 | sort
for args in "--scheme muscl" "--scheme inertial"; do python bench.py --no-cpu-baseline --no-manning-leg --no-moving-leg --repeats 2 $args | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default (K2/K6 classic at this size) [$args]', 'step %.4f' % d['ms_per_step'], round(d['value']))"; done
