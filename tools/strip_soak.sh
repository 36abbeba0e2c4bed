cd $GRAFT_REPO_ROOT
for cfg in "4 0 f64 1 0 1" "4 0 f64 1 1 1" "3 1 f64 1 0 1" "4 0 f32 1 1 2" "2 0 f64 1 0 1" "4 2 f64 1 0 1"; do
  for grid in "1000,1031,400" "4096,520,300"; do
    STRIP_WORKER_GRID=$grid timeout 600 python tests/strip_threads_worker.py $cfg -2 2 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-230
  done
done
