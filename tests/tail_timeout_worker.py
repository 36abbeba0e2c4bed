"""Worker of tests/test_gpu_fallback_paths.py::test_tail_block_gives_up_instead_of_hanging (run with HP_DEBUG_TAIL_EXTRA_WORD=1
and a short HP_TAIL_TIMEOUT_MS): the launch's tail block waits for one word nobody writes; the batch must END, time must stand
still, and the library must refuse the domain from then on with HP_ERR_STATE."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hipims-ocl_amd"))
import hipims_mi as hp                                      # noqa: E402
from hipims_mi import synthetic as syn                      # noqa: E402

st, bed, man = syn.s_rough(342, 195, manning=None)
dom = hp.Domain(342, 195)
dom.upload(st, bed, man)
dom.set_target_time(1e9)
t0 = time.perf_counter()
dom.step_batch(3)                                           # three launches: the first gives up, the other two return at once
try:
    dom.sync()
    print("sync succeeded: the time-out did not fire")
    sys.exit(1)
except hp.HipimsError as e:
    el = time.perf_counter() - t0
    assert "(-5)" in str(e) and "tail block" in str(e), str(e)
    print(f"hp_sync: {e}  [{el * 1e3:.0f} ms]")
for call in (dom.read_scalars, lambda: dom.step_batch(1), dom.download):
    try:
        call()
        print("a call on the failed domain succeeded")
        sys.exit(1)
    except hp.HipimsError as e:
        assert "(-5)" in str(e), str(e)
dom.close()
# the GPU is not left spinning: a fresh domain of the same process runs (and, the debug word still being on, gives up the same way)
dom = hp.Domain(342, 195)
dom.upload(st, bed, man)
dom.set_target_time(1e9)
dom.step_batch(3)
try:
    dom.read_scalars()
    print("second domain: no time-out")
    sys.exit(1)
except hp.HipimsError as e:
    assert "(-5)" in str(e), str(e)
dom.close()
print(f"tail time-out ok after {el * 1e3:.0f} ms")
