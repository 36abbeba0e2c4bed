"""Worker of tests/test_gpu_strict_speculation.py (the switches are read once per process).
usage: spec_worker.py <mode>     mode: default | force | off | denormal"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hipims-ocl_amd")]
import hipims_mi as hp  # noqa: E402
import oracle  # noqa: E402
from hipims_mi import synthetic as syn  # noqa: E402

mode = sys.argv[1]
logs = []
hp.set_log_sink(lambda level, text: logs.append(text))
cols, rows = 190, 101
# (Godunov with a rain boundary: the kernel with FUSED boundaries is never speculated on -- hp_engine.hip: spec_wanted -- so that leg
# holds the switch-over to the plain kernels to the same bar; Godunov without and MUSCL-Hancock run the speculative flavours)
for scheme, rain in ((hp.SCHEME_GODUNOV, True), (hp.SCHEME_GODUNOV, False), (hp.SCHEME_MUSCL_HANCOCK, False)):
    st, bed, man = syn.s_rough(cols, rows, manning=None, seed=31)
    if mode == "denormal":
        # a moving film whose x-discharge is a DENORMAL number: qx / h has a numerator below 2^-969 -- v_div_scale rescales it and
        # the sequence must follow through v_div_fmas / v_div_fixup as the compiler's own division does
        wet = (st[..., 0] - bed) > 1e-3
        st[..., 2][wet] = 3e-310
    quirks = oracle.QUIRKS_REFERENCE & ~(oracle.Q6_MUSCL_SERIAL if scheme == hp.SCHEME_MUSCL_HANCOCK else 0)
    ref = oracle.OracleSim(cols, rows, scheme=scheme, quirks=quirks)
    dom = hp.Domain(cols, rows, scheme=scheme, math_mode=hp.MATH_STRICT)
    for s in (ref, dom):
        s.upload(st, bed, man)
        if rain:
            s.add_uniform(hp.UNIFORM_RAIN_INTENSITY, np.array([[0.0, 90.0], [5.0, 30.0], [10.0, 0.0]]), 5.0, 10.0)
    dom.set_target_time(0.8); ref.set_target(0.8)
    # batches on both sides of the speculation threshold (8), with every kind of call in between: each must first settle the
    # speculative batch in front of it
    for i, n in enumerate((40, 3, 9, 64, 8, 1, 30)):
        ref.run(n); dom.step_batch(n)
        if i == 1:
            assert not dom.is_busy() or True
        if i == 2:
            dom.set_target_time(1.7); ref.set_target(1.7)
        if i == 3:
            dom.update_timestep(); ref.update_timestep()
        if i % 2 == 0:
            a, b = dom.download(), ref.download()
            if not np.array_equal(a, b):                 # (what a failure looks like: which cells, which fields, how far apart)
                bad = np.argwhere(a != b)
                print("MISMATCH", scheme, i, n, "cells", len(bad), "first", bad[:6].tolist(), "max |diff|", float(np.nanmax(np.abs(a - b))),
                      "engine scalars", dom.read_scalars(), "oracle", ref.scalars(), "log", logs[-3:])
            assert np.array_equal(a, b), (scheme, i, n)
    sc, sr = dom.read_scalars(), ref.scalars()
    assert np.array_equal(dom.download(), ref.download())
    assert sc["time"] == sr["t"] and sc["timestep"] == sr["dt"] and sc["batch_successful"] == sr["batch_ok"], (sc, sr)
    assert sc["iterations"] == 155, sc
    dom.close()
replays = sum("re-run with the plain divisions" in l for l in logs)
print(f"replays={replays}")
print("speculative batches bit-identical")
