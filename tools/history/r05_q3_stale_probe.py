#!/usr/bin/env python3
"""round 5 (CPU, needs oracle/_ref): what the two-iterations kernel assumes about quirk Q3.

A cell whose whole neighbourhood is dry is left UNTOUCHED by the reference's flux kernel: the destination buffer keeps what it
held (a state two iterations old), and the next iteration reads that.  godunov_march2 keeps the intermediate state in registers and
lets such a cell pass its CURRENT state on instead.  This probe runs the reference's own kernels on wet/dry workloads and, at every
iteration that reads the primary buffer, (a) counts the untouched cells whose stale value in the other buffer DIFFERS from their
current value, and (b) overwrites the other buffer's untouched cells with the current values -- exactly the pair kernel's
semantics -- and compares the end state with an undisturbed run.  Result (profiles/r05_q3_stale_probe.txt): zero such cells,
identical end states, on S-ROUGH 64^2 x 200, S-ROUGH 256^2 x 600, the dry-bed dam break x 150 and config C1 x 900."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hipims-ocl_amd"), os.path.join(ROOT, "tests")]
import numpy as np      # noqa: E402
import oracle           # noqa: E402
from hipims_mi import synthetic as syn, frontend    # noqa: E402

VS = 1e-10


def dry5(state, bed):
    d = (state[..., 0] - bed) < VS
    m = np.zeros_like(d)
    m[1:-1, 1:-1] = d[1:-1, 1:-1] & d[:-2, 1:-1] & d[2:, 1:-1] & d[1:-1, :-2] & d[1:-1, 2:]
    return m & ~((state[..., 1] <= -9999.0) | (state[..., 0] == -9999.0))


def case(name, make, n):
    a = make(); a.run(n)
    b = make(); events = untouched = 0
    for _ in range(n):
        even = not b.use_alt                             # this iteration reads the primary buffer and writes the other one
        if even:
            m = dry5(b.primary, b.bed)
            untouched += int(m.sum())
            events += int((b.alt[m] != b.primary[m]).any(axis=-1).sum())
        b.run(1)
        if even:
            b.alt[m] = b.primary[m]                      # the pair kernel's semantics for the first iteration of a pair
    da, db = a.depth_velocity()[0], b.depth_velocity()[0]
    print(f"{name:24s} untouched cell-iterations {untouched:9d}  of which stale != current {events}  "
          f"end states equal {np.array_equal(a.download(), b.download())}  depth rmse {np.sqrt(np.mean((da - db) ** 2)):.1e}")


st, bed, man = syn.s_rough(64, 64, manning=None)
case("S-ROUGH 64^2 x 200", lambda: (lambda o: (o.upload(st, bed, man), o.set_target(1e9), o)[2])(oracle.RefSim(64, 64)), 200)
st2, bed2, man2 = syn.s_dam(96, 48, wet_right=False)
case("dry-bed dam break x 150", lambda: (lambda o: (o.upload(st2, bed2, man2), o.set_target(1e9), o)[2])(oracle.RefSim(96, 48)), 150)
from model_dir import make_newcastle     # noqa: E402
with tempfile.TemporaryDirectory() as tmp:
    cfg = frontend.parse_configuration(make_newcastle(tmp))
    st3, bed3, man3, res = frontend.build_domain(cfg)


def c1():
    o = oracle.RefSim(342, 195, dx=res, end_time=cfg.duration, threads=8)
    o.upload(st3, bed3, man3); frontend.attach_boundaries(cfg, o, 342); o.set_target(1e9)
    return o


case("config C1 x 900", c1, 900)
st4, bed4, man4 = syn.s_rough(256, 256, manning=None)
case("S-ROUGH 256^2 x 600", lambda: (lambda o: (o.upload(st4, bed4, man4), o.set_target(1e9), o)[2])(oracle.RefSim(256, 256, threads=8)), 600)
