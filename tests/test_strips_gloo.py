"""world_size-2 (and 3) CPU tests of the strip decomposition protocol over gloo: a decomposed run must be
bit-identical to the single-domain run, for Godunov (1 ghost row) and MUSCL-Hancock (2)."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

import oracle
from hipims_mi import strips, synthetic as syn


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, scheme, cols, rows, steps, q):
    import functools
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from strip_oracle_engine import OracleStripEngine
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    st, bed, man = syn.s_rough(cols, rows, manning=None)
    oscheme = scheme                                  # same numbering: 0 Godunov, 1 MUSCL-Hancock, 2 inertial
    factory = functools.partial(OracleStripEngine, scheme=oscheme)
    r = strips.StripRunner(cols, rows, scheme=scheme, rank=rank, world=world, engine_factory=factory)
    r.upload_global(st, bed, man)
    r.set_target_time(2.0)          # the sync point falls inside the run: suspended iterations on every rank
    r.step(steps)
    full = r.gather_owned()
    sc = r.engine.scalars()
    if rank == 0:
        q.put((full, sc))
    r.close()


@pytest.mark.parametrize("scheme,world", [(strips.SCHEME_GODUNOV, 2), (strips.SCHEME_GODUNOV, 3),
                                          (strips.SCHEME_MUSCL_HANCOCK, 2), (strips.SCHEME_INERTIAL, 2)])
def test_decomposed_run_is_bit_identical(scheme, world):
    cols, rows, steps = 40, 36, 90
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, scheme, cols, rows, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    full, sc = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    st, bed, man = syn.s_rough(cols, rows, manning=None)
    oscheme = scheme
    quirks = oracle.QUIRKS_REFERENCE & ~oracle.Q6_MUSCL_SERIAL
    single = oracle.OracleSim(cols, rows, scheme=oscheme, quirks=quirks)
    single.upload(st, bed, man)
    single.set_target(2.0)
    single.run(steps)
    assert np.array_equal(full, single.download())
    s1 = single.scalars()
    assert (sc["t"], sc["dt"], sc["batch_ok"], sc["batch_skipped"]) == (s1["t"], s1["dt"], s1["batch_ok"], s1["batch_skipped"])
    assert s1["batch_skipped"] > 0


def test_partition_covers_grid_once():
    for rows, world, g in ((4096, 8, 1), (8192, 8, 2), (37, 3, 2)):
        parts = strips.partition(rows, world, g)
        assert parts[0][0] == 0 and parts[-1][1] == rows
        for (a0, a1, l0, l1), (b0, b1, m0, m1) in zip(parts, parts[1:]):
            assert a1 == b0 and l1 == a1 + g and m0 == b0 - g


def test_partition_with_two_reaches_of_ghost_rows_and_period_needs_the_cxx_loop():
    """exchange_period = 2 stores 2g ghost rows per interior side (partition's third argument) and belongs to the library's
    own strip loop: the torch loop of this module exchanges after every iteration and must refuse it."""
    from hipims_mi import strips
    parts = strips.partition(100, 3, 2)
    assert parts == [(0, 33, 0, 35), (33, 66, 31, 68), (66, 100, 64, 100)]
    with pytest.raises(ValueError, match="thinner than its halo"):
        strips.partition(12, 4, 2)
    with pytest.raises(ValueError, match="exchange_period"):
        strips.StripRunner(32, 64, world=1, exchange_period=3, engine_factory=lambda *a: None, init_process_group=False)
    with pytest.raises(ValueError, match="C\\+\\+ strip loop"):
        strips.StripRunner(32, 64, world=1, exchange_period=2, engine_factory=lambda *a: None, init_process_group=False)
