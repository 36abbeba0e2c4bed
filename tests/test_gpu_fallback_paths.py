"""The engine's fused paths each keep a classic twin behind a switch; these legs keep the twins alive:
HP_LAUNCH_TAIL=0 -- the flux launch without its own tail block: atomic maxima + a separate advance launch (hp_kernels.hpp
launch_tail); HP_FUSE_BDY=0 -- rain / loss as the stand-alone boundary pass instead of the flux kernel's store
epilogue (K5).  (HP_PEER_DIRECT=0, ghost rows through the collective library's send / receive, is the peer_max = 1 leg of
tests/test_gpu_strips.py::test_cxx_strip_loop_with_several_ranks.)  Each leg: all three schemes, fp64 and fp32, uniform and
coarse gridded rain, a sync point inside the run, ragged batches -- STRICT, bit for bit against the oracle.
Round 4: HP_SWEEP_ALTERNATE=1 / 0 -- every other whole-domain launch visiting each band's tiles from the top down (TileMap::flip;
the default is on for fp32 and fp64 MUSCL only), forced on for every kernel and forced off."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
WORKER = os.path.join(os.path.dirname(__file__), "fallback_worker.py")


@pytest.mark.parametrize("env", [{}, {"HP_LAUNCH_TAIL": "0"}, {"HP_FUSE_BDY": "0"}, {"HP_LAUNCH_TAIL": "0", "HP_FUSE_BDY": "0"},
                                 {"HP_SWEEP_ALTERNATE": "1"}, {"HP_SWEEP_ALTERNATE": "0"}],
                         ids=["default", "no-tail-block", "stand-alone-rain", "both-off", "every-launch-pair-swept-both-ways", "never-flipped"])
def test_switched_off_variants_equal_the_oracle(env):
    r = subprocess.run([sys.executable, WORKER], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "fallback paths bit-identical" in r.stdout
    # the boundary switch really switched: the Godunov domain reports whether its area boundaries ride in the flux kernel
    assert ("fused=1" in r.stdout) == ("HP_FUSE_BDY" not in env) and ("fused=0" in r.stdout) == ("HP_FUSE_BDY" in env)


@pytest.mark.parametrize("peer_max", [2, 0])
def test_strips_without_the_tail_block(peer_max):
    """Row strips with HP_LAUNCH_TAIL=0: the advance launch carries the mailbox round (and the ghost-row push) again."""
    lib = os.path.join(os.path.dirname(__file__), "fake_rccl", "libfake_rccl.so")
    if not os.path.exists(lib):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", "-o", lib,
                               os.path.join(os.path.dirname(lib), "fake_rccl.cpp")])
    import hipims_mi as hp
    res = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "strip_threads_worker.py"), "3",
                          str(hp.SCHEME_GODUNOV), "f64", "1", "1", "1", "-2", str(peer_max)],
                         capture_output=True, text=True, timeout=600, env=dict(os.environ, HP_LAUNCH_TAIL="0"))
    assert res.returncode == 0, res.stdout + res.stderr
    assert "bit-identical True" in res.stdout


def test_tail_block_gives_up_instead_of_hanging():
    """The tail block's wait for the flux blocks' words is bounded: with one word that nobody writes
    (HP_DEBUG_TAIL_EXTRA_WORD=1) and a 100 ms limit the batch ends, hp_sync / hp_read_scalars / hp_step_batch / downloads fail
    with HP_ERR_STATE, and the process is not left with a spinning GPU."""
    worker = os.path.join(os.path.dirname(__file__), "tail_timeout_worker.py")
    r = subprocess.run([sys.executable, worker], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, HP_DEBUG_TAIL_EXTRA_WORD="1", HP_TAIL_TIMEOUT_MS="100"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "tail time-out ok" in r.stdout


def test_tail_timeout_is_off_the_default_path():
    """... and with a (short) limit but nothing missing, the default path runs as ever: the clock is only read for a word found
    empty, and a word that arrives late but in time is still folded."""
    r = subprocess.run([sys.executable, WORKER], capture_output=True, text=True, timeout=600, env=dict(os.environ, HP_TAIL_TIMEOUT_MS="2000"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "fallback paths bit-identical" in r.stdout
