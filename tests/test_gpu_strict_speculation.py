"""STRICT fp64 batches can run SPECULATIVELY (round 4, HP_STRICT_SPECULATE=1): the flux kernels share refined reciprocals between quotients of one
denominator -- bit-identical to the plain IEEE divisions except for operands at the ends of the exponent range, which they detect
(one word per domain) but do not handle.  hp_step_batch snapshots the domain in front of such a batch and the next entry into the
library re-runs the batch with the plain kernels when the word is raised (hp_engine.hip: spec_begin / spec_resolve).
Every leg: STRICT engine == oracle, bit for bit, over batches on both sides of the speculation threshold with downloads,
a new target time and tst_UpdateTimestep in between."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
WORKER = os.path.join(os.path.dirname(__file__), "spec_worker.py")


def run(mode, **env):
    r = subprocess.run([sys.executable, WORKER, mode], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "speculative batches bit-identical" in r.stdout
    return int(r.stdout.split("replays=")[1].split()[0])


def test_speculative_batches_equal_the_oracle():
    assert run("default", HP_STRICT_SPECULATE="1") == 0                    # ordinary states never raise the word


def test_every_batch_replayed_still_equals_the_oracle():
    """HP_STRICT_SPEC_FORCE=1: every speculative batch is treated as flagged -- snapshot back, plain kernels, same bits; the
    counters, the ping-pong phase and the remembered CFL maxima must all have been put back for that to hold."""
    assert run("force", HP_STRICT_SPECULATE="1", HP_STRICT_SPEC_FORCE="1") >= 8


def test_speculation_is_opt_in():
    """Off unless HP_STRICT_SPECULATE=1 (the rigorous detection costs most of what the sharing saves: hp_engine.hip spec_wanted):
    forcing replays has nothing to replay."""
    assert run("off", HP_STRICT_SPEC_FORCE="1") == 0


def test_denormal_discharges_go_through_the_hardware_path():
    """A denormal discharge (3e-310) is a numerator below 2^-969: v_div_scale rescales it and sets vcc, and the shared-reciprocal
    sequence follows with v_div_fmas / v_div_fixup exactly as the compiler's own does -- no flag, no re-run, and the same bits as
    the oracle's x86 denormal arithmetic."""
    assert run("denormal", HP_STRICT_SPECULATE="1") == 0
