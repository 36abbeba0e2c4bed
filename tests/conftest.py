"""pytest configuration: markers, import paths, and building the CPU-side checkers."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "hipims-ocl_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the C oracle is test infrastructure: build it once per session (2 s with gcc)
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"])


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_golden(name):
    import numpy as np
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def record(name, **numbers):
    """Append achieved parity numbers to gpurun_out/parity_numbers.jsonl (scratch on the GPU box, merged back by
    gpurun) so that DESIGN.md can quote what the tolerances are actually met with."""
    import json
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_numbers.jsonl"), "a") as f:
            f.write(json.dumps(dict(case=name, **{k: (float(v) if hasattr(v, "__float__") else v) for k, v in numbers.items()})) + "\n")
    except OSError:
        pass
