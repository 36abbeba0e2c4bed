"""hp_crmath.h: the one cube root the oracle, the reference build's shim and the STRICT HIP kernels share.

Checked against EXACT rational arithmetic (no libm, no mpmath): y is the correctly rounded cube root of x iff
((y + y_prev)/2)^3 < x < ((y + y_next)/2)^3.  Runs on the CPU; the HIP side is compiled from the same header and is
compared bit for bit with this build in tests/test_gpu_strict_friction.py."""
import ctypes
import os
import subprocess
from fractions import Fraction

import numpy as np
import pytest

from conftest import ROOT

HDR = os.path.join(ROOT, "hipims-ocl_amd", "csrc", "hp_crmath.h")


@pytest.fixture(scope="module")
def libs(tmp_path_factory):
    d = tmp_path_factory.mktemp("crmath")
    src = d / "t.c"
    src.write_text('#include "%s"\n'
                   "void cbrt_many(const double* x, double* y, long n) { for (long i = 0; i < n; ++i) y[i] = hp_cr_cbrt(x[i]); }\n"
                   "void cbrtf_many(const float* x, float* y, long n) { for (long i = 0; i < n; ++i) y[i] = hp_cr_cbrtf(x[i]); }\n"
                   "void pow103_many(const double* x, double* y, long n) { for (long i = 0; i < n; ++i) y[i] = hp_cr_pow103(x[i]); }\n" % HDR)
    out = []
    for name, extra in (("plain", []), ("hwfma", ["-mfma"])):          # software fma() of libm vs the vfmadd instruction
        so = d / f"t_{name}.so"
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", *extra, "-o", str(so), str(src), "-lm"])
        out.append(ctypes.CDLL(str(so)))
    return out


def run(lib, fn, x):
    y = np.empty_like(x)
    getattr(lib, fn)(x.ctypes.data_as(ctypes.c_void_p), y.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(len(x)))
    return y


def correctly_rounded(x, y):
    X, Y = Fraction(float(x)), Fraction(float(y))
    hi = (Y + Fraction(float(np.nextafter(y, np.inf)))) / 2
    lo = (Y + Fraction(float(np.nextafter(y, 0.0)))) / 2
    return lo ** 3 < X < hi ** 3 or Y ** 3 == X


def test_cube_root_is_correctly_rounded_and_platform_independent(libs):
    rng = np.random.default_rng(3)
    x = np.concatenate([10 ** rng.uniform(-10, 3, 30000),              # the depths friction sees
                        rng.uniform(1, 8, 10000), 10 ** rng.uniform(-320, 300, 5000),
                        [5e-324, 2.0 ** -1022, 1e-10, 1e-9, 0.001, 1.0, 8.0, 27.0, 1.7976931348623157e308]])
    y = run(libs[0], "cbrt_many", x)
    assert np.array_equal(y, run(libs[1], "cbrt_many", x))             # no dependence on how fma is provided
    bad = [(a, b) for a, b in zip(x, y) if not correctly_rounded(a, b)]
    assert not bad, bad[:5]
    for lib in libs:
        e = run(lib, "cbrt_many", np.array([0.0, -0.0, -1.0, np.inf, np.nan]))
        assert e[0] == 0 and e[1] == 0 and np.isnan(e[2]) and e[3] == np.inf and np.isnan(e[4])   # what pow(x, 1/3) returns


def test_cube_root_is_a_conforming_pow(libs):
    """OpenCL asks pow for <= 16 ulp; against the infinitely precise pow(x, fl(1/3)) the routine is within 4.5 ulp over the
    depths of the path (0.5 ulp of its own + |ln x| * 2^-54/3 for the exponent not being exactly 1/3: 23 * 1.85e-17
    relative at the dry threshold 1e-10 m)."""
    from decimal import Decimal, getcontext
    getcontext().prec = 60
    rng = np.random.default_rng(4)
    x = 10 ** rng.uniform(-10, 3, 2000)
    y = run(libs[0], "cbrt_many", x)
    third = Decimal(1.0 / 3.0)                                         # the double the reference's source text denotes
    worst = 0.0
    for a, b in zip(x, y):
        exact = Decimal(float(a)) ** third
        ulp = Decimal(float(np.spacing(b)))
        worst = max(worst, float(abs(Decimal(float(b)) - exact) / ulp))
    assert worst < 4.5, worst


def test_float_flavour_and_pow103(libs):
    rng = np.random.default_rng(5)
    xf = (10 ** rng.uniform(-10, 3, 20000)).astype(np.float32)
    yf = run(libs[0], "cbrtf_many", xf)
    assert np.array_equal(yf, run(libs[0], "cbrt_many", xf.astype(np.float64)).astype(np.float32))
    assert np.abs(yf.astype(np.float64) / np.cbrt(xf.astype(np.float64)) - 1).max() < 6.1e-8
    x = 10 ** rng.uniform(-8, 2, 20000)
    assert np.array_equal(run(libs[0], "pow103_many", x), ((x * x) * x) * run(libs[0], "cbrt_many", x))
