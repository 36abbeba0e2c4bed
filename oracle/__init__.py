"""oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front end to

* ``liboracle_f64.so`` / ``liboracle_f32.so`` : the plain-C restatement (``swe_oracle.c``) of the
  reference's shallow-water hot path, and
* ``_ref/libref_{god,mch,ine}_{f64,f64_mad,f32}.so`` : the reference's OWN OpenCL C kernel sources
  compiled for the host (``ref_build/``; only buildable where ``/root/reference`` exists; the
  built ``.so`` files travel to the GPU box, the sources do not).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package -- as the checker / reported baseline, never as the thing measured or shipped.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")
REFERENCE_SRC = "/root/reference/src"

GODUNOV, MUSCL, INERTIAL = 0, 1, 2
Q1_CFL_READS_PRIMARY, Q9_BDY_TRUNCATED, Q6_MUSCL_SERIAL = 1, 2, 4
# the reference's DEFAULT MUSCL predictor (kCachePrediction, CSchemeMUSCLHancock.cpp:46): neighbours reach mch_1st with their
# BED in .y (CLSchemeMUSCLHancock.clc:201, :232-248, :325-330); off = the mch_1st_cacheNone variant (neighbours' Zmax)
Q11_MUSCL_NB_Y_IS_BED = 8
QUIRKS_REFERENCE = 15


def quirks_to_engine(q: int) -> int:
    """Oracle quirk mask -> the engine's HP_QUIRK_* mask (include/hipims_mi.h): Q1 and Q9 share their bits, Q6 is the oracle's
    own (the engine is always snapshot order), Q11 is bit 3 here and bit 2 there."""
    return (q & 3) | (4 if q & Q11_MUSCL_NB_Y_IS_BED else 0)


def quirks_from_engine(q: int, muscl_serial: bool = False) -> int:
    return (q & 3) | (Q11_MUSCL_NB_Y_IS_BED if q & 4 else 0) | (Q6_MUSCL_SERIAL if muscl_serial else 0)


UNIFORM_RAIN_INTENSITY, UNIFORM_LOSS_RATE = 0, 1
GRIDDED_RAIN_INTENSITY, GRIDDED_RAIN_ACCUMUL, GRIDDED_MASS_FLUX = 0, 1, 2
DEPTH_IGNORE, DEPTH_IS_FSL, DEPTH_IS_DEPTH, DEPTH_IS_CRITICAL = 0, 1, 2, 3
DISCHARGE_IGNORE, DISCHARGE_IS_DISCHARGE, DISCHARGE_IS_VELOCITY, DISCHARGE_IS_VOLUME = 0, 1, 2, 3
DIR_N, DIR_E, DIR_S, DIR_W = 0, 1, 2, 3


def build(ref: bool | None = None) -> None:
    """Compile the C oracle; and oracle/_ref when the reference sources are present."""
    subprocess.check_call(["make", "-s", "-C", HERE, "all"])
    if ref is None:
        ref = os.path.isdir(REFERENCE_SRC)
    if ref:
        subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "ref_build"), "all"])


def have_ref(stem: str = "god_f64") -> bool:
    return os.path.exists(os.path.join(REF_DIR, f"libref_{stem}.so"))


def _np_real(precision: str):
    return np.float64 if precision == "f64" else np.float32


def _c_real(precision: str):
    return C.c_double if precision == "f64" else C.c_float


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _aligned_zeros(shape, dtype, align=64):
    """numpy only guarantees 16-byte alignment; the AVX (-mfma) reference build loads double4 with 32-byte moves."""
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    raw = np.zeros(n + align, np.uint8)
    off = (-raw.ctypes.data) % align
    return raw[off:off + n].view(dtype).reshape(shape)


# =================================================================================================
#  C restatement
# =================================================================================================
def _params_struct(creal):
    class Params(C.Structure):
        _fields_ = [("cols", C.c_long), ("rows", C.c_long), ("dx", creal), ("very_small", creal),
                    ("quite_small", creal), ("courant", creal), ("end_time", creal), ("fixed_dt", creal),
                    ("dynamic_dt", C.c_int), ("friction", C.c_int), ("threads", C.c_int), ("simplified_cfl", C.c_int),
                    ("muscl_nb_y_is_bed", C.c_int)]
    return Params


def _scalars_struct(creal):
    class Scalars(C.Structure):
        _fields_ = [("t", creal), ("dt", creal), ("t_hydro", creal), ("t_sync", creal), ("batch_dt", creal),
                    ("batch_ok", C.c_uint), ("batch_skipped", C.c_uint)]
    return Scalars


_LIBS: dict = {}


def _load_oracle(precision: str):
    if precision not in _LIBS:
        path = os.path.join(HERE, f"liboracle_{precision}.so")
        if not os.path.exists(path):
            build(ref=False)
        lib = C.CDLL(path)
        creal = _c_real(precision)
        lib.orc_limited_slope.restype = creal
        lib.orc_limited_slope.argtypes = [creal, creal, creal]
        lib.orc_cfl_max_speed.restype = creal
        lib.orc_inertial_flux.restype = creal
        lib.orc_inertial_flux.argtypes = [C.c_void_p] + [creal] * 7
        lib.orc_sim_create.restype = C.c_void_p
        _LIBS[precision] = lib
    return _LIBS[precision]


class OracleFunctions:
    """Function-level entry points of the C restatement (registers in, registers out)."""

    def __init__(self, precision="f64", very_small=1e-10, dx=1.0, friction=True, muscl_nb_y_is_bed=False):
        self.precision = precision
        self.lib = _load_oracle(precision)
        self.creal = _c_real(precision)
        self.real = _np_real(precision)
        self.Params = _params_struct(self.creal)
        self.p = self.Params(cols=0, rows=0, dx=dx, very_small=very_small, quite_small=very_small * 10,
                             courant=0.5, end_time=1e30, fixed_dt=0.0, dynamic_dt=1, friction=int(friction),
                             threads=1, muscl_nb_y_is_bed=int(muscl_nb_y_is_bed))

    def _a(self, x, n):
        a = np.ascontiguousarray(x, dtype=self.real)
        assert a.size == n
        return a

    def reconstruct(self, direction, sL, bL, sR, bR):
        sL, sR = self._a(sL, 4), self._a(sR, 4)
        oL, oR = np.zeros(8, self.real), np.zeros(8, self.real)
        stop = self.lib.orc_reconstruct(C.byref(self.p), int(direction), _ptr(sL), self.creal(bL), _ptr(sR),
                                        self.creal(bR), _ptr(oL), _ptr(oR))
        return oL, oR, stop

    def reconstruct2(self, direction, sL, bL, sR, bR, eL, eR):
        sL, sR, eL, eR = self._a(sL, 4), self._a(sR, 4), self._a(eL, 4), self._a(eR, 4)
        oL, oR = np.zeros(8, self.real), np.zeros(8, self.real)
        stop = self.lib.orc_reconstruct2(C.byref(self.p), int(direction), _ptr(sL), self.creal(bL), _ptr(sR),
                                         self.creal(bR), _ptr(eL), _ptr(eR), _ptr(oL), _ptr(oR))
        return oL, oR, stop

    def hllc(self, direction, L, R):
        L, R = self._a(L, 8), self._a(R, 8)
        F = np.zeros(4, self.real)
        self.lib.orc_hllc(C.byref(self.p), int(direction), _ptr(L), _ptr(R), _ptr(F))
        return F

    def friction(self, s, bed, n, dt):
        s = self._a(s, 4)
        out = np.zeros(4, self.real)
        self.lib.orc_friction(C.byref(self.p), _ptr(s), self.creal(bed), self.creal(n), self.creal(dt), _ptr(out))
        return out

    def limited_slope(self, l, c, r):
        return self.lib.orc_limited_slope(self.creal(l), self.creal(c), self.creal(r))

    def inertial_flux(self, n, dt, q_prev, z_up, b_up, z_down, b_down):
        return self.lib.orc_inertial_flux(C.byref(self.p), *(self.creal(v) for v in (n, dt, q_prev, z_up, b_up,
                                                                                     z_down, b_down)))

    def limiter(self, sL, sC, sR, bL, bC, bR):
        sL, sC, sR = self._a(sL, 4), self._a(sC, 4), self._a(sR, 4)
        out = np.zeros(4, self.real)
        self.lib.orc_limiter(C.byref(self.p), _ptr(sL), _ptr(sC), _ptr(sR), self.creal(bL), self.creal(bC),
                             self.creal(bR), _ptr(out))
        return out

    def mch_1st(self, dt, states, beds):
        states, beds = self._a(states, 20), self._a(beds, 5)
        faces = np.zeros(16, self.real)
        first = self.lib.orc_mch_1st(C.byref(self.p), self.creal(dt), _ptr(states), _ptr(beds), _ptr(faces))
        return faces.reshape(4, 4), first


class _SimBase:
    """Common surface of OracleSim (C restatement) and RefSim (reference kernels)."""

    def __init__(self, cols, rows, dx=1.0, scheme=GODUNOV, precision="f64", very_small=1e-10, courant=0.5,
                 end_time=1e30, dynamic_dt=True, fixed_dt=0.0, friction=True, quirks=QUIRKS_REFERENCE,
                 dt_initial=0.001, threads=1):
        self.cols, self.rows, self.dx = int(cols), int(rows), float(dx)
        self.scheme, self.precision = scheme, precision
        self.real, self.creal = _np_real(precision), _c_real(precision)
        self.very_small, self.courant, self.end_time = very_small, courant, end_time
        self.dynamic_dt, self.fixed_dt, self.friction = bool(dynamic_dt), fixed_dt, bool(friction)
        self.quirks, self.dt_initial, self.threads = quirks, dt_initial, threads

    def depth_velocity(self, state=None, bed=None):
        """Output derivation of the reference (src/Datasets/CRasterDataset.cpp:214-251)."""
        state = self.download() if state is None else state
        bed = self._bed_host if bed is None else bed
        depth = np.maximum(0.0, state[..., 0].astype(np.float64) - bed.astype(np.float64))
        with np.errstate(divide="ignore", invalid="ignore"):
            u = np.where(depth > 1e-8, state[..., 2] / depth, 0.0)
            v = np.where(depth > 1e-8, state[..., 3] / depth, 0.0)
        return depth, u, v


class OracleSim(_SimBase):
    """Simulation-level driver of the C restatement (orc_sim_*)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.lib = _load_oracle(self.precision)
        self.Params = _params_struct(self.creal)
        self.Scalars = _scalars_struct(self.creal)
        self.p = self.Params(cols=self.cols, rows=self.rows, dx=self.dx, very_small=self.very_small,
                             quite_small=self.real(self.very_small) * self.real(10), courant=self.courant,
                             end_time=self.end_time, fixed_dt=self.fixed_dt, dynamic_dt=int(self.dynamic_dt),
                             friction=int(self.friction), threads=self.threads,
                             simplified_cfl=int(self.scheme == INERTIAL))
        self.h = C.c_void_p(self.lib.orc_sim_create(C.byref(self.p), int(self.scheme), C.c_uint(self.quirks),
                                                    self.creal(self.dt_initial)))
        self._bed_host = None

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.orc_sim_destroy(self.h)
            self.h = None

    def upload(self, state=None, bed=None, manning=None):
        def prep(x, shape):
            if x is None:
                return None
            a = np.ascontiguousarray(x, dtype=self.real)
            assert a.shape == shape, (a.shape, shape)
            return a
        s = prep(state, (self.rows, self.cols, 4))
        b = prep(bed, (self.rows, self.cols))
        m = prep(manning, (self.rows, self.cols))
        if b is not None:
            self._bed_host = b.copy()
        self.lib.orc_sim_upload(self.h, _ptr(s) if s is not None else None, _ptr(b) if b is not None else None,
                                _ptr(m) if m is not None else None)

    def add_uniform(self, definition, series, interval, length):
        s = np.ascontiguousarray(series, dtype=self.real)
        assert s.ndim == 2 and s.shape[1] == 2
        return self.lib.orc_sim_add_uniform(self.h, int(definition), _ptr(s), C.c_uint(s.shape[0]),
                                            self.creal(interval), self.creal(length))

    def add_gridded(self, definition, grids, resolution, off_x, off_y, interval):
        g = np.ascontiguousarray(grids, dtype=self.real)
        assert g.ndim == 3
        return self.lib.orc_sim_add_gridded(self.h, int(definition), _ptr(g), C.c_ulong(g.shape[0]),
                                            C.c_ulong(g.shape[1]), C.c_ulong(g.shape[2]), self.creal(resolution),
                                            self.creal(off_x), self.creal(off_y), self.creal(interval))

    def add_cell(self, depth_def, discharge_def, cells, series, interval, length):
        """cells: flat cell ids (y*cols + x); series: [entries][4] = time, depth/fsl, qx, qy (already per cell)."""
        rel = np.ascontiguousarray(cells, dtype=np.uint64)
        ser = np.ascontiguousarray(series, dtype=self.real)
        assert ser.ndim == 2 and ser.shape[1] == 4
        return self.lib.orc_sim_add_cell(self.h, int(depth_def), int(discharge_def), _ptr(rel), C.c_ulong(rel.size),
                                         _ptr(ser), C.c_ulong(ser.shape[0]), self.creal(interval), self.creal(length))

    def set_target(self, t):
        self.lib.orc_sim_set_target(self.h, self.creal(t))

    def force_dt(self, dt):
        self.lib.orc_sim_force_dt(self.h, self.creal(dt))

    def reset_counters(self):
        self.lib.orc_sim_reset_counters(self.h)

    def update_timestep(self):
        self.lib.orc_sim_update_timestep(self.h)

    def run(self, n):
        trace = np.zeros(n, self.real)
        self.lib.orc_sim_run(self.h, C.c_long(n), _ptr(trace))
        return trace

    def scalars(self):
        sc = self.Scalars()
        self.lib.orc_sim_scalars(self.h, C.byref(sc))
        return dict(t=sc.t, dt=sc.dt, t_hydro=sc.t_hydro, t_sync=sc.t_sync, batch_dt=sc.batch_dt,
                    batch_ok=sc.batch_ok, batch_skipped=sc.batch_skipped)

    def download(self):
        out = np.zeros((self.rows, self.cols, 4), self.real)
        self.lib.orc_sim_download(self.h, _ptr(out))
        return out


# =================================================================================================
#  The reference's own kernels (oracle/_ref) + a restatement of its host-side iteration graph
# =================================================================================================
_REFLIBS: dict = {}


def _load_ref(stem: str):
    if stem not in _REFLIBS:
        lib = C.CDLL(os.path.join(REF_DIR, f"libref_{stem}.so"))
        lib.ref_configure.argtypes = [C.c_long, C.c_long, C.c_double, C.c_double, C.c_double, C.c_double,
                                      C.c_uint, C.c_double]
        _REFLIBS[stem] = lib
    return _REFLIBS[stem]


class RefFunctions:
    """Function-level entry points of the reference (through ref_build/wrappers.cl)."""

    def __init__(self, precision="f64", mad=False, very_small=1e-10, dx=1.0):
        sfx = precision + ("_mad" if mad else "")
        self.god = _load_ref("god_" + sfx)
        self.mch = _load_ref("mch_" + sfx)
        self.ine = _load_ref("ine_" + sfx) if have_ref("ine_" + sfx) else None
        self.real, self.creal = _np_real(precision), _c_real(precision)
        self.cfg = (8, 8, dx, very_small, 0.5, 1e30, 1, 0.0)

    def _conf(self, lib):
        lib.ref_configure(*self.cfg)

    def _a(self, x, n):
        a = np.ascontiguousarray(x, dtype=self.real)
        assert a.size == n
        return a

    def reconstruct(self, direction, sL, bL, sR, bR):
        self._conf(self.god)
        sL, sR = self._a(sL, 4), self._a(sR, 4)
        oL, oR, stop = np.zeros(8, self.real), np.zeros(8, self.real), C.c_int(0)
        self.god.refw_reconstruct(C.c_int(direction), _ptr(sL), self.creal(bL), _ptr(sR), self.creal(bR),
                                  _ptr(oL), _ptr(oR), C.byref(stop))
        return oL, oR, stop.value

    def reconstruct2(self, direction, sL, bL, sR, bR, eL, eR):
        self._conf(self.mch)
        sL, sR, eL, eR = self._a(sL, 4), self._a(sR, 4), self._a(eL, 4), self._a(eR, 4)
        oL, oR, stop = np.zeros(8, self.real), np.zeros(8, self.real), C.c_int(0)
        self.mch.refw_reconstruct2(C.c_int(direction), _ptr(sL), self.creal(bL), _ptr(sR), self.creal(bR),
                                   _ptr(eL), _ptr(eR), _ptr(oL), _ptr(oR), C.byref(stop))
        return oL, oR, stop.value

    def hllc(self, direction, L, R):
        self._conf(self.god)
        L, R = self._a(L, 8), self._a(R, 8)
        F = np.zeros(4, self.real)
        self.god.refw_hllc(C.c_int(direction), _ptr(L), _ptr(R), _ptr(F))
        return F

    def friction(self, s, bed, n, dt):
        self._conf(self.god)
        s = self._a(s, 4)
        out = np.zeros(4, self.real)
        self.god.refw_friction(_ptr(s), self.creal(bed), self.creal(n), self.creal(dt), _ptr(out))
        return out

    def limiter(self, sL, sC, sR, bL, bC, bR):
        self._conf(self.mch)
        sL, sC, sR = self._a(sL, 4), self._a(sC, 4), self._a(sR, 4)
        out = np.zeros(4, self.real)
        self.mch.refw_limiter(_ptr(sL), _ptr(sC), _ptr(sR), self.creal(bL), self.creal(bC), self.creal(bR),
                              _ptr(out))
        return out

    def mch_1st(self, dt, states, beds):
        self._conf(self.mch)
        states, beds = self._a(states, 20), self._a(beds, 5)
        faces, first = np.zeros(16, self.real), C.c_int(0)
        self.mch.refw_mch_1st(self.creal(dt), _ptr(states), _ptr(beds), _ptr(faces), C.byref(first))
        return faces.reshape(4, 4), first.value


    def inertial_flux(self, n, dt, q_prev, z_up, b_up, z_down, b_down):
        self._conf(self.ine)
        a = self._a([n, dt, q_prev, z_up, b_up, z_down, b_down], 7)
        out = np.zeros(1, self.real)
        self.ine.refw_inertial_flux(_ptr(a), _ptr(out))
        return out[0]


class RefSim(_SimBase):
    """The reference's kernels driven by a restatement of its HOST-side iteration graph.

    The per-iteration graph (boundaries -> flux -> reduce -> advance, ping-pong, quirk Q1) lives in
    the reference's C++ (``CSchemeGodunov::scheduleIteration`` :1617-1666,
    ``CSchemeMUSCLHancock::scheduleIteration`` :646-680), not in its ``.clc`` files, so it is
    restated here around the compiled kernels.  ``mad=True`` selects the build with
    ``-cl-mad-enable`` FMA contraction (what the reference ships, COCLProgram.cpp:73).
    """
    WORKERS = 64          # TIMESTEP_WORKERS; any value gives the same max

    def __init__(self, *a, mad=False, libm_pow=False, **k):
        super().__init__(*a, **k)
        stem = {GODUNOV: "god_", MUSCL: "mch_", INERTIAL: "ine_"}[self.scheme] + self.precision + ("_mad" if mad else "")
        if libm_pow:                                 # pow() from the host libm instead of hp_crmath.h (fp64 strict builds)
            assert stem in ("god_f64", "mch_f64", "ine_f64"), "libm-pow reference build: fp64 strict only"
            stem += "_libm"
        if not self.friction:                        # FRICTION_ENABLED undefined: Godunov / MUSCL fp64 builds exist
            assert not libm_pow
            assert stem in ("god_f64", "mch_f64") and self.dynamic_dt, "no-friction reference build: fp64 Godunov/MUSCL"
            stem += "_nofric"
        if not self.dynamic_dt:                      # the TIMESTEP_FIXED program exists for Godunov fp64 only
            assert stem == "god_f64", "fixed-timestep reference build: Godunov fp64 only"
            stem += "_fixed"
        self.lib = _load_ref(stem)
        n = self.rows * self.cols
        self.primary = _aligned_zeros((self.rows, self.cols, 4), self.real)
        self.alt = _aligned_zeros((self.rows, self.cols, 4), self.real)
        self.bed = np.zeros((self.rows, self.cols), self.real)
        self.manning = np.zeros((self.rows, self.cols), self.real)
        if self.scheme == MUSCL:
            self.faces = [_aligned_zeros((n, 4), self.real) for _ in range(4)]
        z = lambda v=0.0: np.array([v], self.real)
        self.t, self.dt, self.t_hydro, self.t_sync, self.batch_dt = z(), z(self.dt_initial), z(), z(), z()
        self.ok, self.skipped = np.zeros(1, np.uint32), np.zeros(1, np.uint32)
        self.scratch = np.zeros(self.WORKERS, self.real)
        self.use_alt = False
        self.bdy = []
        self._bed_host = None

    def _configure(self):
        self.lib.ref_configure(self.cols, self.rows, self.dx, self.very_small, self.courant, self.end_time,
                               self.WORKERS, self.fixed_dt)
        if hasattr(self.lib, "ref_set_threads"):
            self.lib.ref_set_threads(int(self.threads))            # rows of independent work-items over host threads

    def upload(self, state=None, bed=None, manning=None):
        if state is not None:
            self.primary[...] = state
            self.alt[...] = state
        if bed is not None:
            self.bed[...] = bed
            self._bed_host = self.bed.copy()
        if manning is not None:
            self.manning[...] = manning
        self.use_alt = False

    def add_uniform(self, definition, series, interval, length):
        s = np.ascontiguousarray(series, dtype=self.real)
        if self.precision == "f64":          # sBdyUniformConfiguration, CLBoundaries.clh:76-82
            class Cfg(C.Structure):
                _fields_ = [("entries", C.c_uint), ("interval", C.c_double), ("length", C.c_double),
                            ("definition", C.c_uint)]
        else:
            class Cfg(C.Structure):
                _fields_ = [("entries", C.c_uint), ("interval", C.c_float), ("length", C.c_float),
                            ("definition", C.c_uint)]
        self.bdy.append(("uniform", Cfg(s.shape[0], interval, length, definition), s))

    def add_gridded(self, definition, grids, resolution, off_x, off_y, interval):
        g = np.ascontiguousarray(grids, dtype=self.real)
        cr = self.creal

        class Cfg(C.Structure):              # sBdyGriddedConfiguration, CLBoundaries.clh:64-74
            _fields_ = [("interval", cr), ("resolution", cr), ("off_x", cr), ("off_y", cr),
                        ("entries", C.c_ulong), ("definition", C.c_ulong), ("grows", C.c_ulong),
                        ("gcols", C.c_ulong)]
        self.bdy.append(("gridded", Cfg(interval, resolution, off_x, off_y, g.shape[0], definition,
                                        g.shape[1], g.shape[2]), g))

    def add_cell(self, depth_def, discharge_def, cells, series, interval, length):
        rel = np.ascontiguousarray(cells, dtype=np.uint64)
        ser = _aligned_zeros((len(series), 4), self.real)
        ser[...] = series
        cr = self.creal

        class Cfg(C.Structure):              # sBdyCellConfiguration, CLBoundaries.clh:54-62
            _fields_ = [("entries", C.c_ulong), ("interval", cr), ("length", cr), ("count", C.c_ulong),
                        ("depth_def", C.c_uint), ("discharge_def", C.c_uint)]
        self.bdy.append(("cell", Cfg(ser.shape[0], interval, length, rel.size, depth_def, discharge_def), (rel, ser)))

    def set_target(self, t):
        self.t_sync[0] = t

    def force_dt(self, dt):
        self.dt[0] = dt

    def reset_counters(self):
        self.batch_dt[0] = 0
        self.ok[0] = 0
        self.skipped[0] = 0

    def update_timestep(self):
        self._configure()
        if self.dynamic_dt:
            self.lib.ref_reduce(_ptr(self.primary), _ptr(self.bed), _ptr(self.scratch))
        self.lib.ref_update_timestep(_ptr(self.t), _ptr(self.dt), _ptr(self.scratch), _ptr(self.t_sync), _ptr(self.batch_dt))

    def _apply_boundaries(self, target):
        full = 0 if (self.quirks & Q9_BDY_TRUNCATED) else 1
        for kind, cfg, data in self.bdy:
            if kind == "cell":
                rel, ser = data
                self.lib.ref_bdy_cell(C.byref(cfg), _ptr(rel), _ptr(ser), _ptr(self.t), _ptr(self.dt), _ptr(self.t_hydro),
                                      _ptr(target), _ptr(self.bed), _ptr(self.manning), C.c_long(rel.size))
                continue
            fn = self.lib.ref_bdy_uniform if kind == "uniform" else self.lib.ref_bdy_gridded
            fn(C.byref(cfg), _ptr(data), _ptr(self.t), _ptr(self.dt), _ptr(self.t_hydro), _ptr(target),
               _ptr(self.bed), _ptr(self.manning), C.c_int(full))

    def _reduce_advance(self, reduce_buf):
        if self.dynamic_dt:                          # CSchemeGodunov.cpp:1653-1657
            self.lib.ref_reduce(_ptr(reduce_buf), _ptr(self.bed), _ptr(self.scratch))
        self.lib.ref_advance(_ptr(self.t), _ptr(self.dt), _ptr(self.t_hydro), _ptr(self.scratch),
                             _ptr(self.primary), _ptr(self.bed), _ptr(self.t_sync), _ptr(self.batch_dt),
                             _ptr(self.ok), _ptr(self.skipped))

    def run(self, n):
        self._configure()
        trace = np.zeros(n, self.real)
        for i in range(n):
            trace[i] = self.dt[0]
            if self.scheme != MUSCL:           # CSchemeInertial inherits CSchemeGodunov::scheduleIteration
                src, dst = (self.alt, self.primary) if self.use_alt else (self.primary, self.alt)
                self._apply_boundaries(src)
                flux = self.lib.ref_gts if self.scheme == GODUNOV else self.lib.ref_ine
                flux(_ptr(self.dt), _ptr(self.bed), _ptr(src), _ptr(dst), _ptr(self.manning))
                self._reduce_advance(self.primary if (self.quirks & Q1_CFL_READS_PRIMARY) else dst)
                self.use_alt = not self.use_alt
            else:
                f = self.faces
                # the default configuration's predictor kernel, run as real 16 x 16 work-groups (shim.cpp), or the cacheNone twin
                first = self.lib.ref_mch_1st_cached if (self.quirks & Q11_MUSCL_NB_Y_IS_BED) else self.lib.ref_mch_1st
                first(_ptr(self.dt), _ptr(self.bed), _ptr(self.primary), _ptr(f[0]), _ptr(f[1]),
                                     _ptr(f[2]), _ptr(f[3]))
                self.lib.ref_mch_2nd(_ptr(self.dt), _ptr(self.primary), _ptr(self.bed), _ptr(self.manning),
                                     _ptr(f[0]), _ptr(f[1]), _ptr(f[2]), _ptr(f[3]))
                self._reduce_advance(self.primary)
        return trace

    def scalars(self):
        return dict(t=self.t[0], dt=self.dt[0], t_hydro=self.t_hydro[0], t_sync=self.t_sync[0],
                    batch_dt=self.batch_dt[0], batch_ok=int(self.ok[0]), batch_skipped=int(self.skipped[0]))

    def download(self):
        return (self.alt if (self.scheme != MUSCL and self.use_alt) else self.primary).copy()
