"""hipims_mi -- thin ctypes binding of ``libhipims_mi.so`` (the C ABI in ``include/hipims_mi.h``).

This is the Python face of the drop-in boundary used by the tests and ``bench.py``; the product is the
shared library.  There is no CPU fallback anywhere in this package: loading fails loudly when the
library is missing, and creating a domain fails when no HIP device is usable.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(HERE)
LIB_PATH = os.path.join(PKG_ROOT, "lib", "libhipims_mi.so")

# enums of include/hipims_mi.h
SCHEME_GODUNOV, SCHEME_MUSCL_HANCOCK, SCHEME_INERTIAL = 0, 1, 2
ARRAY_STATE, ARRAY_BED, ARRAY_MANNING = 0, 1, 2
QUIRK_CFL_READS_PRIMARY, QUIRK_BDY_TRUNCATED, QUIRK_MUSCL_NEIGHBOUR_Y_IS_BED, QUIRKS_REFERENCE = 1, 2, 4, 7
MATH_FAST, MATH_STRICT = 0, 1
KERNEL_AUTO, KERNEL_BASIC = 0, 1
UNIFORM_RAIN_INTENSITY, UNIFORM_LOSS_RATE = 0, 1
GRIDDED_RAIN_INTENSITY, GRIDDED_RAIN_ACCUMUL, GRIDDED_MASS_FLUX = 0, 1, 2
DEPTH_IGNORE, DEPTH_IS_FSL, DEPTH_IS_DEPTH, DEPTH_IS_CRITICAL = 0, 1, 2, 3
DISCHARGE_IGNORE, DISCHARGE_IS_DISCHARGE, DISCHARGE_IS_VELOCITY, DISCHARGE_IS_VOLUME = 0, 1, 2, 3
PTR_STATE_NEXT_SRC, PTR_STATE_OTHER, PTR_BED, PTR_MANNING, PTR_CFL_MAX, PTR_SCALARS = range(6)

EXPORTS = [
    "hp_abi_version", "hp_device_count", "hp_device_info", "hp_last_error", "hp_set_log_sink", "hp_domain_desc_default",
    "hp_domain_create", "hp_domain_destroy", "hp_domain_upload", "hp_domain_download", "hp_domain_upload_rows", "hp_state_save", "hp_state_restore",
    "hp_boundary_add_uniform", "hp_boundary_add_gridded", "hp_boundary_add_cell", "hp_boundary_clear", "hp_boundaries_fused", "hp_set_target_time", "hp_set_time",
    "hp_force_timestep", "hp_reset_counters", "hp_update_timestep", "hp_step_batch", "hp_read_scalars",
    "hp_sync", "hp_is_busy", "hp_step_begin", "hp_step_end", "hp_step_needs_reduction", "hp_device_ptr", "hp_stream", "hp_set_halo_overlap",
    "hp_stream_halo", "hp_comm_load", "hp_comm_unique_id", "hp_strip_comm_init", "hp_strip_step_batch", "hp_strip_update_timestep",
    "hp_strip_comm_destroy", "hp_strip_info", "hp_strip_peer_ticket", "hp_strip_peer_connect", "hp_strip_peer_round",
    "hp_strip_peer_disconnect", "hp_timer_start",
    "hp_timer_stop", "hp_kernel_timing", "hp_kernel_timing_read", "hp_kernel_timing_overhead",
    "hp_launch_counts", "hp_pair_stats",
]


class HipimsError(RuntimeError):
    pass


LOG_SINK = C.CFUNCTYPE(None, C.c_int, C.c_char_p, C.c_void_p)     # hp_log_sink_t
_log_sink_ref = None


def set_log_sink(callback):
    """Route every failing call's message to `callback(level, text)` (the reference's model::doError ->
    CLog::writeError place, main.cpp:631-652).  None removes the sink."""
    global _log_sink_ref
    lib = load_library()
    if callback is None:
        sink = C.cast(None, LOG_SINK)
    else:
        sink = LOG_SINK(lambda level, msg, _user: callback(int(level), msg.decode(errors="replace")))
    _check(lib, lib.hp_set_log_sink(sink, None), "hp_set_log_sink")
    _log_sink_ref = sink          # keep the trampoline alive while the library may call it


class DeviceInfo(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("arch", C.c_char * 32), ("compute_units", C.c_int32),
                ("clock_mhz", C.c_int32), ("global_mem_bytes", C.c_uint64), ("lds_bytes_per_cu", C.c_uint64),
                ("wavefront", C.c_int32), ("fp64", C.c_int32)]


class DomainDesc(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("cols", C.c_int64), ("rows", C.c_int64),
                ("dx", C.c_double), ("precision", C.c_int32), ("scheme", C.c_int32), ("courant", C.c_double),
                ("dry_threshold", C.c_double), ("friction", C.c_int32), ("dynamic_dt", C.c_int32),
                ("dt_fixed", C.c_double), ("dt_initial", C.c_double), ("t_end", C.c_double),
                ("quirks", C.c_uint32), ("math_mode", C.c_int32), ("kernel", C.c_int32),
                ("global_rows", C.c_int64), ("row_offset", C.c_int64), ("ghost_rows", C.c_int32), ("reserved0", C.c_int32)]


class StripInfo(C.Structure):
    _fields_ = [("library", C.c_char * 256), ("comm_ranks", C.c_int32), ("comm_rank", C.c_int32),
                ("halo_overlap", C.c_int32), ("ghost_rows", C.c_int32), ("peer_max", C.c_int32), ("peer_halo", C.c_int32)]


class ScalarsOut(C.Structure):
    _fields_ = [("time", C.c_double), ("timestep", C.c_double), ("time_hydrological", C.c_double),
                ("time_target", C.c_double), ("batch_timesteps", C.c_double), ("batch_successful", C.c_uint32),
                ("batch_skipped", C.c_uint32), ("cells_calculated", C.c_uint64), ("iterations", C.c_uint64)]


_lib = None


def load_library(path: str | None = None):
    """dlopen libhipims_mi.so (no GPU needed for this) and declare the signatures."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("HIPIMS_MI_LIB") or LIB_PATH       # HIPIMS_MI_LIB: kernel-variant experiments
    # One HIP runtime per process.  torch wheels bundle their own libamdhip64 (same SONAME as /opt/rocm's): if
    # torch initialises AFTER this library has pulled in the system runtime, the process ends up with two
    # runtimes and torch sees no GPU.  Loading torch first makes the dynamic loader bind this library to torch's
    # copy, so device pointers and streams can be shared with torch.distributed (RCCL).  Set
    # HIPIMS_MI_NO_TORCH=1 for a torch-free process (then the system runtime is used).
    if os.environ.get("HIPIMS_MI_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(path):
        raise HipimsError(f"{path} is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          f"(make -C hipims-ocl_amd/csrc); there is no CPU fallback")
    lib = C.CDLL(path)
    lib.hp_last_error.restype = C.c_char_p
    lib.hp_set_log_sink.argtypes = [LOG_SINK, C.c_void_p]
    lib.hp_domain_desc_default.restype = None
    lib.hp_domain_desc_default.argtypes = [C.POINTER(DomainDesc)]
    lib.hp_domain_create.argtypes = [C.POINTER(DomainDesc), C.POINTER(C.c_void_p)]
    lib.hp_domain_destroy.argtypes = [C.c_void_p]
    lib.hp_domain_upload.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    lib.hp_domain_download.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int64]
    lib.hp_domain_upload_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]
    lib.hp_state_save.argtypes = [C.c_void_p]
    lib.hp_state_restore.argtypes = [C.c_void_p]
    lib.hp_boundary_add_uniform.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_double, C.c_double]
    lib.hp_boundary_add_gridded.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64,
                                            C.c_double, C.c_double, C.c_double, C.c_double]
    lib.hp_boundary_add_cell.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                         C.c_double, C.c_double]
    lib.hp_boundary_clear.argtypes = [C.c_void_p]
    lib.hp_boundaries_fused.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    lib.hp_set_target_time.argtypes = [C.c_void_p, C.c_double]
    lib.hp_force_timestep.argtypes = [C.c_void_p, C.c_double]
    lib.hp_set_time.argtypes = [C.c_void_p, C.c_double]
    lib.hp_reset_counters.argtypes = [C.c_void_p]
    lib.hp_update_timestep.argtypes = [C.c_void_p]
    lib.hp_step_batch.argtypes = [C.c_void_p, C.c_uint32]
    lib.hp_read_scalars.argtypes = [C.c_void_p, C.POINTER(ScalarsOut)]
    lib.hp_sync.argtypes = [C.c_void_p]
    lib.hp_is_busy.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    lib.hp_step_begin.argtypes = [C.c_void_p]
    lib.hp_step_end.argtypes = [C.c_void_p]
    lib.hp_step_needs_reduction.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    lib.hp_device_ptr.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    lib.hp_stream.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    lib.hp_set_halo_overlap.argtypes = [C.c_void_p, C.c_int]
    lib.hp_stream_halo.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    lib.hp_comm_load.argtypes = [C.c_char_p]
    lib.hp_comm_unique_id.argtypes = [C.c_void_p]
    lib.hp_strip_comm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lib.hp_strip_step_batch.argtypes = [C.c_void_p, C.c_uint32]
    lib.hp_strip_update_timestep.argtypes = [C.c_void_p]
    lib.hp_strip_comm_destroy.argtypes = [C.c_void_p]
    lib.hp_strip_info.argtypes = [C.c_void_p, C.POINTER(StripInfo)]
    lib.hp_strip_peer_ticket.argtypes = [C.c_void_p, C.c_void_p]
    lib.hp_strip_peer_connect.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
    lib.hp_strip_peer_round.argtypes = [C.c_void_p, C.c_double, C.POINTER(C.c_double)]
    lib.hp_strip_peer_disconnect.argtypes = [C.c_void_p]
    lib.hp_timer_start.argtypes = [C.c_void_p]
    lib.hp_timer_stop.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    lib.hp_kernel_timing.argtypes = [C.c_void_p, C.c_int]
    lib.hp_kernel_timing_read.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
    if hasattr(lib, "hp_kernel_timing_overhead"):       # (absent from round-3 builds loaded through HIPIMS_MI_LIB for A/B runs)
        lib.hp_kernel_timing_overhead.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    if hasattr(lib, "hp_launch_counts"):                # (absent from builds before round 5, as above)
        lib.hp_launch_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.hp_device_count.argtypes = [C.POINTER(C.c_int)]
    lib.hp_device_info.argtypes = [C.c_int, C.POINTER(DeviceInfo)]
    _lib = lib
    return lib


def _check(lib, rc, what):
    if rc != 0:
        raise HipimsError(f"{what} failed ({rc}): {lib.hp_last_error().decode(errors='replace')}")


COMM_ID_BYTES = 128
PEER_TICKET_BYTES = 384


def comm_load(path: str | None = None):
    """Load the collective library for the C++ strip loop (hp_strip_*).  In a torch process the copy torch carries is
    used, so that one RCCL (bound to one HIP runtime) serves the whole process."""
    lib = load_library()
    if path is None:
        try:
            import torch
            cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
            path = cand if os.path.exists(cand) else None
        except ImportError:
            path = None
    _check(lib, lib.hp_comm_load(path.encode() if path else None), "hp_comm_load")


def comm_unique_id() -> bytes:
    lib = load_library()
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _check(lib, lib.hp_comm_unique_id(buf), "hp_comm_unique_id")
    return buf.raw


def device_count() -> int:
    lib = load_library()
    n = C.c_int(0)
    _check(lib, lib.hp_device_count(C.byref(n)), "hp_device_count")
    return n.value


def device_info(device: int = 0) -> dict:
    lib = load_library()
    info = DeviceInfo()
    _check(lib, lib.hp_device_info(device, C.byref(info)), "hp_device_info")
    return dict(name=info.name.decode(), arch=info.arch.decode(), compute_units=info.compute_units,
                clock_mhz=info.clock_mhz, global_mem_bytes=info.global_mem_bytes,
                lds_bytes_per_cu=info.lds_bytes_per_cu, wavefront=info.wavefront, fp64=bool(info.fp64))


class _DevicePointer:
    """Raw device allocation exposed through __cuda_array_interface__ so torch can wrap it without a copy."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = dict(shape=tuple(shape), typestr=typestr, data=(int(ptr), False), version=3,
                                             strides=None)


class Domain:
    """One Cartesian domain (or one row strip of it) resident on one GPU.

    Method names follow the reference's scheme/executor surface for this path:
    ``upload`` = COCLBuffer::queueWriteAll, ``download`` = readDomainAll/queueReadAll,
    ``step_batch`` = the scheduleIteration loop of Threaded_runBatch, ``read_scalars`` = readKeyStatistics,
    ``sync`` = COCLDevice::blockUntilFinished, ``set_target_time`` / ``force_timestep`` = CScheme's.
    """

    def __init__(self, cols, rows, dx=1.0, scheme=SCHEME_GODUNOV, precision="f64", dry_threshold=1e-10,
                 courant=0.5, t_end=1e30, dynamic_dt=True, dt_fixed=0.001, dt_initial=0.001, friction=True,
                 quirks=QUIRKS_REFERENCE, math_mode=MATH_FAST, kernel=KERNEL_AUTO, device=0,
                 global_rows=0, row_offset=0, ghost_rows=0):
        self.lib = load_library()
        self.cols, self.rows = int(cols), int(rows)
        self.precision = precision
        self.real = np.float64 if precision == "f64" else np.float32
        desc = DomainDesc()
        self.lib.hp_domain_desc_default(C.byref(desc))
        desc.device = device
        desc.cols, desc.rows, desc.dx = self.cols, self.rows, dx
        desc.precision = 8 if precision == "f64" else 4
        desc.scheme = scheme
        desc.courant, desc.dry_threshold = courant, dry_threshold
        desc.friction, desc.dynamic_dt = int(friction), int(dynamic_dt)
        desc.dt_fixed, desc.dt_initial, desc.t_end = dt_fixed, dt_initial, t_end
        desc.quirks, desc.math_mode, desc.kernel = quirks, math_mode, kernel
        desc.global_rows, desc.row_offset = global_rows, row_offset
        desc.ghost_rows = ghost_rows
        self.desc = desc
        self.h = C.c_void_p()
        _check(self.lib, self.lib.hp_domain_create(C.byref(desc), C.byref(self.h)), "hp_domain_create")
        self._keepalive = []
        self._bed_host = None

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.lib.hp_domain_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- transfers ----
    def upload(self, state=None, bed=None, manning=None):
        for which, arr, shape in ((ARRAY_BED, bed, (self.rows, self.cols)),
                                  (ARRAY_MANNING, manning, (self.rows, self.cols)),
                                  (ARRAY_STATE, state, (self.rows, self.cols, 4))):
            if arr is None:
                continue
            a = np.ascontiguousarray(arr, dtype=self.real)
            if a.shape != shape:
                raise ValueError(f"array shape {a.shape} != {shape}")
            if which == ARRAY_BED:
                self._bed_host = a.copy()
            self._keepalive.append(a)
            _check(self.lib, self.lib.hp_domain_upload(self.h, which, a.ctypes.data_as(C.c_void_p), a.nbytes),
                   "hp_domain_upload")
        self.sync()
        self._keepalive.clear()

    def download(self, which=ARRAY_STATE, row0=0, nrows=None):
        nrows = self.rows - row0 if nrows is None else nrows
        shape = (nrows, self.cols, 4) if which == ARRAY_STATE else (nrows, self.cols)
        out = np.empty(shape, self.real)
        _check(self.lib, self.lib.hp_domain_download(self.h, which, out.ctypes.data_as(C.c_void_p), row0, nrows),
               "hp_domain_download")
        self.sync()
        return out

    def upload_rows(self, rows_state, row0):
        a = np.ascontiguousarray(rows_state, dtype=self.real)
        _check(self.lib, self.lib.hp_domain_upload_rows(self.h, a.ctypes.data_as(C.c_void_p), row0, a.shape[0]),
               "hp_domain_upload_rows")
        self.sync()

    def state_save(self):
        """Device-side checkpoint of cell states + time-control block (saveCurrentState without the PCIe trip)."""
        _check(self.lib, self.lib.hp_state_save(self.h), "hp_state_save")

    def state_restore(self):
        _check(self.lib, self.lib.hp_state_restore(self.h), "hp_state_restore")

    # ---- boundaries ----
    def add_uniform(self, definition, series, interval, length):
        s = np.ascontiguousarray(series, dtype=self.real)
        assert s.ndim == 2 and s.shape[1] == 2
        _check(self.lib, self.lib.hp_boundary_add_uniform(self.h, definition, s.ctypes.data_as(C.c_void_p),
                                                          s.shape[0], interval, length), "hp_boundary_add_uniform")

    def add_gridded(self, definition, grids, resolution, off_x, off_y, interval):
        g = np.ascontiguousarray(grids, dtype=self.real)
        assert g.ndim == 3
        _check(self.lib, self.lib.hp_boundary_add_gridded(self.h, definition, g.ctypes.data_as(C.c_void_p), g.shape[0],
                                                          g.shape[1], g.shape[2], resolution, off_x, off_y, interval),
               "hp_boundary_add_gridded")

    def add_cell(self, depth_def, discharge_def, cells, series, interval, length):
        rel = np.ascontiguousarray(cells, dtype=np.uint64)
        ser = np.ascontiguousarray(series, dtype=self.real)
        assert ser.ndim == 2 and ser.shape[1] == 4
        _check(self.lib, self.lib.hp_boundary_add_cell(self.h, depth_def, discharge_def, rel.ctypes.data_as(C.c_void_p),
                                                       rel.size, ser.ctypes.data_as(C.c_void_p), ser.shape[0], interval,
                                                       length), "hp_boundary_add_cell")

    def clear_boundaries(self):
        _check(self.lib, self.lib.hp_boundary_clear(self.h), "hp_boundary_clear")

    def boundaries_fused(self):
        """True when rain / loss ride in the flux kernel's store epilogue instead of a pass of their own."""
        f = C.c_int(0)
        _check(self.lib, self.lib.hp_boundaries_fused(self.h, C.byref(f)), "hp_boundaries_fused")
        return bool(f.value)

    # ---- time control / stepping ----
    def set_target_time(self, t):
        _check(self.lib, self.lib.hp_set_target_time(self.h, t), "hp_set_target_time")

    set_target = set_target_time

    def set_time(self, t):
        _check(self.lib, self.lib.hp_set_time(self.h, t), "hp_set_time")

    def force_timestep(self, dt):
        _check(self.lib, self.lib.hp_force_timestep(self.h, dt), "hp_force_timestep")

    def reset_counters(self):
        _check(self.lib, self.lib.hp_reset_counters(self.h), "hp_reset_counters")

    def update_timestep(self):
        _check(self.lib, self.lib.hp_update_timestep(self.h), "hp_update_timestep")

    def step_batch(self, n):
        _check(self.lib, self.lib.hp_step_batch(self.h, int(n)), "hp_step_batch")

    def step_begin(self):
        _check(self.lib, self.lib.hp_step_begin(self.h), "hp_step_begin")

    def step_end(self):
        _check(self.lib, self.lib.hp_step_end(self.h), "hp_step_end")

    def step_needs_reduction(self):
        f = C.c_int(0)
        _check(self.lib, self.lib.hp_step_needs_reduction(self.h, C.byref(f)), "hp_step_needs_reduction")
        return bool(f.value)

    # ---- the strip loop in C++ over RCCL (hp_strip_*) ----
    def strip_comm_init(self, unique_id: bytes, rank: int, world: int):
        assert len(unique_id) == COMM_ID_BYTES
        _check(self.lib, self.lib.hp_strip_comm_init(self.h, unique_id, rank, world), "hp_strip_comm_init")

    def strip_step_batch(self, n):
        _check(self.lib, self.lib.hp_strip_step_batch(self.h, int(n)), "hp_strip_step_batch")

    def strip_update_timestep(self):
        _check(self.lib, self.lib.hp_strip_update_timestep(self.h), "hp_strip_update_timestep")

    def strip_info(self):
        info = StripInfo()
        _check(self.lib, self.lib.hp_strip_info(self.h, C.byref(info)), "hp_strip_info")
        return dict(library=info.library.decode(errors="replace"), comm_ranks=info.comm_ranks, comm_rank=info.comm_rank,
                    halo_overlap=bool(info.halo_overlap), ghost_rows=info.ghost_rows, peer_max=bool(info.peer_max),
                    peer_halo=bool(info.peer_halo))

    # ---- the maximum over the strips through peer-written mailboxes (hp_strip_peer_*) ----
    def strip_peer_ticket(self) -> bytes:
        buf = C.create_string_buffer(PEER_TICKET_BYTES)
        _check(self.lib, self.lib.hp_strip_peer_ticket(self.h, buf), "hp_strip_peer_ticket")
        return buf.raw

    def strip_peer_connect(self, tickets, rank) -> int:
        """`tickets`: every rank's ticket, in rank order.  Collective.  0: everything stays with the collective library,
        1: the maximum over the strips goes through the mailboxes, 2: and the ghost rows are written by the strips into
        each other's buffers."""
        blob = b"".join(tickets)
        assert len(blob) == PEER_TICKET_BYTES * len(tickets)
        active = C.c_int(0)
        _check(self.lib, self.lib.hp_strip_peer_connect(self.h, blob, len(tickets), int(rank), C.byref(active)), "hp_strip_peer_connect")
        return int(active.value)

    def strip_peer_round(self, value: float) -> float:
        out = C.c_double(0.0)
        _check(self.lib, self.lib.hp_strip_peer_round(self.h, float(value), C.byref(out)), "hp_strip_peer_round")
        return out.value

    def strip_peer_disconnect(self):
        _check(self.lib, self.lib.hp_strip_peer_disconnect(self.h), "hp_strip_peer_disconnect")

    def strip_comm_destroy(self):
        _check(self.lib, self.lib.hp_strip_comm_destroy(self.h), "hp_strip_comm_destroy")

    def run(self, n):
        """Run n iterations and return the timestep USED by each (one blocking read per iteration: tests only)."""
        trace = np.zeros(n, self.real)
        for i in range(n):
            trace[i] = self.read_scalars()["timestep"]
            self.step_batch(1)
        return trace

    def read_scalars(self):
        out = ScalarsOut()
        _check(self.lib, self.lib.hp_read_scalars(self.h, C.byref(out)), "hp_read_scalars")
        return {k: getattr(out, k) for k, _ in ScalarsOut._fields_}

    def sync(self):
        _check(self.lib, self.lib.hp_sync(self.h), "hp_sync")

    def is_busy(self):
        b = C.c_int(0)
        _check(self.lib, self.lib.hp_is_busy(self.h, C.byref(b)), "hp_is_busy")
        return bool(b.value)

    # ---- raw device access for the strip exchange ----
    def device_ptr(self, which):
        p = C.c_void_p()
        _check(self.lib, self.lib.hp_device_ptr(self.h, which, C.byref(p)), "hp_device_ptr")
        return p.value

    def stream_ptr(self):
        p = C.c_void_p()
        _check(self.lib, self.lib.hp_stream(self.h, C.byref(p)), "hp_stream")
        return p.value or 0

    def halo_stream_ptr(self):
        p = C.c_void_p()
        _check(self.lib, self.lib.hp_stream_halo(self.h, C.byref(p)), "hp_stream_halo")
        return p.value or 0

    def set_halo_overlap(self, on=True):
        _check(self.lib, self.lib.hp_set_halo_overlap(self.h, int(bool(on))), "hp_set_halo_overlap")

    def device_array(self, which):
        """Zero-copy view (``__cuda_array_interface__``) of a device array, for torch.as_tensor(...)."""
        typestr = "<f8" if self.precision == "f64" else "<f4"
        if which in (PTR_STATE_NEXT_SRC, PTR_STATE_OTHER):
            shape = (self.rows, self.cols, 4)
        elif which in (PTR_BED, PTR_MANNING):
            shape = (self.rows, self.cols)
        elif which == PTR_CFL_MAX:
            shape = (1,)
        else:
            raise ValueError(which)
        return _DevicePointer(self.device_ptr(which), shape, typestr)

    # ---- measurement ----
    def timer_start(self):
        _check(self.lib, self.lib.hp_timer_start(self.h), "hp_timer_start")

    def timer_stop(self):
        ms = C.c_float(0)
        _check(self.lib, self.lib.hp_timer_stop(self.h, C.byref(ms)), "hp_timer_stop")
        return ms.value

    def kernel_timing(self, stride):
        _check(self.lib, self.lib.hp_kernel_timing(self.h, stride), "hp_kernel_timing")

    def kernel_timing_read(self):
        avg, n = C.c_double(0), C.c_uint32(0)
        _check(self.lib, self.lib.hp_kernel_timing_read(self.h, C.byref(avg), C.byref(n)), "hp_kernel_timing_read")
        return avg.value, n.value

    def launch_counts(self):
        """(whole-domain flux launches queued so far, how many carried their own tail block); None with a library that
        predates the call (round-4 builds loaded through HIPIMS_MI_LIB for A/B runs)."""
        if not hasattr(self.lib, "hp_launch_counts"):
            return None
        a, b = C.c_uint64(0), C.c_uint64(0)
        _check(self.lib, self.lib.hp_launch_counts(self.h, C.byref(a), C.byref(b)), "hp_launch_counts")
        return a.value, b.value

    def pair_stats(self):
        """Diagnostics of the iteration pairs (hp_pair_stats): dict(pairs, cold_starts, stamped_last, stamped_ever); blocks."""
        if not hasattr(self.lib, "hp_pair_stats"):
            return None
        out = (C.c_uint64 * 12)()
        self.lib.hp_pair_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        _check(self.lib, self.lib.hp_pair_stats(self.h, out), "hp_pair_stats")
        return dict(pairs=out[0], cold_starts=out[1], stamped_last=out[2], stamped_ever=out[3], tune_samples=out[4],
                    tune_switches=out[5], prefers_pairs=bool(out[6]), pair_over_single=out[7] / 1000.0, stale_used=out[8])

    def kernel_timing_overhead(self):
        """Cost of an empty event pair (ms) that kernel_timing_read() has taken off every sample (0.0 with a library that
        predates the call)."""
        if not hasattr(self.lib, "hp_kernel_timing_overhead"):
            return 0.0
        ms = C.c_double(0)
        _check(self.lib, self.lib.hp_kernel_timing_overhead(self.h, C.byref(ms)), "hp_kernel_timing_overhead")
        return ms.value

    # ---- output derivation (src/Datasets/CRasterDataset.cpp:214-251) ----
    def depth_velocity(self, state=None, bed=None):
        state = self.download() if state is None else state
        bed = self._bed_host if bed is None else bed
        depth = np.maximum(0.0, state[..., 0].astype(np.float64) - bed.astype(np.float64))
        with np.errstate(divide="ignore", invalid="ignore"):
            u = np.where(depth > 1e-8, state[..., 2] / depth, 0.0)
            v = np.where(depth > 1e-8, state[..., 3] / depth, 0.0)
        return depth, u, v
