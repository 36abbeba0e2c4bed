"""Oracle-backed strip engine for the CPU (gloo) tests of hipims_mi.strips.StripRunner.

Implements the engine surface (step_begin / new_state / cfl_slot / step_end ...) with the plain-C oracle's grid
functions on numpy arrays shared with torch CPU tensors, so the decomposition protocol is exercised without a GPU.
Test infrastructure only.
"""
import ctypes as C

import numpy as np
import torch

import oracle


class OracleStripEngine:
    def __init__(self, cols, local_rows, global_rows, row_offset, scheme=oracle.GODUNOV, precision="f64",
                 quirks=oracle.QUIRKS_REFERENCE, dt_initial=0.001):
        self.sim = oracle.OracleSim(cols, local_rows, scheme=scheme, precision=precision)   # parameter holder
        self.lib, self.p, self.creal, self.real = self.sim.lib, self.sim.p, self.sim.creal, self.sim.real
        self.cols, self.rows, self.scheme, self.quirks = cols, local_rows, scheme, quirks
        g = 2 if scheme == oracle.MUSCL else 1
        self.own_lo = g if row_offset > 0 else 0
        self.own_hi = local_rows - (g if row_offset + local_rows < global_rows else 0)
        self.buf = [np.zeros((local_rows, cols, 4), self.real) for _ in range(2)]
        self.bed = np.zeros((local_rows, cols), self.real)
        self.man = np.zeros((local_rows, cols), self.real)
        self.faces = [np.zeros((local_rows * cols, 4), self.real) for _ in range(4)]
        self.sc = self.sim.Scalars(t=0, dt=dt_initial, t_hydro=0, t_sync=0, batch_dt=0, batch_ok=0, batch_skipped=0)
        self.use_alt = 0
        self.slot = np.zeros(1, self.real)
        self._written = None

    @staticmethod
    def _ptr(a):
        return a.ctypes.data_as(C.c_void_p)

    def upload(self, st, bed, man):
        for b in self.buf:
            b[...] = st
        self.bed[...] = bed
        self.man[...] = man
        self.use_alt = 0

    def set_target_time(self, t):
        self.sc.t_sync = t

    def _owned_max(self, state):
        # tst_Reduce over the rows this rank owns: reuse the flat-array reduction on a row window
        q = self.sim.Params.from_buffer_copy(self.p)
        q.rows = self.own_hi - self.own_lo
        return self.lib.orc_cfl_max_speed(C.byref(q), self._ptr(state[self.own_lo:]), self._ptr(self.bed[self.own_lo:]))

    def step_begin(self):
        P = self._ptr
        src, dst = self.buf[self.use_alt], self.buf[self.use_alt ^ 1]
        dt = self.creal(self.sc.dt)
        if self.scheme != oracle.MUSCL:
            step = self.lib.orc_godunov_step if self.scheme == oracle.GODUNOV else self.lib.orc_inertial_step
            step(C.byref(self.p), dt, P(self.bed), P(src), P(dst), P(self.man))
            reduce_buf = self.buf[0] if (self.quirks & oracle.Q1_CFL_READS_PRIMARY) else dst
        else:
            f = self.faces
            self.lib.orc_muscl_predict(C.byref(self.p), dt, P(self.bed), P(src), P(f[0]), P(f[1]), P(f[2]), P(f[3]))
            dst[...] = src
            self.lib.orc_muscl_correct(C.byref(self.p), dt, P(src), P(dst), P(self.bed), P(self.man),
                                       P(f[0]), P(f[1]), P(f[2]), P(f[3]))
            reduce_buf = dst
        self._written = dst
        self.slot[0] = self._owned_max(reduce_buf)

    def needs_reduction(self):
        return True

    def new_state(self):
        return torch.from_numpy(self._written)

    def cfl_slot(self):
        return torch.from_numpy(self.slot)

    def step_end(self):
        self.lib.orc_advance(C.byref(self.p), C.byref(self.sc), self.creal(self.slot[0]))
        self.use_alt ^= 1

    def download(self):
        return self.buf[self.use_alt].copy()

    def scalars(self):
        return dict(t=self.sc.t, dt=self.sc.dt, batch_ok=self.sc.batch_ok, batch_skipped=self.sc.batch_skipped)

    def sync(self):
        pass

    def close(self):
        pass
