"""1-D row-strip decomposition of ONE logical grid over the GPUs of a node (one process per GPU).

Replaces the reference's overlapping-domain links (src/Domain/Links/CDomainLink.cpp:286-382: strips staged through
host memory, optionally MPI, several iterations between exchanges, forecast/rollback) with the `timestep` sync
semantic only (src/CModel.cpp:650-694): every iteration

    step_begin  : boundaries + flux kernel + LOCAL max wave speed          (device, domain stream)
    halo        : `g` ghost rows of the NEW state <-> the two strip neighbours   (RCCL send/recv over xGMI)
    all-reduce  : MAX of one scalar (the wave speed; replaces MPI_Allreduce(MIN dt), CMPIManager.cpp:852-861)
    step_end    : tst_Advance_Normal on every rank redundantly -> identical dt everywhere, no host round trip

so a decomposed run is bit-identical to the single-GPU run (max is exact and order independent).
g = 1 row for Godunov, 2 for MUSCL-Hancock (13-point stencil).

Overlap: with more than one rank the engine computes the row segments next to the ghost rows on a second stream
(hp_set_halo_overlap); the halo transfer is ordered after that stream only, so it travels while the interior
segments are still being computed on the domain's stream, which waits for it just before step_end.

The exchange is written against torch.distributed so that the very same code runs over RCCL ("nccl") on GPUs and
over "gloo" on CPU tensors in the world_size-2 tests (tests/test_strips_gloo.py, with the oracle as engine).
Rehearsal transport: `backend="gloo"` together with the HIP engine stages the ghost rows and the scalar through
host memory.  It exists so that the whole multi-rank path (partition, strip inputs, bench.py's N > 1 branch) can be
run with several processes on ONE GPU, where RCCL refuses to put two ranks; it is not a production path.
"""
from __future__ import annotations

import contextlib
import os

import numpy as np

from . import (KERNEL_AUTO, MATH_FAST, PTR_CFL_MAX, PTR_STATE_NEXT_SRC, PTR_STATE_OTHER, QUIRKS_REFERENCE,
               SCHEME_GODUNOV, SCHEME_INERTIAL, SCHEME_MUSCL_HANCOCK, Domain)
from . import synthetic as syn


FLUX_KERNELS = {SCHEME_GODUNOV: "hp::godunov_march", SCHEME_MUSCL_HANCOCK: "hp::muscl_march", SCHEME_INERTIAL: "hp::inertial_march"}


def ghost_rows(scheme) -> int:
    return 2 if scheme == SCHEME_MUSCL_HANCOCK else 1


def partition(global_rows: int, world: int, g: int):
    """Rows owned by each rank and the rows it stores: [(own_lo, own_hi, local_lo, local_hi)] (global indices).
    `g` = ghost rows stored per interior side (the stencil reach, or a multiple of it for several iterations per exchange)."""
    parts = []
    for k in range(world):
        own_lo, own_hi = (k * global_rows) // world, ((k + 1) * global_rows) // world
        parts.append((own_lo, own_hi, max(0, own_lo - g), min(global_rows, own_hi + g)))
    for own_lo, own_hi, _, _ in parts:
        if own_hi - own_lo < 2 * g + 1:
            raise ValueError("strip thinner than its halo")
    return parts


class HipEngine:
    """The HIP domain of one strip + zero-copy torch views of its device buffers."""

    def __init__(self, cols, local_rows, global_rows, row_offset, **kw):
        import torch
        self.flux_kernel_name = FLUX_KERNELS[kw.get("scheme", SCHEME_GODUNOV)]
        self.torch = torch
        self.domain = Domain(cols, local_rows, global_rows=global_rows, row_offset=row_offset, **kw)
        self.device = torch.device("cuda", kw.get("device", 0))
        torch.cuda.set_device(self.device)
        # run torch's collectives in order with the engine's kernels: make the domain's stream torch's current one
        self.stream = torch.cuda.ExternalStream(self.domain.stream_ptr(), device=self.device)
        self.halo_stream = torch.cuda.ExternalStream(self.domain.halo_stream_ptr(), device=self.device)
        torch.cuda.set_stream(self.stream)
        self._views = {}

    def set_halo_overlap(self, on):
        self.domain.set_halo_overlap(on)
        self._overlap = bool(on)

    def halo_context(self):
        """Stream context the halo send/recv is issued in: the collective library orders itself after the work of
        torch's current stream, which inside this context is the stream the halo segments were computed on."""
        if getattr(self, "_overlap", False):
            return self.torch.cuda.stream(self.halo_stream)
        return contextlib.nullcontext()

    def _view(self, which):
        ptr = self.domain.device_ptr(which)
        key = (which, ptr)
        if key not in self._views:
            self._views[key] = self.torch.as_tensor(self.domain.device_array(which), device=self.device)
        return self._views[key]

    def step_begin(self):
        self.domain.step_begin()

    def step_end(self):
        self.domain.step_end()

    def needs_reduction(self):
        return self.domain.step_needs_reduction()

    def new_state(self):
        """Buffer the iteration in flight has just written (valid between step_begin and step_end)."""
        return self._view(PTR_STATE_OTHER)

    def cfl_slot(self):
        return self._view(PTR_CFL_MAX)

    def upload(self, st, bed, man):
        self.domain.upload(st, bed, man)

    def download(self):
        return self.domain.download()

    def set_target_time(self, t):
        self.domain.set_target_time(t)

    def scalars(self):
        s = self.domain.read_scalars()
        return dict(t=s["time"], dt=s["timestep"], batch_ok=s["batch_successful"], batch_skipped=s["batch_skipped"])

    def sync(self):
        self.domain.sync()

    def close(self):
        self.domain.close()


class SingleRunner:
    """N = 1: the plain batch call, no torch involved."""

    def __init__(self, cols, rows, **kw):
        self.domain = Domain(cols, rows, **kw)
        self.local_rows_total = rows
        self.local_lo, self.local_hi = 0, rows
        self.flux_kernel_name = FLUX_KERNELS[kw.get("scheme", SCHEME_GODUNOV)]

    def upload(self, st, bed, man):
        self.domain.upload(st, bed, man)

    def set_target_time(self, t):
        self.domain.set_target_time(t)

    def step(self, n):
        self.domain.step_batch(n)

    def save(self):
        self.domain.state_save()

    def restore(self):
        self.domain.state_restore()

    def barrier(self):
        self.domain.sync()

    def max_over_ranks(self, x):
        return x

    def close(self):
        self.domain.close()


class StripRunner:
    """One rank of the strip-decomposed run.  `engine_factory(cols, local_rows, global_rows, row_offset)` lets the
    CPU tests substitute an oracle-backed engine; the default builds the HIP engine."""

    def __init__(self, cols, rows, scheme=SCHEME_GODUNOV, precision="f64", rank=0, world=1, device=0,
                 engine_factory=None, backend=None, init_process_group=True, overlap=None, loop=None, exchange_period=None,
                 area_boundaries=False, **kw):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.cols, self.rows, self.rank, self.world = cols, rows, rank, world
        # ghost rows stored per interior side: one stencil reach, exchanged after every iteration -- or two, exchanged after
        # every second one by the library's own strip loop (the strip recomputes a reach of its neighbour's rows in between).
        # None = the best the configuration allows (round 6): two reaches wherever the strips can then run iteration PAIRS -- the HIP
        # engine under the library's own loop (one GPU per rank), Godunov scheme, FAST arithmetic, the tuned kernel, no area boundaries
        # to come (`area_boundaries`: the caller's word) -- whether every rank really can is settled by the ranks' handshake at the
        # start of each batch (hp_strip_step_batch); everything else exchanges after every iteration, as an explicit 1 does.
        if exchange_period is None:
            from . import KERNEL_AUTO, MATH_FAST
            want_loop = loop or os.environ.get("HIPIMS_MI_STRIP_LOOP", "cxx" if (engine_factory is None and (backend or "nccl") == "nccl") else "torch")
            pairs = (engine_factory is None and (backend or "nccl") == "nccl" and want_loop == "cxx" and world > 1 and scheme == SCHEME_GODUNOV
                     and kw.get("math_mode", MATH_FAST) == MATH_FAST and kw.get("kernel", KERNEL_AUTO) == KERNEL_AUTO and not area_boundaries
                     and os.environ.get("HP_TWO_STEP", "") != "0")
            if pairs:
                # (two reaches belong to the library's own loop: without the collective library -- every rank finds that out the same
                # way -- the run falls back to the torch loop below, which exchanges after every iteration)
                from . import HipimsError, comm_load
                try:
                    comm_load()
                except HipimsError:
                    pairs = False
            exchange_period = 2 if pairs else 1
        if exchange_period not in (1, 2):
            raise ValueError("exchange_period must be 1 or 2")
        self.exchange_period = exchange_period
        self.g = ghost_rows(scheme) * exchange_period
        self.scheme, self.precision = scheme, precision
        self.parts = partition(rows, world, self.g)
        self.own_lo, self.own_hi, self.local_lo, self.local_hi = self.parts[rank]
        self.local_rows_total = self.local_hi - self.local_lo
        if engine_factory is None:
            self.engine = HipEngine(cols, self.local_rows_total, rows, self.local_lo, scheme=scheme,
                                    precision=precision, device=device, ghost_rows=self.g if exchange_period > 1 else 0, **kw)
            backend = backend or "nccl"
        else:
            self.engine = engine_factory(cols, self.local_rows_total, rows, self.local_lo)
            backend = backend or "gloo"
        self.domain = getattr(self.engine, "domain", None)
        self.flux_kernel_name = getattr(self.engine, "flux_kernel_name", "")
        if init_process_group and not dist.is_initialized():
            kwargs = {}
            if backend == "nccl":
                kwargs["device_id"] = torch.device("cuda", device)
            dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
        # who runs the per-iteration loop: "cxx" = the library itself over RCCL (hp_strip_step_batch: a handful of
        # enqueues per iteration, no Python in the loop), "torch" = this class over torch.distributed.  The C++ loop
        # needs RCCL, i.e. one GPU per rank; rehearsals on a shared GPU (gloo) and the CPU tests use the torch loop.
        if loop is None:
            loop = os.environ.get("HIPIMS_MI_STRIP_LOOP", "cxx" if (engine_factory is None and backend == "nccl") else "torch")
        self.loop = loop
        self.peer_max = 0          # what the ranks agreed on (C++ loop): 0 library, 1 mailboxes, 2 mailboxes + pushed ghost rows
        if exchange_period > 1 and loop != "cxx":
            raise ValueError("several iterations per exchange are the C++ strip loop's (loop='cxx')")
        if loop == "cxx":
            if engine_factory is not None or backend != "nccl":
                raise ValueError("the C++ strip loop runs on the HIP engine over RCCL only")
            from . import HipimsError, comm_load, comm_unique_id
            try:
                comm_load()
            except HipimsError as e:
                # every rank loads the same library the same way, so they all land here together: the run goes on with
                # the torch loop (slower host side, same results) and says so in `loop`
                if "HIPIMS_MI_STRIP_LOOP" in os.environ:
                    raise
                import sys
                if exchange_period > 1:
                    raise
                print(f"[hipims_mi] C++ strip loop unavailable ({e}); using the torch loop", file=sys.stderr, flush=True)
                self.loop = loop = "torch"
        if loop == "cxx":
            from . import comm_unique_id
            box = [comm_unique_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(box, src=0)        # the id travels by the host's own means (here: torch)
            self.domain.strip_comm_init(box[0], rank, world)
            # the maximum over the strips: peer-written mailboxes where every rank can reach every other one's (the
            # library tests that and the ranks agree), the collective library's all-reduce otherwise
            if world > 1 and os.environ.get("HIPIMS_MI_PEER_MAX", "1") != "0":
                tickets = [None] * world
                dist.all_gather_object(tickets, self.domain.strip_peer_ticket())
                self.peer_max = self.domain.strip_peer_connect(tickets, rank)
        self.south = rank - 1 if rank > 0 else None
        self.north = rank + 1 if rank < world - 1 else None
        self.staged = engine_factory is None and backend == "gloo"      # device buffers, host transport (rehearsal)
        self._halo_ops = {}
        if overlap is None:
            overlap = world > 1
        if hasattr(self.engine, "set_halo_overlap"):
            self.engine.set_halo_overlap(overlap)

    # ---- data placement ----
    def local_slice(self):
        return slice(self.local_lo, self.local_hi)

    def make_s_dam(self, real, levels=(10.0, 1.0)):
        """This rank's rows of the global S-DAM input (built strip by strip: a 16384 x 8192 fp64 state is 4 GiB)."""
        st, bed, man = syn.s_dam(self.cols, self.local_rows_total, dtype=real, levels=levels)
        # s_dam walls all four edges of what it builds; only the global south/north rows are walls
        z_col = st[1, :, 0].copy() if self.local_rows_total > 2 else None
        if self.local_lo > 0:
            st[0, :, 0] = z_col; st[0, :, 1] = z_col; bed[0, :] = 0.0
            st[0, 0, :] = 0; st[0, -1, :] = 0; bed[0, 0] = bed[0, -1] = syn.WALL_BED
        if self.local_hi < self.rows:
            st[-1, :, 0] = z_col; st[-1, :, 1] = z_col; bed[-1, :] = 0.0
            st[-1, 0, :] = 0; st[-1, -1, :] = 0; bed[-1, 0] = bed[-1, -1] = syn.WALL_BED
        return st, bed, man

    def upload(self, st_local, bed_local, man_local):
        self.engine.upload(st_local, bed_local, man_local)

    def upload_global(self, st, bed, man):
        sl = self.local_slice()
        self.engine.upload(st[sl], bed[sl], man[sl])

    def set_target_time(self, t):
        self.engine.set_target_time(t)

    # ---- the per-iteration protocol ----
    def _start_halo(self):
        """Queue the ghost-row exchange of the iteration in flight; returns the requests to wait for."""
        dist, g = self.dist, self.g
        new = self.engine.new_state()            # [local_rows, cols, 4]
        if self.staged:
            return self._start_halo_staged(new)
        ops = self._halo_ops.get(new.data_ptr())  # the two ping-pong buffers alternate: build each list once
        if ops is None:
            n = new.shape[0]
            ops = []
            if self.south is not None:
                ops.append(dist.P2POp(dist.isend, new[g:2 * g], self.south))          # my first owned rows
                ops.append(dist.P2POp(dist.irecv, new[0:g], self.south))              # into my south ghost rows
            if self.north is not None:
                ops.append(dist.P2POp(dist.isend, new[n - 2 * g:n - g], self.north))  # my last owned rows
                ops.append(dist.P2POp(dist.irecv, new[n - g:n], self.north))          # into my north ghost rows
            self._halo_ops[new.data_ptr()] = ops
        if not ops:
            return []
        ctx = self.engine.halo_context() if hasattr(self.engine, "halo_context") else contextlib.nullcontext()
        with ctx:
            return dist.batch_isend_irecv(ops)

    def _start_halo_staged(self, new):
        """Rehearsal transport: rows go device -> host -> gloo -> host -> device."""
        dist, g, n = self.dist, self.g, new.shape[0]
        ops, landings = [], []
        with self.engine.halo_context():                      # .cpu() waits for the halo segments only
            for peer, send, recv in ((self.south, new[g:2 * g], new[0:g]), (self.north, new[n - 2 * g:n - g], new[n - g:n])):
                if peer is None:
                    continue
                buf = self.torch.empty(recv.shape, dtype=recv.dtype)
                ops += [dist.P2POp(dist.isend, send.cpu(), peer), dist.P2POp(dist.irecv, buf, peer)]
                landings.append((recv, buf))
        reqs = dist.batch_isend_irecv(ops) if ops else []

        class _Landing:
            def wait(_self):
                for r in reqs:
                    r.wait()
                for recv, buf in landings:
                    recv.copy_(buf)                           # on the domain stream, before step_end
        return [_Landing()] if ops else []

    def _all_reduce_max(self, slot):
        if self.staged:
            host = slot.cpu()                                 # waits for the whole flux launch
            self.dist.all_reduce(host, op=self.dist.ReduceOp.MAX)
            slot.copy_(host)
        else:
            self.dist.all_reduce(slot, op=self.dist.ReduceOp.MAX)

    def _exchange_halo(self):
        for req in self._start_halo():
            req.wait()

    def step(self, n):
        if self.loop == "cxx":
            self.domain.strip_step_batch(n)
            return
        dist = self.dist
        for _ in range(n):
            self.engine.step_begin()
            halo = self._start_halo()                  # travels while the interior segments are computed
            # the scalar is new only when the reduction priced a buffer that changed (every iteration without
            # quirk Q1, every other one with it); the decision is identical on all ranks (same iteration parity)
            if self.world > 1 and self.engine.needs_reduction():
                self._all_reduce_max(self.engine.cfl_slot())
            for req in halo:
                req.wait()                             # domain stream (or the host, with gloo) waits for the rows
            self.engine.step_end()

    def save(self):
        """Device-side checkpoint of this rank's strip (ghost rows included: they are consistent at iteration ends)."""
        self.engine.domain.state_save()

    def restore(self):
        self.engine.domain.state_restore()

    def barrier(self):
        self.engine.sync()
        if self.world > 1:
            self.dist.barrier()
        self.engine.sync()

    def max_over_ranks(self, x):
        if self.world == 1:
            return x
        dev = "cpu" if self.staged else getattr(self.engine, "device", "cpu")
        t = self.torch.tensor([x], dtype=self.torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def verify_ghost_rows(self):
        """After a batch: do this strip's ghost rows hold, bit for bit, what their owners hold?  (The exchange follows the
        last iteration of a batch whenever one reach of ghost rows is stored.)  Every rank sends the edge rows it owns to
        the neighbours by torch.distributed -- a channel that has nothing to do with the transport under test -- and
        compares.  Returns the largest number of differing cells any rank found (0 = verified); collective."""
        g, n = self.g, self.local_rows_total
        dom = self.engine.domain
        edges = {}
        if self.south is not None:
            edges[self.south] = dom.download(row0=g, nrows=g)                   # my first owned rows = the south neighbour's north ghost rows
        if self.north is not None:
            edges[self.north] = dom.download(row0=n - 2 * g, nrows=g)
        everyone = [None] * self.world
        self.dist.all_gather_object(everyone, edges)
        bad = 0
        if self.south is not None:
            bad += int((dom.download(row0=0, nrows=g).view(np.uint8) != everyone[self.south][self.rank].view(np.uint8)).reshape(-1, 4 * dom.real().itemsize).any(axis=1).sum())
        if self.north is not None:
            bad += int((dom.download(row0=n - g, nrows=g).view(np.uint8) != everyone[self.north][self.rank].view(np.uint8)).reshape(-1, 4 * dom.real().itemsize).any(axis=1).sum())
        return int(self.max_over_ranks(float(bad)))

    def gather_owned(self):
        """All ranks' owned rows assembled on every rank (tests)."""
        local = self.engine.download()
        mine = np.ascontiguousarray(local[self.own_lo - self.local_lo:self.own_hi - self.local_lo])
        out = [None] * self.world
        self.dist.all_gather_object(out, mine)
        return np.concatenate(out, axis=0)

    def close(self, destroy_group=True):
        """`destroy_group=False` keeps the process group for another StripRunner of this process (bench.py's second leg)."""
        if self.loop == "cxx":
            self.domain.strip_comm_destroy()
        self.engine.close()
        if destroy_group and self.dist.is_initialized():
            self.dist.destroy_process_group()
