"""Strip decomposition on real hardware (one GPU): two/three strip domains of ONE grid live on the same device, the
StripRunner protocol is driven by hand with device-to-device copies standing in for RCCL send/recv, and the result
must be bit-identical to the single-domain run.  Also exercises the torch plumbing the multi-GPU bench relies on
(zero-copy tensor views of the engine's buffers, ExternalStream ordering, a 1-rank NCCL all-reduce)."""
import os

import numpy as np
import pytest

import hipims_mi as hp
from hipims_mi import strips, synthetic as syn

pytestmark = pytest.mark.gpu


def run_strips(cols, rows, nstrips, scheme, steps, st, bed, man, rain=None, overlap=False, **kw):
    import torch
    g = strips.ghost_rows(scheme)
    parts = strips.partition(rows, nstrips, g)
    engines = []
    for own_lo, own_hi, lo, hi in parts:
        e = strips.HipEngine(cols, hi - lo, rows, lo, scheme=scheme, **kw)
        e.upload(st[lo:hi], bed[lo:hi], man[lo:hi])
        if rain is not None:
            e.domain.add_gridded(hp.GRIDDED_RAIN_INTENSITY, *rain)
        e.set_target_time(1e9)
        e.set_halo_overlap(overlap)
        engines.append(e)
    comm = torch.cuda.Stream()                  # stands in for the collective library's own stream
    for _ in range(steps):
        for e in engines:
            e.step_begin()
        if not overlap:
            for e in engines:
                e.sync()
        # halo: what StripRunner._start_halo sends/receives, as plain device copies
        with torch.cuda.stream(comm):
            if overlap:                         # ordered after the halo segments only, like RCCL inside halo_context()
                for e in engines:
                    comm.wait_stream(e.halo_stream)
            for k in range(nstrips - 1):
                south, north = engines[k].new_state(), engines[k + 1].new_state()
                n_s = south.shape[0]
                north[0:g].copy_(south[n_s - 2 * g:n_s - g])
                south[n_s - g:n_s].copy_(north[g:2 * g])
            halo_done = comm.record_event()
            # all-reduce(MAX) of the wave speed: needs every strip's whole flux launch
            for e in engines:
                comm.wait_stream(e.stream)
            vmax = torch.stack([e.cfl_slot() for e in engines]).max()
            for e in engines:
                e.cfl_slot().fill_(vmax)
        for e in engines:
            e.stream.wait_stream(comm)
            e.stream.wait_event(halo_done)
        if not overlap:
            torch.cuda.synchronize()
        for e in engines:
            e.step_end()
    out = np.concatenate([e.download()[own_lo - lo:own_hi - lo] for e, (own_lo, own_hi, lo, hi) in zip(engines, parts)])
    sc = engines[0].scalars()
    for e in engines:
        assert e.scalars() == sc
        e.close()
    return out, sc


@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("scheme,nstrips,rows", [(hp.SCHEME_GODUNOV, 2, 96), (hp.SCHEME_GODUNOV, 3, 96),
                                                 (hp.SCHEME_GODUNOV, 3, 211), (hp.SCHEME_MUSCL_HANCOCK, 2, 96),
                                                 (hp.SCHEME_MUSCL_HANCOCK, 2, 263), (hp.SCHEME_INERTIAL, 2, 211)])
def test_strips_are_bit_identical_to_single_domain(scheme, nstrips, rows, overlap):
    # the taller grids have interior row segments, so the overlap mode really splits the launch (263: MUSCL's last
    # segment holds a single row and the halo part takes the last two segments)
    cols, steps = 200, 60
    st, bed, man = syn.s_rough(cols, rows, manning=None)
    single = hp.Domain(cols, rows, scheme=scheme)
    single.upload(st, bed, man)
    single.set_target_time(1e9)
    single.step_batch(steps)
    ref, sref = single.download(), single.read_scalars()
    out, sc = run_strips(cols, rows, nstrips, scheme, steps, st, bed, man, overlap=overlap)
    assert np.array_equal(out, ref)
    assert sc["t"] == sref["time"] and sc["dt"] == sref["timestep"]


def test_strips_with_gridded_rain_match_single_domain():
    cols, rows, steps = 120, 90, 260
    st, bed, man = syn.s_rough(cols, rows, pool_level=-10.0, amplitude=0.2, walls=False)
    st[..., 2:] = 0
    grids = np.random.default_rng(5).uniform(0, 120, (3, 6, 8))
    rain = (grids, 16.0, 0.0, 0.0, 20.0)
    single = hp.Domain(cols, rows)
    single.upload(st, bed, man)
    single.add_gridded(hp.GRIDDED_RAIN_INTENSITY, *rain)
    single.set_target_time(1e9)
    single.step_batch(steps)
    ref = single.download()
    assert (ref[..., 0] - bed).max() > 1e-5
    out, _ = run_strips(cols, rows, 2, hp.SCHEME_GODUNOV, steps, st, bed, man, rain=rain, overlap=True)
    assert np.array_equal(out, ref)


def test_single_rank_nccl_runner_matches_batch_call():
    """StripRunner over a 1-rank RCCL group: process-group init, ExternalStream, tensor views, all_reduce on the
    engine's own CFL slot -- everything the N-GPU bench does except a second rank."""
    import socket
    import torch
    import torch.distributed as dist
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    cols, rows, steps = 256, 128, 40
    st, bed, man = syn.s_dam(cols, rows)
    r = strips.StripRunner(cols, rows, rank=0, world=1, overlap=True)
    try:
        r.upload_global(st, bed, man)
        r.set_target_time(1e9)
        for _ in range(steps):                       # StripRunner.step with the collective forced on
            r.engine.step_begin()
            r._exchange_halo()
            dist.all_reduce(r.engine.cfl_slot(), op=dist.ReduceOp.MAX)
            r.engine.step_end()
        r.barrier()
        out = r.gather_owned()
        assert r.max_over_ranks(1.5) == 1.5
    finally:
        r.close()
    torch.cuda.set_stream(torch.cuda.default_stream())
    single = hp.Domain(cols, rows)
    single.upload(st, bed, man)
    single.set_target_time(1e9)
    single.step_batch(steps)
    assert np.array_equal(out, single.download())


@pytest.mark.parametrize("scheme", [hp.SCHEME_GODUNOV, hp.SCHEME_MUSCL_HANCOCK])
def test_cxx_strip_loop_single_rank_matches_batch_call(scheme):
    """The per-iteration loop run by the library itself over RCCL (hp_comm_load / hp_strip_comm_init /
    hp_strip_step_batch) with a 1-rank communicator: library loading, unique id, communicator, the all-reduce on the
    engine's CFL word in stream order -- everything but a neighbour.  Bit-identical to the plain batch call."""
    import socket
    import torch
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    cols, rows, steps = 300, 180, 120
    st, bed, man = syn.s_rough(cols, rows, manning=None)
    r = strips.StripRunner(cols, rows, scheme=scheme, rank=0, world=1, loop="cxx")
    try:
        assert r.loop == "cxx"
        r.upload_global(st, bed, man)
        r.set_target_time(3.0)                       # runs into the sync point: skipped iterations included
        r.step(steps)
        r.barrier()
        out, sc = r.gather_owned(), r.engine.scalars()
        r.set_target_time(6.0)
        r.domain.strip_update_timestep()             # resume from the sync point through the collective form
        r.step(40)
        r.barrier()
        out2, sc2 = r.gather_owned(), r.engine.scalars()
    finally:
        r.close()
    torch.cuda.set_stream(torch.cuda.default_stream())
    single = hp.Domain(cols, rows, scheme=scheme)
    single.upload(st, bed, man)
    single.set_target_time(3.0)
    single.step_batch(steps)
    ss = single.read_scalars()
    assert np.array_equal(out, single.download()) and sc["t"] == ss["time"] and sc["batch_skipped"] == ss["batch_skipped"] > 0
    single.set_target_time(6.0)
    single.update_timestep()
    single.step_batch(40)
    assert np.array_equal(out2, single.download()) and sc2["t"] == single.read_scalars()["time"]


def _torchrun(nproc, script_args, env_extra=None, timeout=600):
    import socket
    import subprocess
    import sys
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, **(env_extra or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(port)] + script_args
    return subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout)


@pytest.mark.parametrize("scheme,world", [(hp.SCHEME_GODUNOV, 2), (hp.SCHEME_GODUNOV, 3), (hp.SCHEME_MUSCL_HANCOCK, 2),
                                          (hp.SCHEME_INERTIAL, 2)])
def test_multi_process_rehearsal_on_one_gpu(scheme, world, tmp_path):
    """Several PROCESSES (torch.distributed.run, one rank each) decompose one grid with the HIP engine on the same
    GPU; the exchange goes through host memory over gloo (RCCL refuses two ranks on one device).  Everything but the
    transport is what the N-GPU run executes: bit-identical to the single domain."""
    cols, rows, steps = 200, 211, 80
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(str(tmp_path), "full.npz")
    r = _torchrun(world, [os.path.join(here, "strip_rehearsal_worker.py"), out, str(scheme), str(cols), str(rows), str(steps)])
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.load(out)
    st, bed, man = syn.s_rough(cols, rows, manning=None)
    single = hp.Domain(cols, rows, scheme=scheme)
    single.upload(st, bed, man)
    single.set_target_time(2.0)
    single.step_batch(steps)
    sc = single.read_scalars()
    assert np.array_equal(got["state"], single.download())
    assert float(got["t"]) == sc["time"] and float(got["dt"]) == sc["timestep"]
    assert int(got["skipped"]) == sc["batch_skipped"] > 0
    # bench.py's guard on N > 1 runs: nothing to report after the run, one spoiled ghost cell is reported
    assert int(got["clean"]) == 0 and int(got["spoiled"]) == 1


def test_bench_multi_rank_branch_rehearsal():
    """bench.py's N > 1 branch end to end THROUGH ITS OWN LAUNCHER (`python bench.py --gpus 2`, no torchrun around it:
    it starts the two rank processes itself), default time-based pre-warm included (its pass count must be the same on
    both ranks), strip inputs, barrier, max-over-ranks timing, rank-0 JSON line; two processes share the GPU over the
    rehearsal transport; the numbers are meaningless, the line must be well formed.  BASELINE.json's metric is quoted on
    ONE grid at 1/2/4/8 GPUs (north_star: strong scaling), so the default N > 1 line carries BOTH legs: `value` = the grid cut
    into N strips (`scaling: "strong"`, with the speed-up over the same grid as one domain measured in the same run), and
    `weak` = N times the rows (the configs' ladder) with its own roofline."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "30", "--warmup", "5",
                        "--cols", "512", "--rows", "512", "--repeats", "1"], capture_output=True, text=True, timeout=900,
                       env=dict(env, HIPIMS_MI_BACKEND="gloo"))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                     # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 30 and d["value"] > 0
    assert "REHEARSAL" in d["config"]["parallelism"] and "cpu_baseline" not in d
    # (round 6: the line says how many iterations go between ghost-row exchanges; a rehearsal -- host-staged transport, the torch loop --
    # exchanges after every one; under RCCL with the library's own loop eligible strips take two and run iteration pairs)
    assert "1 iteration(s) per ghost-row exchange" in d["config"]["parallelism"]
    # the strong leg is the line's value: the metric's grid (here 512 x 512) cut into two strips
    assert d["scaling"] == "strong" and "512x512" in d["config"]["workload"] and d["config"]["cells_per_gpu"] == 512 * 256
    assert d["config"]["successful_iterations"] == 35 and d["roofline"]["cells_per_launch"] == 512 * (256 + 1)
    assert d["config"]["ghost_rows_verified_after_timed_batch"] is True     # each strip's ghost rows == their owners' rows, bit for bit
    assert d["speedup_vs_1gpu_same_run"] > 0 and "512x512" in d["single_gpu_same_run"]["workload"]
    assert abs(d["speedup_vs_1gpu_same_run"] - d["value"] / d["single_gpu_same_run"]["value"]) < 1e-9
    # the weak leg: the ladder (N times the rows), same cells per GPU as the single domain
    w = d["weak"]
    assert w["scaling"] == "weak" and "512x1024" in w["workload"] and w["cells_per_gpu"] == 512 * 512 and w["value"] > 0
    assert w["roofline"]["cells_per_launch"] == 512 * (512 + 1) and 0 < w["roofline"]["frac"] < 1
    assert w["ghost_rows_verified_after_timed_batch"] is True
    for rl in (d["roofline"], w["roofline"]):
        assert {"frac", "frac_event_sampled", "event_pair_overhead_ms", "avg_launch_ms_raw", "frac_basis"} <= set(rl)


def test_bench_single_leg_options():
    """`--scaling weak` / `--scaling strong` run one leg only and say which."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2",
                        "--cols", "256", "--rows", "256", "--repeats", "1", "--scaling", "weak", "--prewarm-s", "0.05"],
                       capture_output=True, text=True, timeout=900, env=dict(env, HIPIMS_MI_BACKEND="gloo"))
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["scaling"] == "weak" and "256x512" in d["config"]["workload"] and "weak" not in d and "speedup_vs_1gpu_same_run" not in d


def test_bench_line_names_the_collective_library_on_the_cxx_loop():
    """One real RCCL rank through bench.py's N > 1 code path is not reachable (N = 1 takes the batch call), so the
    reporting hook is checked where it lives: the C++ strip loop with a 1-rank communicator says which library it
    loaded and how many ranks THAT LIBRARY counts."""
    import socket
    with socket.socket() as s:                      # (its own rendezvous: the test does not lean on an earlier one's environment)
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r = strips.StripRunner(64, 64, rank=0, world=1, loop="cxx")
    try:
        info = r.domain.strip_info()
    finally:
        r.close()
    assert info["library"].endswith(".so") or ".so." in info["library"]
    assert "rccl" in info["library"] and info["comm_ranks"] == 1 and info["comm_rank"] == 0


@pytest.mark.parametrize("peer_max", [2, 1, 0])
@pytest.mark.parametrize("world,scheme,precision,overlap,rain,period,cell_rank", [
    (2, hp.SCHEME_GODUNOV, "f64", 1, 0, 1, -2), (3, hp.SCHEME_GODUNOV, "f64", 0, 0, 1, -2), (4, hp.SCHEME_GODUNOV, "f32", 1, 1, 1, -2),
    (2, hp.SCHEME_MUSCL_HANCOCK, "f64", 1, 0, 1, -2), (3, hp.SCHEME_MUSCL_HANCOCK, "f64", 1, 0, 1, -2), (2, hp.SCHEME_INERTIAL, "f64", 1, 0, 1, -2),
    (3, hp.SCHEME_GODUNOV, "f64", 1, 1, 1, -2),
    # two reaches of ghost rows, one exchange per TWO iterations (round 3)
    (2, hp.SCHEME_GODUNOV, "f64", 1, 0, 2, -2), (3, hp.SCHEME_GODUNOV, "f64", 0, 0, 2, -2), (4, hp.SCHEME_GODUNOV, "f32", 1, 1, 2, -2),
    (3, hp.SCHEME_MUSCL_HANCOCK, "f64", 1, 0, 2, -2), (2, hp.SCHEME_INERTIAL, "f64", 1, 0, 2, -2), (3, hp.SCHEME_GODUNOV, "f64", 1, 1, 2, -2),
    # a cell boundary that only the strip holding its cells is told about: the other ranks must still enter every collective
    (3, hp.SCHEME_GODUNOV, "f64", 1, 0, 1, 1), (3, hp.SCHEME_GODUNOV, "f64", 1, 0, 2, 2), (2, hp.SCHEME_INERTIAL, "f64", 1, 0, 2, 0),
    # eight ranks, as on the 8-GPU node the scaling runs are made on (every rank's mailbox has seven writers)
    (8, hp.SCHEME_GODUNOV, "f32", 1, 1, 1, -2)])
def test_cxx_strip_loop_with_several_ranks(world, scheme, precision, overlap, rain, period, cell_rank, peer_max):
    """hp_strip_step_batch / hp_strip_update_timestep with 2-4 REAL ranks: the ranks are threads of one process sharing
    the GPU, the collective library is the in-process test double tests/fake_rccl (RCCL itself refuses two ranks on
    one device).  Everything on the engine's side of the nine ncclXxx entry points is the production code: which rows
    are sent and received where, on which stream, behind which events, and which iterations all-reduce.  The gathered
    strips must equal the single domain bit for bit, and every rank must report the single domain's time and dt.
    peer_max = 2: nothing of the collective library inside an iteration -- the advance kernels write the ghost rows into
    the neighbours' buffers and trade the maxima through the mailboxes (hp_strip_peer_*), waiting for each other;
    1: the mailboxes for the maximum, the (test double's) send / receive for the rows; 0: everything through the double."""
    import subprocess
    import sys
    lib = os.path.join(os.path.dirname(__file__), "fake_rccl", "libfake_rccl.so")
    if not os.path.exists(lib):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", "-o", lib,
                               os.path.join(os.path.dirname(lib), "fake_rccl.cpp")])
    res = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "strip_threads_worker.py"), str(world),
                          str(scheme), precision, str(overlap), str(rain), str(period), str(cell_rank), str(peer_max)],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "bit-identical True" in res.stdout
    assert (f"peer-written maximum [{peer_max}" in res.stdout) == bool(peer_max)


@pytest.mark.parametrize("variant,period", [("fixed", 1), ("fixed", 2), ("basic", 1), ("basic", 2), ("noq1", 1), ("noq1", 2)])
def test_cxx_strip_loop_direct_transport_variants(variant, period):
    """The strips' own transport (level 2) on the iterations' other shapes: a fixed timestep (nothing to reduce -- the
    mailbox round is then only the hand-over of the rows), the cross-check kernel (a separate reduction launch every
    iteration; with two reaches of ghost rows it used to update the inner ghost rows as well -- which the neighbours'
    advance kernels write at the same time: found by tools/strip_fuzz.py, the kernel now keeps to the launch's rows) and
    quirk Q1 off (a new maximum on EVERY iteration)."""
    import subprocess
    import sys
    res = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "strip_threads_worker.py"), "3", str(hp.SCHEME_GODUNOV), "f64",
                          "1", "0", str(period), "-2", "2", variant], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "bit-identical True" in res.stdout and "peer-written maximum [2, 2, 2]" in res.stdout


def test_strip_fuzz_a_few_seeds():
    """tools/strip_fuzz.py: random world / scheme / precision / exchange period / rain / overlap / transport level / variant /
    grid shape, each bit for bit against the single domain.  20 seeds here; 600 were run as a soak (profiles/r03ag_strip_fuzz.txt)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "strip_fuzz.py"), "1000", "20"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "failed: 0 of 20" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    # the seeds that found something: 169 / 184 (cross-check kernel writing a strip's inner ghost rows), 2175 / 2188 (FAST: a
    # strip's remembered maximum priced by the stand-alone reduction in IEEE arithmetic, the single domain's by the flux
    # kernel's epilogue in FAST arithmetic -- one ulp apart; both now price in the domain's own arithmetic)
    for seed in (169, 184, 2175, 2188):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "strip_fuzz.py"), str(seed), "1"], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "failed: 0 of 1" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("world", [2, 4])
def test_peer_mailboxes_between_processes(world, tmp_path):
    """The peer-written maximum between PROCESSES: every rank maps the other ranks' mailboxes through IPC handles
    (hipIpcGetMemHandle / hipIpcOpenMemHandle on uncached device memory), passes the library's connection test, and
    200 reductions of known values come out right on every rank.  The ranks share the box's one GPU -- what this
    cannot show is the xGMI hop between two GPUs; what it does show is that a store by another process's kernel is
    seen by this process's polling kernel, in both mailbox sets, with the bounded wait never expiring."""
    import subprocess
    import sys
    env = dict(os.environ, HP_PEER_TEST_MS="20000")
    procs = [subprocess.Popen([sys.executable, os.path.join(os.path.dirname(__file__), "peer_ipc_worker.py"), str(r), str(world), str(tmp_path)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=300)[0])
        except subprocess.TimeoutExpired:
            p.kill()
            outs.append("TIMEOUT " + p.communicate()[0])
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert all("wrong maxima 0" in o for o in outs), "\n".join(outs)
    print("\n".join(o.strip().splitlines()[-1] for o in outs))


def test_peer_mailboxes_report_a_missing_rank_instead_of_hanging():
    """A rank that never shows up: the connection test's bounded wait expires, the call says "not active" (and why, through
    the log sink) and returns -- the GPU is not left spinning."""
    import time
    dom = hp.Domain(64, 64)
    try:
        mine = dom.strip_peer_ticket()
        ghost = bytearray(mine)                       # "rank 1": a ticket for a mailbox nobody will ever write from -- this one's own
        os.environ["HP_PEER_TEST_MS"] = "300"
        t0 = time.perf_counter()
        active = dom.strip_peer_connect([mine, bytes(ghost)], 0)
        assert not active and time.perf_counter() - t0 < 5.0
        with pytest.raises(hp.HipimsError):
            dom.strip_peer_round(1.0)                 # nothing is connected any more
    finally:
        os.environ.pop("HP_PEER_TEST_MS", None)
        dom.close()


@pytest.mark.parametrize("world,scheme,precision,rain,period", [
    (2, hp.SCHEME_GODUNOV, "f64", 0, 1), (3, hp.SCHEME_GODUNOV, "f32", 1, 1), (3, hp.SCHEME_MUSCL_HANCOCK, "f64", 0, 1), (2, hp.SCHEME_GODUNOV, "f64", 1, 2),
    (8, hp.SCHEME_GODUNOV, "f64", 1, 1),
    # round 5: iteration PAIRS on the ranks (two reaches of ghost rows, no rain; period -2 = period 2 with HP_TWO_STEP=1 in the ranks'
    # environment): the pair launch stores its edge rows into the other PROCESSES' buffers, every rank swaps its IPC-mapped views
    (3, hp.SCHEME_GODUNOV, "f64", 0, -2), (4, hp.SCHEME_GODUNOV, "f32", 0, -2)])
def test_cxx_strip_loop_with_ranks_that_are_processes(world, scheme, precision, rain, period, tmp_path):
    """The strip loop as it runs in production -- one PROCESS per rank -- on the one GPU of the box: every rank maps its
    neighbours' state buffers and all ranks' mailboxes through IPC handles, the advance kernels write ghost rows and maxima
    into the other processes' memory, and nothing of the collective library is called inside an iteration (the tests' double
    serves the start-of-batch handshake through a shared file; it has no send / receive in this mode, so a fall-back to the
    library's transport would fail the run).  The gathered strips equal the single domain bit for bit; every rank reports its
    time and dt."""
    import subprocess
    import sys
    from hipims_mi import synthetic as syn
    lib = os.path.join(os.path.dirname(__file__), "fake_rccl", "libfake_rccl.so")
    if not os.path.exists(lib):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", "-o", lib,
                               os.path.join(os.path.dirname(lib), "fake_rccl.cpp")])
    shm = tmp_path / "allreduce.shm"
    shm.write_bytes(bytes(4096))
    env = dict(os.environ, FAKE_RCCL_SHM=str(shm), HP_PEER_TEST_MS="20000")
    pairs = period < 0
    if pairs:
        period, env = 2, dict(env, HP_TWO_STEP="1")
    procs = [subprocess.Popen([sys.executable, os.path.join(os.path.dirname(__file__), "strip_procs_worker.py"), str(r), str(world), str(tmp_path),
                               str(scheme), precision, str(rain), str(period)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
             for r in range(world)]
    # the single domain, meanwhile, in this process
    cols, rows, steps = 300, 157, 90
    real = np.float64 if precision == "f64" else np.float32
    if rain:
        st, bed, man, rn = syn.s_rain_rows(cols, rows, 0, rows, dx=2.0, dtype=real)
        dx = 2.0
    else:
        st, bed, man = syn.s_rough(cols, rows, dtype=real)
        rn, dx = None, 1.0
    single = hp.Domain(cols, rows, dx=dx, scheme=scheme, precision=precision)
    single.upload(st, bed, man)
    if rn is not None:
        single.add_gridded(hp.GRIDDED_RAIN_INTENSITY, rn["grids"], rn["resolution"], rn["off_x"], rn["off_y"], rn["interval"])
    single.set_target_time(1e9)
    single.update_timestep()
    single.step_batch(steps)
    want, want_sc = single.download(), single.read_scalars()
    single.close()
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=300)[0])
        except subprocess.TimeoutExpired:
            p.kill()
            outs.append("TIMEOUT " + p.communicate()[0])
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    got = np.concatenate([np.load(tmp_path / f"owned.{r}.npy") for r in range(world)], axis=0)
    assert np.array_equal(got.view(np.uint8), want.view(np.uint8))
    stamp = "time %.17g dt %.17g" % (want_sc["time"], want_sc["timestep"])
    assert all(stamp in o for o in outs), (stamp, outs)
    if pairs:                                                  # ... and the ranks really paired up
        for o in outs:
            line = [l for l in o.splitlines() if "flux launches" in l][0]
            assert int(line.split("flux launches ")[1].split()[0]) < 0.7 * int(line.split("iterations ")[1]), line


@pytest.mark.parametrize("world,precision,variant,batches", [(2, "f64", "", "1,2,87"), (3, "f64", "", "40,7,64"), (4, "f32", "", "5,2,90"),
                                                              (3, "f64", "fixed", "1,2,60"), (8, "f64", "", "16,33,40")])
def test_strips_run_iteration_pairs(world, precision, variant, batches):
    """Round 5: row strips with two reaches of ghost rows and the strips' own transport run PAIRS of Godunov iterations as one
    launch (godunov_march2 with the edge rows stored into the neighbours and ONE mailbox round per pair; HP_TWO_STEP=1 forces it
    on these small grids).  The single domain beside them runs pairs too; the gathered strips must equal it bit for bit, time and
    timestep included, over batches that start on either iteration of a pair -- and the strips must really have paired up."""
    import subprocess
    import sys
    lib = os.path.join(os.path.dirname(__file__), "fake_rccl", "libfake_rccl.so")
    if not os.path.exists(lib):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", "-o", lib,
                               os.path.join(os.path.dirname(lib), "fake_rccl.cpp")])
    args = [str(world), str(hp.SCHEME_GODUNOV), precision, "0", "0", "2", "-2", "2"] + ([variant] if variant else [])
    res = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "strip_threads_worker.py")] + args,
                         capture_output=True, text=True, timeout=600, env=dict(os.environ, HP_TWO_STEP="1", STRIP_WORKER_BATCHES=batches))
    assert res.returncode == 0, res.stdout + res.stderr
    assert "bit-identical True" in res.stdout
    line = [l for l in res.stdout.splitlines() if l.startswith("flux launches per rank")][0]
    counts = eval(line.split("flux launches per rank ")[1].split(" iterations")[0])
    iterations = int(line.split("iterations ")[1])
    assert all(c < 0.7 * iterations for c in counts), line                    # pairs, not single iterations
