#!/usr/bin/env python3
"""bench.py -- Mcell-steps/s of the Godunov + HLLC fp64 step on the synthetic flat-DEM dam-break (S-DAM).

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
N = 1 : BASELINE.json configs[1], 4096 x 4096, one MI355X.  The line also carries a `strict` leg (the exact mode: the
        reference's bits, the only mode that meets 1e-9 m on every config) and the Manning-array leg.
N > 1 : one process per GPU (torch.distributed.run), 1-D row strips with per-iteration ghost rows and an 8-byte MAX, the
        per-iteration loop run by the library itself (hp_strip_step_batch).  BASELINE.json's metric is quoted on the 4096^2
        grid at 1/2/4/8 GPUs and north_star's target is STRONG scaling, so `value` is the strong-scaling leg -- the 4096^2
        grid cut into N strips, `scaling: "strong"` -- with `speedup_vs_1gpu_same_run` from a single-domain leg that rank 0
        runs on its own GPU in the same invocation; the object `weak` is the second leg, the configs' ladder 4096^2 ->
        8192x4096 -> 8192^2 -> 16384x8192 (configs[3] at N = 8; 16,777,216 cells per GPU at every N) with its own
        roofline.  `--scaling weak|strong` runs one leg only (then `value` is that leg's).

The timed region starts with all inputs resident in HBM.  An untimed, time-based pre-warm (--prewarm-s) settles clocks
and caches; the state is then put back from a device-side checkpoint (hp_state_restore), W untimed warm-up steps run,
and EXACTLY K steps are timed (barrier + device sync on both sides).  That is done `--repeats` times -- every repeat
times the same steps [W, W+K) of the workload from t = 0 -- and the MEDIAN repeat is reported.  `--evolve-steps E`
moves the timed window to [W+E, W+E+K): a developed flood instead of the first moments after the dam has gone.
`roofline`: when an iteration is ONE launch (the flux launch carries its own tail block; no separate boundary pass) the
kernel cannot take longer than the step, and `frac` is priced from `ms_per_step` -- reproducible from the line itself and
from the driver's clock; the flux kernel is also sampled with HIP events on the domain's own stream (sparse samples, events
created before the timed region): `frac_event_sampled`, with the raw average and the cost of an empty event pair that was
taken off (`avg_launch_ms_raw`, `event_pair_overhead_ms`).  `roofline_manning_array` is the same kernel with a spatially
varying Manning array; `cpu_baseline` times the reference's kernel sources compiled for the host (oracle/_ref) / the plain-C
oracle on the host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "hipims-ocl_amd"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
BYTES_PER_CELL_STEP = {"f64": 80.0, "f32": 40.0}      # read state+bed+Manning, write state

LADDER = {1: (4096, 4096), 2: (8192, 4096), 4: (8192, 8192), 8: (16384, 8192)}


def pmc_counters(kernel_substr, stem):
    """Per-launch PMC figures of the flux kernel from the newest committed summary of THIS workload (profiles/
    rNN*_<stem>_pmc.json, produced by tools/profile_bench.sh + tools/summarize_profile.py from the same bench command: FETCH_SIZE x2 +
    WRITE_SIZE in separate passes = HBM bytes; SQ_INSTS_VALU = vector instructions issued, per wavefront)."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{stem}_*pmc.json")) + glob.glob(os.path.join(ROOT, "profiles", f"*_{stem}_pmc.json"))):
        if "developed" in os.path.basename(f) and "developed" not in stem:
            continue
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        ks = [k for name, k in d.get("kernels", {}).items() if kernel_substr in name and "hbm_bytes_per_launch" in k]
        if ks:
            k = max(ks, key=lambda e: e.get("hbm_bytes_per_launch", 0))          # (several instantiations: the one that moves the grid)
            best = {"hbm_bytes": k["hbm_bytes_per_launch"], "valu": k.get("SQ_INSTS_VALU"), "file": os.path.basename(f)}
    return best


# MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 2.4 GHz; a wave64 vector instruction issues over 2 cycles in fp32 (32 lanes per cycle; `v_fma_f32`
# 2 cyc with several waves on the SIMD) and over 4 in fp64 (half rate; measured 4.3 per fp64 wave-instruction by the round-5 profiles)
SIMDS, CLOCK_HZ, CYCLES_PER_WAVE_INSTRUCTION = 1024, 2.4e9, {"f64": 4, "f32": 2}


def attach_pmc(roof, kernel_substr, stem, precision="f64"):
    """`traffic` (measured HBM GB/s) and the two fractions that say what bounds the kernel, from the committed PMC summary of the same
    command: hbm_frac_measured = traffic / peak (what the HBM interface really sees; `frac` is ALGORITHMIC bytes per second, SURVEY
    8(d)'s 80 B per cell-step, and exceeds it where a launch covers two iterations), valu_issue_frac = the share of the launch the
    vector ALUs need for issue alone."""
    c = pmc_counters(kernel_substr, stem)
    k_ms = roof["avg_launch_ms"]
    if not c or not k_ms or k_ms <= 0:
        return
    roof["traffic"] = c["hbm_bytes"] / 1e9 / (k_ms * 1e-3)                           # GB/s, same unit as achieved
    roof["traffic_bytes_per_launch"] = c["hbm_bytes"]
    roof["traffic_source"] = "profiles/" + c["file"]
    roof["hbm_frac_measured"] = roof["traffic"] / HBM_PEAK_GBS
    if c.get("valu"):
        roof["valu_instructions_per_launch"] = c["valu"]
        roof["valu_cycles_per_wave_instruction"] = CYCLES_PER_WAVE_INSTRUCTION[precision]
        roof["valu_issue_frac"] = c["valu"] * CYCLES_PER_WAVE_INSTRUCTION[precision] / (SIMDS * CLOCK_HZ) / (k_ms * 1e-3)


def usable_cores():
    """Host threads this process may really use: the affinity mask capped by the cgroup CPU quota
    (the GPU boxes expose 256 hardware threads behind a 16-CPU quota; oversubscribing it is 10x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _time_cpu(sim, cells, budget_s):
    sim.run(4)                                   # thread pool up, first-touch done
    steps, t0 = 0, time.perf_counter()
    while True:
        sim.run(2)
        steps += 2
        el = time.perf_counter() - t0
        if el >= budget_s or steps >= 400:
            break
    return cells * steps / el / 1e6, steps, el


def cpu_baseline(cols, precision, scheme, levels=(10.0, 1.0), budget_s=8.0):
    """CPU legs on a bounded strip of the same workload, on the cores the cgroup really grants:
    kind "reference" = the reference's OWN kernel sources (gts_cacheDisabled / ine_cacheDisabled + tst_Reduce +
    tst_Advance_Normal) compiled for the host (oracle/_ref, built where /root/reference exists and shipped as .so),
    work-items spread over the cores by rows as a CPU OpenCL runtime would; kind "port" = the plain-C restatement
    (oracle/swe_oracle.c, OpenMP over rows).  The reference leg is reported when its library is present (the MUSCL
    corrector is in-place and order dependent, so that scheme only has the port leg); the other leg rides along."""
    import oracle
    from hipims_mi import synthetic as syn
    cores = usable_cores()
    rows = 512
    real = np.float64 if precision == "f64" else np.float32
    st, bed, man = syn.s_dam(cols, rows, dtype=real, levels=levels)
    legs = {}
    sim = oracle.OracleSim(cols, rows, precision=precision, scheme=scheme, threads=cores,
                           quirks=oracle.QUIRKS_REFERENCE & ~oracle.Q6_MUSCL_SERIAL)
    sim.upload(st, bed, man)
    sim.set_target(1e9)
    legs["port"] = _time_cpu(sim, cols * rows, budget_s)
    stem = {oracle.GODUNOV: "god_", oracle.INERTIAL: "ine_"}.get(scheme)
    if stem and oracle.have_ref(stem + precision):
        ref = oracle.RefSim(cols, rows, precision=precision, scheme=scheme, threads=cores)
        ref.upload(st, bed, man)
        ref.set_target(1e9)
        legs["reference"] = _time_cpu(ref, cols * rows, budget_s)
    kind = "reference" if "reference" in legs else "port"
    value, steps, el = legs[kind]
    what = ("the reference's kernel sources compiled for the host (oracle/_ref), rows over %d threads" % cores
            if kind == "reference" else "oracle/swe_oracle.c OpenMP x%d" % cores)
    out = {"value": value, "unit": "Mcell-steps/s", "cores": cores, "kind": kind,
           "sample": f"S-DAM {cols}x{rows} strip, {steps} steps, {el:.1f} s, {what}"}
    if kind == "reference":
        out["port_value"] = legs["port"][0]
    return out


def visible_gpus():
    """GPUs this process could use, counted WITHOUT the HIP runtime (the launcher must stay GPU-free: it starts the rank
    processes, and torch.cuda.device_count() falls through to hipGetDeviceCount -- which opens /dev/kfd -- when amdsmi is
    not importable).  The KFD topology lists every node; GPUs are the nodes with SIMDs.  HIP_VISIBLE_DEVICES /
    ROCR_VISIBLE_DEVICES narrow the set the ranks will see."""
    import glob
    n = 0
    for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            props = dict(l.split()[:2] for l in open(f) if len(l.split()) >= 2)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def self_launch(args):
    """`python bench.py --gpus N` without a launcher's environment: start N fresh rank processes (one per GPU) through
    torch.distributed.run BEFORE anything here has touched a GPU, relay rank 0's JSON line, exit non-zero if any rank
    failed.  Stands where the reference starts its per-device domains in-process (Domain/CDomainManager.cpp:203-220)
    and its MPI ranks through mpirun (MPI/CMPIManager.cpp:852-861).  Never degrades to fewer ranks than asked for."""
    import socket
    import subprocess
    rehearsal = os.environ.get("HIPIMS_MI_BACKEND", "nccl") != "nccl"
    have = visible_gpus()
    need = 1 if rehearsal else args.gpus
    if have < need:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible (RCCL wants one GPU per rank); "
                         f"refusing to run a smaller job under an N = {args.gpus} request")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("HIPIMS_MI_NO_TORCH", None)
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    for l in proc.stdout.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if proc.returncode != 0:
        raise SystemExit(f"bench.py --gpus {args.gpus}: a rank failed (exit code {proc.returncode})")
    if len(lines) != 1 or json.loads(lines[0]).get("n_gpus") != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: expected one JSON line for {args.gpus} ranks, got {lines!r}")
    print(lines[0], flush=True)



WORKLOAD_NAMES = {"s-rain": "S-RAIN gridded-rainfall on dry terrain", "s-rough": "S-ROUGH wet/dry rough terrain",
                  "s-dam": "S-DAM flat-DEM dam-break"}


def variant(args, **kw):
    """A copy of the command line with some fields replaced: the extra legs of the default line (other scheme, precision, window)."""
    return argparse.Namespace(**{**vars(args), **kw})


def run_leg(args, hp, cols, rows, world, rank, local_rank, math, manning_array=False, repeats=None, last=True, workload=None, force_period=None):
    """One timed leg: build this rank's runner for a `cols x rows` grid cut into `world` strips, load the workload, pre-warm,
    then `repeats` x (restore, W warm-up steps, EXACTLY K timed steps between barrier + device sync); median repeat.
    Returns what rank 0 needs for the line (every rank gets the same timing numbers: max over ranks)."""
    from hipims_mi import synthetic as syn
    workload = workload or args.workload
    scheme = {"godunov": hp.SCHEME_GODUNOV, "muscl": hp.SCHEME_MUSCL_HANCOCK, "inertial": hp.SCHEME_INERTIAL}[args.scheme]
    levels = (2.0, 1.6) if args.scheme == "inertial" else (10.0, 1.0)     # the partial-inertial scheme wants a gentle step
    math_mode = hp.MATH_FAST if math == "fast" else hp.MATH_STRICT
    kernel = hp.KERNEL_AUTO if args.kernel == "auto" else hp.KERNEL_BASIC
    real = np.float64 if args.precision == "f64" else np.float32
    dx = 2.0 if workload == "s-rain" else 1.0
    if world == 1:
        from hipims_mi.strips import SingleRunner
        runner = SingleRunner(cols, rows, dx=dx, scheme=scheme, precision=args.precision, math_mode=math_mode, kernel=kernel,
                              device=local_rank)
    else:
        from hipims_mi.strips import StripRunner
        # HIPIMS_MI_BACKEND=gloo: rehearsal of this branch with several processes on one GPU (host-staged exchange)
        backend = os.environ.get("HIPIMS_MI_BACKEND", "nccl")
        device = local_rank if backend == "nccl" else local_rank % max(1, hp.device_count())
        # default (round 6): two reaches of ghost rows wherever the strips can then run iteration PAIRS as one launch (Godunov FAST, the
        # library's own loop, no area boundaries: hipims_mi/strips.py) -- the weak-scaling shape gains 20 % on one GPU, the 4096 x 514
        # strong-scaling strip 10 % -- one reach otherwise; --exchange-period 1 / 2 forces
        runner = StripRunner(cols, rows, dx=dx, scheme=scheme, precision=args.precision, math_mode=math_mode, kernel=kernel,
                             device=device, rank=rank, world=world, backend=backend, exchange_period=(force_period or args.exchange_period or None),
                             area_boundaries=(workload == "s-rain"))
    if workload == "s-rain":
        # built in row blocks (the float64 intermediates of 8192^2 at once are 6 GB and most of the leg's wall time)
        n_local = runner.local_hi - runner.local_lo
        st, bed, man = np.empty((n_local, cols, 4), real), np.empty((n_local, cols), real), np.empty((n_local, cols), real)
        for r0 in range(runner.local_lo, runner.local_hi, 1024):
            r1 = min(r0 + 1024, runner.local_hi)
            a, b, c, rain = syn.s_rain_rows(cols, rows, r0, r1, dx=dx, dtype=real)
            st[r0 - runner.local_lo:r1 - runner.local_lo], bed[r0 - runner.local_lo:r1 - runner.local_lo] = a, b
            man[r0 - runner.local_lo:r1 - runner.local_lo] = c
        del a, b, c
        runner.upload(st, bed, man)
        runner.domain.add_gridded(hp.GRIDDED_RAIN_INTENSITY, rain["grids"], rain["resolution"], rain["off_x"],
                                  rain["off_y"], rain["interval"])
    elif workload == "s-rough":
        if world != 1:
            raise SystemExit("--workload s-rough is a single-GPU diagnostic")
        st, bed, man = syn.s_rough(cols, rows, dtype=real, manning=0.03)
        runner.upload(st, bed, man)
    else:
        st, bed, man = (syn.s_dam(cols, runner.local_rows_total, dtype=real, levels=levels) if world == 1
                        else runner.make_s_dam(real, levels=levels))
        if manning_array:     # all 80 B/cell streamed: the uniform n of S-DAM is otherwise passed as a scalar
            man = (0.03 + np.random.default_rng(11).uniform(-0.005, 0.005, man.shape)).astype(real)
        runner.upload(st, bed, man)
    del st, bed, man
    runner.set_target_time(1e9)

    # ---- untimed: a time-based pre-warm so that a short run (the driver's 20 steps are 5 ms of GPU time) is not
    #      measured on clocks and caches that have not settled.  The pre-warm advances the flood by however many steps
    #      fit into its time -- a faster kernel would hand the timed region a more developed (more expensive) flood --
    #      so the state is then put back from a device-side checkpoint: what is timed is always steps [W, W+K) of the
    #      workload from t = 0, whatever the hardware did before ----
    runner.save()
    runner.barrier()
    t_warm = time.perf_counter()
    # every rank must run the same number of passes (each is 25 collective iterations): the decision is taken on the
    # maximum over ranks of the elapsed time, which is one value everywhere
    while runner.max_over_ranks(time.perf_counter() - t_warm) < args.prewarm_s:
        runner.step(25)
        runner.barrier()

    stride = max(1, args.steps // 12) | 1              # odd: both CFL flavours of the kernel get sampled
    runs = []
    tail_carried = True                                 # every flux launch of every timed region carried its own tail block
    pair_counts = (0, 0)
    per_launch = set()                                  # iterations one flux launch covered (2: the two-iterations kernel ran pairs)
    for _ in range(max(1, repeats if repeats is not None else args.repeats)):
        runner.restore()
        runner.step(args.warmup + args.evolve_steps)
        runner.domain.kernel_timing(stride)
        runner.barrier()
        c0 = runner.domain.launch_counts()
        t0 = time.perf_counter()
        runner.step(args.steps)
        runner.barrier()
        el = runner.max_over_ranks(time.perf_counter() - t0)
        c1 = runner.domain.launch_counts()
        # asked of the library, not guessed from the command line: the engine can decline the tail block
        # (HP_TAIL_MAX_BLOCKS, split steps, a library that predates the call)
        launches = (c1[0] - c0[0]) if c0 is not None else 0
        tail_carried = tail_carried and launches > 0 and c1[1] - c0[1] == launches
        # launches < steps: iteration PAIRS ran (godunov_march2; a batch that starts on the odd iteration of a pair runs one
        # single iteration first and may end with one) -- the dominant kernel then covers two iterations per launch
        per_launch.add(2 if 0 < launches < args.steps else 1)
        pair_counts = (args.steps - launches, 2 * launches - args.steps) if 0 < launches < args.steps else (0, launches)
        k_ms, k_n = runner.domain.kernel_timing_read()
        runs.append((el, k_ms, k_n))
    overhead_ms = runner.domain.kernel_timing_overhead()
    runs.sort()
    elapsed = runs[len(runs) // 2][0]
    k_n = runs[len(runs) // 2][2]
    k_ms = sorted(r[1] for r in runs)[len(runs) // 2]
    sc = runner.domain.read_scalars()
    # ---- N > 1: a number is only a measurement if the strips really saw each other's rows.  After the timed batch every
    #      rank's ghost rows must equal, bit for bit, the rows their owners hold -- checked over torch.distributed's own
    #      channel, whatever transport the strip loop used (RCCL send/receive, or the strips' direct writes over xGMI) ----
    if world > 1:
        bad = runner.verify_ghost_rows()
        if bad and args.exchange_period == 0 and force_period is None and getattr(runner, "exchange_period", 1) == 2:
            # the automatic choice -- two reaches of ghost rows, iteration pairs on the strips, a path no multi-GPU node had run before
            # this line was written -- did not verify (the same verdict on every rank: verify_ghost_rows is collective): the leg is run
            # again on the transport every earlier probe used, and the line says so
            runner.close(destroy_group=False)
            leg = run_leg(args, hp, cols, rows, world, rank, local_rank, math, manning_array=manning_array, repeats=repeats, last=last,
                          workload=workload, force_period=1)
            leg["exchange_fallback"] = f"two reaches + iteration pairs: {bad} ghost-row cells differed from their owners' after the timed batch; re-run with one reach"
            return leg
        if bad:
            raise SystemExit(f"bench.py --gpus {world}: {bad} ghost-row cells differ from their owners' values after the timed "
                             f"batch of the {cols}x{rows} leg -- the strips did not exchange correctly; no line is printed for a broken run")
    strip_info = runner.domain.strip_info() if world > 1 and getattr(runner, "loop", "") == "cxx" else None
    leg = dict(cols=cols, rows=rows, world=world, math=math, elapsed=elapsed, runs=runs, k_ms=k_ms, k_n=k_n, sc=sc,
               exchange_period=getattr(runner, "exchange_period", 1),
               overhead_ms=overhead_ms, strip_info=strip_info, loop=getattr(runner, "loop", "batch call"),
               cells_per_launch=cols * runner.local_rows_total, flux_kernel=runner.flux_kernel_name, levels=levels,
               fused=(runner.domain.boundaries_fused() if workload == "s-rain" else None), workload=workload,
               tail_carried=tail_carried, iterations_per_launch=(max(per_launch) if per_launch else 1), pair_counts=pair_counts,
               # an iteration is ONE launch: the flux launch carried the time advance (counted by the library) and nothing else
               # is queued per iteration (S-RAIN keeps the stand-alone boundary pass's launch, which declines when fused)
               one_launch=(tail_carried and workload != "s-rain"))
    if world > 1 and not last:
        runner.close(destroy_group=False)
    else:
        runner.close()
    return leg


def roofline_of(args, leg):
    """algorithmic bytes per launch / the flux kernel's launch duration, against the 8 TB/s HBM line."""
    bpc = BYTES_PER_CELL_STEP[args.precision]
    ms_step = leg["elapsed"] / args.steps * 1e3
    raw = leg["k_ms"] + leg["overhead_ms"] if leg["k_ms"] > 0 else 0.0
    ipl = max(1, leg.get("iterations_per_launch") or 1)       # 2: a launch covers a PAIR of iterations (hp_kernels.hpp: godunov_march2)
    bytes_launch = bpc * leg["cells_per_launch"] * ipl          # SURVEY 8(d)'s per-unit figure x the cell-steps one launch processes
    frac_of = lambda ms: (bytes_launch / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms > 0 else 0.0
    ms_launch = ms_step * ipl       # (pairs: a single iteration at a batch's ends takes longer per step than a pair, so this bounds the pair launch too)
    ev_ms = min(leg["k_ms"], ms_launch) if (leg["one_launch"] and leg["k_ms"] > 0) else leg["k_ms"]   # a kernel is never longer than the iterations it is all of
    launch_ms = ms_launch if leg["one_launch"] else ev_ms
    r = {"bound": "hbm", "achieved": frac_of(launch_ms) * HBM_PEAK_GBS, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": frac_of(launch_ms), "traffic": None, "kernel": leg["flux_kernel"] + ("2 (two iterations per launch)" if ipl == 2 else ""),
         "frac_basis": (("ms_per_step: an iteration is one launch (the flux launch carries the time advance), so the kernel's "
                         "duration is at most the step's" if ipl == 1 else
                         f"ms_per_step x {ipl}: one launch covers {ipl} iterations and carries their time advances, so the kernel's duration "
                         f"is at most {ipl} steps'") if leg["one_launch"] else "HIP-event samples of the flux kernel"),
         "iterations_per_launch": ipl, "timed_region_launches": {"pairs": leg.get("pair_counts", (0, 0))[0], "single_iterations": leg.get("pair_counts", (0, 0))[1]},
         "avg_launch_ms": launch_ms, "frac_event_sampled": frac_of(ev_ms), "avg_launch_ms_event_sampled": ev_ms,
         "avg_launch_ms_raw": raw, "event_pair_overhead_ms": leg["overhead_ms"], "launches_sampled": leg["k_n"],
         "algorithmic_bytes_per_cell_step": bpc, "cells_per_launch": leg["cells_per_launch"],
         "cell_steps_per_launch": leg["cells_per_launch"] * ipl}
    r["flux_launches_carried_their_tail"] = bool(leg.get("tail_carried"))
    # What `frac` is NOT: an HBM utilisation.  It prices SURVEY 8(d)'s 80 (40) B per cell-step, the
    # contract's figure; a launch that covers TWO iterations keeps the intermediate state in registers and HAS to move only half of
    # that per cell-step, so `frac` can pass 1.0 there.  The bytes such a launch must move, priced the same way:
    r["bytes_launch_must_move"] = bpc * leg["cells_per_launch"]          # state in + bed + Manning + state out, once per launch
    r["frac_of_bytes_launch_must_move"] = r["frac"] / ipl
    r["frac_is"] = ("algorithmic bytes (SURVEY 8d: %g B per cell-step x cell-steps per launch) / launch time / 8 TB/s -- not HBM "
                    "utilisation: see hbm_frac_measured (PMC traffic / peak) and valu_issue_frac beside it" % bpc)
    if leg["workload"] in ("s-dam", "s-rain", "s-rough") and leg.get("manning_uniform", True):
        # the synthetic workloads have ONE Manning value, which the engine passes as a scalar: the bytes such a launch
        # really has to move are 72 (fp64) / 36 (fp32) per cell; SURVEY 8(d)'s contract figure stays above
        r["uniform_manning_bytes_per_cell_step"] = bpc * 0.9
        r["frac_at_uniform_manning_bytes"] = 0.9 * r["frac"]
    return r


def parallelism_of(leg):
    world, si = leg["world"], leg["strip_info"]
    s = f"row-strips x{world}"
    if world > 1:
        s += f", per-iteration loop: {leg['loop']}, {leg.get('exchange_period', 1)} iteration(s) per ghost-row exchange"
    if si:
        s += (f", collective library {si['library']} reporting {si['comm_ranks']} ranks, halo overlap {'on' if si['halo_overlap'] else 'off'}, "
              f"maximum over the strips by {'peer-written mailboxes' if si['peer_max'] else 'all-reduce'}, ghost rows "
              f"{'stored into the neighbours by the strips themselves' if si['peer_halo'] else 'sent and received through the library'}")
    if leg.get("exchange_fallback"):
        s += f" [FALLBACK: {leg['exchange_fallback']}]"
    if world > 1 and os.environ.get("HIPIMS_MI_BACKEND", "nccl") != "nccl":
        s += " (REHEARSAL: gloo, host-staged exchange, shared GPU -- not a measurement)"
    return s


def workload_of(args, leg):
    lv = leg["levels"]
    return (f"{WORKLOAD_NAMES[leg['workload']]} {leg['cols']}x{leg['rows']}{'' if lv[0] == 10.0 else ' (levels %g|%g m)' % lv}, "
            f"{args.scheme + '+HLLC' if args.scheme != 'inertial' else 'partial-inertial'}, friction fused, "
            f"dynamic CFL dt, quirks=reference, math={leg['math']}, kernel={args.kernel}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--cols", type=int, default=0, help="with --rows: the grid of the N = 1 / strong-scaling leg (the weak leg has N times the rows)")
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--scheme", choices=["godunov", "muscl", "inertial"], default="godunov")
    ap.add_argument("--precision", choices=["f64", "f32"], default="f64")
    ap.add_argument("--math", choices=["fast", "strict"], default="fast")
    ap.add_argument("--kernel", choices=["auto", "basic"], default="auto")
    ap.add_argument("--workload", choices=["s-dam", "s-rain", "s-rough"], default="s-dam",
                    help="s-dam: BASELINE configs[1..3]; s-rain: configs[4] (initially dry terrain + gridded rainfall, dx = 2 m); "
                         "s-rough: SURVEY 8(d)'s wet/dry rough terrain at full size (every tile on the general path; N = 1)")
    ap.add_argument("--scaling", choices=["both", "weak", "strong"], default="both",
                    help="N > 1: both (default) = `value` from the strong leg (the metric's 4096^2 grid cut into N strips) and the "
                         "object `weak` from the configs' ladder (16.8 Mcell per GPU); weak / strong = that leg only")
    ap.add_argument("--exchange-period", type=int, choices=[0, 1, 2], default=0,
                    help="N > 1: iterations per ghost-row exchange (2: two reaches of ghost rows, which lets Godunov FAST strips run iteration "
                         "pairs; 0 = 2 where that is possible, else 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-manning-leg", action="store_true")
    ap.add_argument("--no-strict-leg", action="store_true")
    ap.add_argument("--no-moving-leg", action="store_true", help="skip the moving-water leg (S-ROUGH) of the default line")
    ap.add_argument("--no-config-legs", action="store_true", help="skip the c3_muscl / c5_fp32_rain legs of the default line (BASELINE configs[2], [4])")
    ap.add_argument("--no-single-leg", action="store_true", help="N > 1: skip rank 0's single-domain leg (speedup_vs_1gpu_same_run)")
    ap.add_argument("--repeats", type=int, default=3, help="timed repeats of --steps steps; the median is reported")
    ap.add_argument("--prewarm-s", type=float, default=0.4, help="untimed time-based pre-warm (the state is restored afterwards)")
    ap.add_argument("--evolve-steps", type=int, default=0, help="untimed steps after the warm-up: time a developed flood")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if world == 1:
        # single GPU needs no collectives: keep torch (and its bundled, older HIP runtime) out of the process
        os.environ.setdefault("HIPIMS_MI_NO_TORCH", "1")
    import hipims_mi as hp

    custom = bool(args.cols and args.rows)
    strong_grid = (args.cols, args.rows) if custom else LADDER[1]
    weak_grid = ((args.cols, args.rows * world) if custom else LADDER.get(world, (4096, 4096 * world)))
    bpc = BYTES_PER_CELL_STEP[args.precision]

    if world == 1:
        cols, rows = strong_grid
        main_leg = run_leg(args, hp, cols, rows, 1, 0, local_rank, args.math)
        default_cfg = ((cols, rows) == (4096, 4096) and args.precision == "f64" and args.kernel == "auto"
                       and args.workload == "s-dam" and args.math == "fast")
        manning_leg = strict_leg = None
        if args.workload == "s-dam" and not args.no_manning_leg:
            manning_leg = run_leg(args, hp, cols, rows, 1, 0, local_rank, args.math, manning_array=True, repeats=1)
            manning_leg["manning_uniform"] = False
        strict_args = args
        if args.math == "fast" and not args.no_strict_leg:
            # (the exact mode chooses between iteration pairs and single iterations by measurement -- the same bits either way -- from
            # a sample of twelve iterations every 128, taken anew after a state restore: at least 24 warm-up steps keep that sample
            # out of a short timed region, where it would stand for 60 % of the steps instead of 9 %)
            strict_args = variant(args, warmup=max(args.warmup, 24))
            strict_leg = run_leg(strict_args, hp, cols, rows, 1, 0, local_rank, "strict", repeats=1)
        c3_legs, c5_leg = {}, None
        if default_cfg and not args.no_config_legs:
            # BASELINE.json configs[2] and configs[4] on the driver's own line (missing #2): the MUSCL-Hancock kernel on
            # the same grid -- the bench window and a developed flood -- and C5's shape, 8192^2 fp32 with gridded rain fused into the
            # flux kernel (at least 50 warm-up steps: the hydrological gate, one second of model time, opens inside the window)
            a3 = variant(args, scheme="muscl")
            c3_legs["window"] = (a3, run_leg(a3, hp, cols, rows, 1, 0, local_rank, "fast", repeats=1))
            a3d = variant(args, scheme="muscl", evolve_steps=1500)
            c3_legs["developed"] = (a3d, run_leg(a3d, hp, cols, rows, 1, 0, local_rank, "fast", repeats=1))
            a5 = variant(args, precision="f32", workload="s-rain", warmup=max(args.warmup, 50))
            c5_leg = (a5, run_leg(a5, hp, 8192, 8192, 1, 0, local_rank, "fast", repeats=1, workload="s-rain"))
        moving_leg = None
        if args.workload == "s-dam" and not args.no_moving_leg:
            # the headline window of S-DAM is ~98 % still water; this leg is the same grid, scheme and mode on water that moves
            # everywhere (S-ROUGH: wet/dry rough terrain, every tile live) -- VERDICT r04 item 1
            moving_leg = run_leg(args, hp, cols, rows, 1, 0, local_rank, args.math, repeats=1, workload="s-rough")
        scaling, weak_leg, single_leg = "weak", None, None
    else:
        legs = {"both": ("strong", "weak"), "weak": ("weak",), "strong": ("strong",)}[args.scaling]
        single = "strong" in legs and not args.no_single_leg
        results = {}
        for i, kind in enumerate(legs):
            cols, rows = strong_grid if kind == "strong" else weak_grid
            results[kind] = run_leg(args, hp, cols, rows, world, rank, local_rank, args.math,
                                    last=(i == len(legs) - 1 and not single))
        main_leg = results[legs[0]]
        weak_leg = results.get("weak") if legs[0] == "strong" else None
        scaling = legs[0]
        single_leg = None
        if single:
            # the strong-scaling denominator, measured in the same invocation: rank 0 runs the whole grid as ONE domain on
            # its own GPU (same protocol, one repeat less) while the other ranks wait at the barrier below
            import torch.distributed as dist
            if rank == 0:
                single_leg = run_leg(args, hp, strong_grid[0], strong_grid[1], 1, 0, local_rank, args.math, repeats=max(1, args.repeats - 1))
            dist.barrier()
            dist.destroy_process_group()
        manning_leg = strict_leg = moving_leg = c5_leg = None
        c3_legs = {}
        default_cfg = False

    if rank == 0:
        cells = main_leg["cols"] * main_leg["rows"]
        value = cells * args.steps / main_leg["elapsed"] / 1e6
        out = {
            "metric": "Mcell-steps/sec fp64 Godunov+HLLC, 4096^2 grid, 1/2/4/8 MI355X; % HBM roofline",
            "value": value, "unit": "Mcell-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": main_leg["elapsed"] / args.steps * 1e3, "higher_is_better": True, "scaling": scaling,
            "repeats_ms_per_step": [r[0] / args.steps * 1e3 for r in main_leg["runs"]],
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": workload_of(args, main_leg),
                       "cells_per_gpu": cells // world, "parallelism": parallelism_of(main_leg),
                       "area_boundaries": ("none" if main_leg["workload"] != "s-rain" else
                                           "fused into the flux kernel's store epilogue" if main_leg["fused"] else "separate pass"),
                       "timed_steps": [args.warmup + args.evolve_steps, args.warmup + args.evolve_steps + args.steps],
                       "sim_time_s": main_leg["sc"]["time"], "successful_iterations": main_leg["sc"]["batch_successful"],
                       **({"ghost_rows_verified_after_timed_batch": True} if world > 1 else {})},
            "roofline": roofline_of(args, main_leg),
        }
        if single_leg:
            v1 = single_leg["cols"] * single_leg["rows"] * args.steps / single_leg["elapsed"] / 1e6
            out["speedup_vs_1gpu_same_run"] = value / v1
            out["single_gpu_same_run"] = {"what": "the same grid as ONE domain on rank 0's GPU, same invocation, same protocol",
                                          "value": v1, "ms_per_step": single_leg["elapsed"] / args.steps * 1e3,
                                          "workload": workload_of(args, single_leg)}
        if weak_leg:
            wc = weak_leg["cols"] * weak_leg["rows"]
            out["weak"] = {"what": "weak-scaling leg: the configs' ladder, fixed cells per GPU" + (" (configs[3])" if (weak_leg["cols"], weak_leg["rows"]) == (16384, 8192) else ""),
                           "value": wc * args.steps / weak_leg["elapsed"] / 1e6, "unit": "Mcell-steps/s", "scaling": "weak",
                           "ms_per_step": weak_leg["elapsed"] / args.steps * 1e3,
                           "repeats_ms_per_step": [r[0] / args.steps * 1e3 for r in weak_leg["runs"]],
                           "workload": workload_of(args, weak_leg), "cells_per_gpu": wc // world,
                           "parallelism": parallelism_of(weak_leg), "ghost_rows_verified_after_timed_batch": True,
                           "roofline": roofline_of(args, weak_leg)}
        if manning_leg:
            rm = roofline_of(args, manning_leg)
            out["roofline_manning_array"] = {"what": "same workload with a spatially varying Manning array (all 80 B/cell streamed)",
                                             "value": cells * args.steps / manning_leg["elapsed"] / 1e6, "achieved": rm["achieved"],
                                             "frac": rm["frac"], "avg_launch_ms": rm["avg_launch_ms"],
                                             "frac_event_sampled": rm["frac_event_sampled"], "launches_sampled": rm["launches_sampled"]}
        if strict_leg:
            rs = roofline_of(strict_args, strict_leg)
            out["strict"] = {"what": "the exact mode (math=strict): bit-identical to the reference's kernels, same workload, same number of steps "
                                     "(pairs or single iterations: chosen by the engine's own measurement during the warm-up)",
                             "value": cells * args.steps / strict_leg["elapsed"] / 1e6, "unit": "Mcell-steps/s",
                             "ms_per_step": strict_leg["elapsed"] / args.steps * 1e3, "frac": rs["frac"],
                             "timed_steps": [strict_args.warmup + strict_args.evolve_steps, strict_args.warmup + strict_args.evolve_steps + strict_args.steps],
                             "iterations_per_launch": rs["iterations_per_launch"], "timed_region_launches": rs["timed_region_launches"],
                             "frac_event_sampled": rs["frac_event_sampled"], "kernel": rs["kernel"]}
        if moving_leg:
            rw = roofline_of(args, moving_leg)
            out["moving_water"] = {"what": "same grid, scheme and math mode on water that moves everywhere: " + workload_of(args, moving_leg),
                                   "value": cells * args.steps / moving_leg["elapsed"] / 1e6, "unit": "Mcell-steps/s",
                                   "ms_per_step": moving_leg["elapsed"] / args.steps * 1e3, "frac": rw["frac"],
                                   "frac_basis": rw["frac_basis"], "frac_event_sampled": rw["frac_event_sampled"],
                                   "kernel": rw["kernel"], "sim_time_s": moving_leg["sc"]["time"]}
        def extra_leg(what, a, leg, stem, kernel_substr):
            r = roofline_of(a, leg)
            attach_pmc(r, kernel_substr, stem, a.precision)
            n = leg["cols"] * leg["rows"]
            e = {"what": what + ": " + workload_of(a, leg), "value": n * a.steps / leg["elapsed"] / 1e6, "unit": "Mcell-steps/s",
                 "dtype": a.precision, "ms_per_step": leg["elapsed"] / a.steps * 1e3,
                 "timed_steps": [a.warmup + a.evolve_steps, a.warmup + a.evolve_steps + a.steps], "sim_time_s": leg["sc"]["time"]}
            for key in ("frac", "achieved", "kernel", "frac_basis", "avg_launch_ms", "frac_event_sampled", "launches_sampled", "iterations_per_launch",
                        "algorithmic_bytes_per_cell_step", "frac_of_bytes_launch_must_move", "traffic", "traffic_source", "hbm_frac_measured", "valu_issue_frac",
                        "valu_cycles_per_wave_instruction"):
                if key in r:
                    e[key] = r[key]
            if leg.get("fused") is not None:
                e["area_boundaries"] = "fused into the flux kernel's store epilogue" if leg["fused"] else "separate pass"
            return e
        if c3_legs:
            a, leg = c3_legs["window"]
            out["c3_muscl"] = extra_leg("BASELINE configs[2]: MUSCL-Hancock + MINMOD, bench window", a, leg, "muscl4096", "muscl_march<false")
            a, leg = c3_legs["developed"]
            out["c3_muscl"]["developed_flood"] = extra_leg("the same after 1500 untimed steps", a, leg, "muscl4096_developed", "muscl_march<false")
        if c5_leg:
            a, leg = c5_leg
            out["c5_fp32_rain"] = extra_leg("BASELINE configs[4]'s shape on one GPU: fp32, spatially varying rainfall", a, leg,
                                            "godunov_srain_f32_8192", "godunov_march")
        if default_cfg:
            # (the pair kernel's name when launches covered two iterations: hp::godunov_march2<...>)
            attach_pmc(out["roofline"], "godunov_march2<" if out["roofline"]["iterations_per_launch"] == 2 else args.scheme + "_march<false",
                       args.scheme + "4096", args.precision)
        if not args.no_cpu_baseline and world == 1:
            scheme = {"godunov": hp.SCHEME_GODUNOV, "muscl": hp.SCHEME_MUSCL_HANCOCK, "inertial": hp.SCHEME_INERTIAL}[args.scheme]
            out["cpu_baseline"] = cpu_baseline(main_leg["cols"], args.precision, scheme, main_leg["levels"])
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
