#!/usr/bin/env python3
"""bench.py -- Mcell-steps/s of the Godunov + HLLC fp64 step on the synthetic flat-DEM dam-break (S-DAM).

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
N = 1 : BASELINE.json configs[1], 4096 x 4096, one MI355X.
N > 1 : one process per GPU (torch.distributed.run), 1-D row strips with per-step ghost-row exchange and an 8-byte
        MAX all-reduce over RCCL, the per-iteration loop run by the library itself (hp_strip_step_batch); weak scaling
        along the configs' ladder 4096^2 -> 8192x4096 -> 8192^2 -> 16384x8192 (configs[3] at N = 8), i.e. 16,777,216
        cells per GPU at every N.  `--scaling strong` cuts the 4096^2 grid into N strips instead.

The timed region starts with all inputs resident in HBM.  An untimed, time-based pre-warm (--prewarm-s) settles clocks
and caches; the state is then put back from a device-side checkpoint (hp_state_restore), W untimed warm-up steps run,
and EXACTLY K steps are timed (barrier + device sync on both sides).  That is done `--repeats` times -- every repeat
times the same steps [W, W+K) of the workload from t = 0 -- and the MEDIAN repeat is reported.  `--evolve-steps E`
moves the timed window to [W+E, W+E+K): a developed flood instead of the first moments after the dam has gone.  `roofline` prices the flux kernel from HIP events recorded on
the domain's own stream (sparse samples, events created before the timed region); `roofline_manning_array` is the
same kernel with a spatially varying Manning array; `cpu_baseline` times the reference's kernel sources compiled for
the host (oracle/_ref) / the plain-C oracle on the host cores.
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


def grid_for(n_gpus, cols, rows, scaling="weak"):
    if cols and rows:
        return cols, rows
    if scaling == "strong":
        return LADDER[1]                 # configs[1]'s 4096 x 4096 cut into N strips (north_star's strong-scaling target)
    if n_gpus in LADDER:
        return LADDER[n_gpus]
    return 4096, 4096 * n_gpus


def pmc_traffic(kernel_substr, scheme):
    """HBM bytes per launch of the flux kernel from the newest committed PMC summary of THIS workload (profiles/
    rNN*_<scheme>4096_pmc.json, produced by tools/profile_bench.sh + tools/summarize_profile.py from the default bench
    command: FETCH_SIZE x2 + WRITE_SIZE, separate passes)."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{scheme}4096_*pmc.json"))):
        if "developed" in os.path.basename(f):
            continue
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        vals = [k["hbm_bytes_per_launch"] for name, k in d.get("kernels", {}).items()
                if kernel_substr in name and "hbm_bytes_per_launch" in k]
        if vals:
            best = (sum(vals) / len(vals), os.path.basename(f))
    return best


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
    """GPUs this process could use, WITHOUT initialising the HIP runtime (the launcher must stay GPU-free: it starts
    the rank processes and a process that has touched the GPU must not spawn-and-replace itself on this pool)."""
    import torch
    return int(torch.cuda.device_count())


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--cols", type=int, default=0)
    ap.add_argument("--rows", type=int, default=0)
    ap.add_argument("--scheme", choices=["godunov", "muscl", "inertial"], default="godunov")
    ap.add_argument("--precision", choices=["f64", "f32"], default="f64")
    ap.add_argument("--math", choices=["fast", "strict"], default="fast")
    ap.add_argument("--kernel", choices=["auto", "basic"], default="auto")
    ap.add_argument("--workload", choices=["s-dam", "s-rain", "s-rough"], default="s-dam",
                    help="s-dam: BASELINE configs[1..3]; s-rain: configs[4] (initially dry terrain + gridded rainfall, dx = 2 m); "
                         "s-rough: SURVEY 8(d)'s wet/dry rough terrain at full size (every tile on the general path; N = 1)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N > 1: weak = 16.8 Mcell per GPU along the configs' ladder (default); strong = 4096^2 cut into N strips")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-manning-leg", action="store_true")
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
    from hipims_mi import synthetic as syn

    cols, rows = grid_for(world, args.cols, args.rows, args.scaling)
    scheme = {"godunov": hp.SCHEME_GODUNOV, "muscl": hp.SCHEME_MUSCL_HANCOCK, "inertial": hp.SCHEME_INERTIAL}[args.scheme]
    # the partial-inertial scheme is only meaningful for a gentle step (2.0 m | 1.6 m instead of 10 m | 1 m)
    levels = (2.0, 1.6) if args.scheme == "inertial" else (10.0, 1.0)
    math_mode = hp.MATH_FAST if args.math == "fast" else hp.MATH_STRICT
    kernel = hp.KERNEL_AUTO if args.kernel == "auto" else hp.KERNEL_BASIC
    real = np.float64 if args.precision == "f64" else np.float32

    dx = 2.0 if args.workload == "s-rain" else 1.0
    if world == 1:
        from hipims_mi.strips import SingleRunner as Runner
        runner = Runner(cols, rows, dx=dx, scheme=scheme, precision=args.precision, math_mode=math_mode, kernel=kernel,
                        device=local_rank)
    else:
        from hipims_mi.strips import StripRunner as Runner
        # HIPIMS_MI_BACKEND=gloo: rehearsal of this branch with several processes on one GPU (host-staged exchange)
        backend = os.environ.get("HIPIMS_MI_BACKEND", "nccl")
        device = local_rank if backend == "nccl" else local_rank % max(1, hp.device_count())
        runner = Runner(cols, rows, dx=dx, scheme=scheme, precision=args.precision, math_mode=math_mode, kernel=kernel,
                        device=device, rank=rank, world=world, backend=backend)

    if args.workload == "s-rain":
        st, bed, man, rain = syn.s_rain_rows(cols, rows, runner.local_lo, runner.local_hi, dx=dx, dtype=real)
        runner.upload(st, bed, man)
        runner.domain.add_gridded(hp.GRIDDED_RAIN_INTENSITY, rain["grids"], rain["resolution"], rain["off_x"],
                                  rain["off_y"], rain["interval"])
    elif args.workload == "s-rough":
        if world != 1:
            raise SystemExit("--workload s-rough is a single-GPU diagnostic")
        st, bed, man = syn.s_rough(cols, rows, dtype=real, manning=0.03)
        runner.upload(st, bed, man)
    else:
        st, bed, man = (syn.s_dam(cols, runner.local_rows_total, dtype=real, levels=levels) if world == 1
                        else runner.make_s_dam(real, levels=levels))
        runner.upload(st, bed, man)
    del st, bed, man
    runner.set_target_time(1e9)

    # ---- untimed: a time-based pre-warm so that a short run (the driver's 20 steps are 6 ms of GPU time) is not
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

    # ---- `repeats` x (restore, W untimed warm-up steps, EXACTLY K timed steps bracketed by barrier + device sync);
    #      the median repeat is reported.  The flux kernel is sampled sparsely (<= 16 launches per repeat, events
    #      created beforehand) ----
    stride = max(1, args.steps // 12) | 1              # odd: both CFL flavours of the kernel get sampled
    runs = []
    for _ in range(max(1, args.repeats)):
        runner.restore()
        runner.step(args.warmup + args.evolve_steps)
        runner.domain.kernel_timing(stride)
        runner.barrier()
        t0 = time.perf_counter()
        runner.step(args.steps)
        runner.barrier()
        el = runner.max_over_ranks(time.perf_counter() - t0)
        k_ms, k_n = runner.domain.kernel_timing_read()
        runs.append((el, k_ms, k_n))
    runs.sort()
    elapsed, k_ms, k_n = runs[len(runs) // 2]
    k_ms = sorted(r[1] for r in runs)[len(runs) // 2]
    sc = runner.domain.read_scalars()
    # ---- N > 1: a number is only a measurement if the strips really saw each other's rows.  After the timed batch every
    #      rank's ghost rows must equal, bit for bit, the rows their owners hold -- checked over torch.distributed's own
    #      channel, whatever transport the strip loop used (RCCL send/receive, or the strips' direct writes over xGMI) ----
    ghost_rows_bad = runner.verify_ghost_rows() if world > 1 else 0
    if ghost_rows_bad:
        raise SystemExit(f"bench.py --gpus {world}: {ghost_rows_bad} ghost-row cells differ from their owners' values after the "
                         f"timed batch -- the strips did not exchange correctly; no line is printed for a broken run")

    # ---- the same kernel with a spatially varying Manning array (the uniform n of S-DAM is passed as a scalar and
    #      its 8 B/cell are not streamed): one more repeat, N = 1 only ----
    manning_leg = None
    if world == 1 and args.workload == "s-dam" and not args.no_manning_leg:
        rng = np.random.default_rng(11)
        man = (0.03 + rng.uniform(-0.005, 0.005, (rows, cols))).astype(real)
        runner.domain.upload(manning=man)                  # a 128 MiB host copy: the clocks settle again afterwards
        del man
        t_warm = time.perf_counter()
        while runner.max_over_ranks(time.perf_counter() - t_warm) < args.prewarm_s:
            runner.step(25)
            runner.barrier()
        runner.restore()
        runner.step(args.warmup + args.evolve_steps)
        runner.domain.kernel_timing(stride)
        runner.barrier()
        t0 = time.perf_counter()
        runner.step(args.steps)
        runner.barrier()
        el_m = time.perf_counter() - t0
        km_ms, km_n = runner.domain.kernel_timing_read()
        manning_leg = (el_m, km_ms, km_n)

    strip_info = runner.domain.strip_info() if world > 1 and getattr(runner, "loop", "") == "cxx" else None
    if rank == 0:
        cells = cols * rows
        value = cells * args.steps / elapsed / 1e6
        cells_per_launch = cols * runner.local_rows_total
        bpc = BYTES_PER_CELL_STEP[args.precision]
        achieved = bpc * cells_per_launch / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        out = {
            "metric": "Mcell-steps/sec fp64 Godunov+HLLC, 4096^2 grid, 1/2/4/8 MI355X; % HBM roofline",
            "value": value, "unit": "Mcell-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak",
            "repeats_ms_per_step": [r[0] / args.steps * 1e3 for r in runs],
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"{ {'s-rain': 'S-RAIN gridded-rainfall on dry terrain', 's-rough': 'S-ROUGH wet/dry rough terrain', 's-dam': 'S-DAM flat-DEM dam-break'}[args.workload] } "
                                   f"{cols}x{rows}{'' if levels[0] == 10.0 else ' (levels %g|%g m)' % levels}, "
                                   f"{args.scheme + '+HLLC' if args.scheme != 'inertial' else 'partial-inertial'}, friction fused, "
                                   f"dynamic CFL dt, quirks=reference, math={args.math}, kernel={args.kernel}",
                       "cells_per_gpu": cells // world, "parallelism": f"row-strips x{world}" + (f", per-iteration loop: {getattr(runner, 'loop', 'batch call')}" if world > 1 else "") + (f", collective library {strip_info['library']} reporting {strip_info['comm_ranks']} ranks, halo overlap {'on' if strip_info['halo_overlap'] else 'off'}, maximum over the strips by {'peer-written mailboxes' if strip_info['peer_max'] else 'all-reduce'}, ghost rows {'stored into the neighbours by the strips themselves' if strip_info['peer_halo'] else 'sent and received through the library'}" if strip_info else "") + ("" if world == 1 or os.environ.get("HIPIMS_MI_BACKEND", "nccl") == "nccl"
                                                                else " (REHEARSAL: gloo, host-staged exchange, shared GPU -- not a measurement)"),
                       "area_boundaries": ("none" if args.workload != "s-rain" else
                                           "fused into the flux kernel's store epilogue" if runner.domain.boundaries_fused() else "separate pass"),
                       "timed_steps": [args.warmup + args.evolve_steps, args.warmup + args.evolve_steps + args.steps],
                       "sim_time_s": sc["time"], "successful_iterations": sc["batch_successful"],
                       **({"ghost_rows_verified_after_timed_batch": True} if world > 1 else {})},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": runner.flux_kernel_name, "avg_launch_ms": k_ms, "launches_sampled": k_n,
                         "algorithmic_bytes_per_cell_step": bpc, "cells_per_launch": cells_per_launch},
        }
        if args.workload in ("s-dam", "s-rain", "s-rough"):
            # both synthetic workloads have ONE Manning value, which the engine passes as a scalar: the bytes this
            # launch really has to move are 72 (fp64) / 36 (fp32) per cell; SURVEY 8(d)'s contract figure stays above
            wb = bpc * 0.9
            out["roofline"]["uniform_manning_bytes_per_cell_step"] = wb
            out["roofline"]["frac_at_uniform_manning_bytes"] = wb * cells_per_launch / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if k_ms > 0 else 0.0
        if manning_leg:
            el_m, km_ms, km_n = manning_leg
            ach_m = bpc * cells_per_launch / (km_ms * 1e-3) / 1e9 if km_ms > 0 else 0.0
            out["roofline_manning_array"] = {"what": "same workload with a spatially varying Manning array (all 80 B/cell streamed)",
                                             "value": cells * args.steps / el_m / 1e6, "achieved": ach_m, "frac": ach_m / HBM_PEAK_GBS,
                                             "avg_launch_ms": km_ms, "launches_sampled": km_n}
        default_cfg = (cols, rows) == (4096, 4096) and args.precision == "f64" and args.kernel == "auto" \
            and args.workload == "s-dam" \
            and args.math == "fast" and world == 1
        if default_cfg:
            tr = pmc_traffic(args.scheme + "_march<false", args.scheme)
            if tr:
                out["roofline"]["traffic"] = tr[0] / 1e9 / (k_ms * 1e-3) if k_ms > 0 else None   # GB/s, same unit as achieved
                out["roofline"]["traffic_bytes_per_launch"] = tr[0]
                out["roofline"]["traffic_source"] = "profiles/" + tr[1]
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cols, args.precision, scheme, levels)
        print(json.dumps(out), flush=True)
    runner.close()


if __name__ == "__main__":
    main()
