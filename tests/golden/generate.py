#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Runs only where /root/reference exists: the reference's OpenCL C kernel sources are compiled for the
host (oracle/ref_build -> oracle/_ref/*.so, strict = -ffp-contract=off) and executed on seeded inputs;
inputs and outputs are stored as .npz.  The fixtures are data only -- no reference source travels.

    python tests/golden/generate.py          # rewrites tests/golden/*.npz

Fixture list follows SURVEY.md section 8(c) (F1..F10), plus F11 (cell boundaries) and F12 (partial-inertial scheme).
`python tests/golden/generate.py f12` regenerates only the fixtures whose file name starts with f12.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "hipims-ocl_amd"))

import oracle  # noqa: E402
from hipims_mi import synthetic as syn  # noqa: E402

VS = 1e-10


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.0f} KiB")


def random_cells(rng, n, real):
    """Random (state, bed) pairs mixing deep / shallow / dry / stepped-bed cases."""
    bed = rng.uniform(-1.0, 1.0, n)
    kind = rng.integers(0, 5, n)
    depth = np.where(kind == 0, 0.0,                                  # dry
            np.where(kind == 1, rng.uniform(0, 2e-10, n),             # around VERY_SMALL
            np.where(kind == 2, rng.uniform(1e-6, 1e-3, n),           # thin film
                     rng.uniform(0.01, 3.0, n))))                     # ordinary
    st = np.zeros((n, 4))
    st[:, 0] = bed + depth
    st[:, 1] = st[:, 0]
    vel = rng.uniform(-3.0, 3.0, (n, 2)) * (rng.random((n, 1)) > 0.2)
    st[:, 2] = depth * vel[:, 0]
    st[:, 3] = depth * vel[:, 1]
    # a few dry cells that still carry discharge (exercise the stopping conditions)
    odd = (kind == 0) & (rng.random(n) < 0.5)
    st[odd, 2] = rng.uniform(-0.1, 0.1, odd.sum())
    st[odd, 3] = rng.uniform(-0.1, 0.1, odd.sum())
    return st.astype(real), bed.astype(real)


def function_level(precision, tag):
    real = np.float64 if precision == "f64" else np.float32
    ref = oracle.RefFunctions(precision=precision)
    rng = np.random.default_rng(20240917)
    n = 384
    sL, bL = random_cells(rng, n, real)
    sR, bR = random_cells(rng, n, real)
    # equal beds on a third of the pairs (the common flat case)
    same = rng.random(n) < 0.33
    dz = (sR[:, 0] - bR).copy()
    bR[same] = bL[same]
    sR[same, 0] = bR[same] + dz[same]
    sR[:, 1] = sR[:, 0]

    rec_L = np.zeros((4, n, 8), real); rec_R = np.zeros((4, n, 8), real); rec_stop = np.zeros((4, n), np.int32)
    flux = np.zeros((4, n, 4), real)
    for d in range(4):
        for i in range(n):
            oL, oR, stop = ref.reconstruct(d, sL[i], bL[i], sR[i], bR[i])
            rec_L[d, i], rec_R[d, i], rec_stop[d, i] = oL, oR, stop
            flux[d, i] = ref.hllc(d, oL, oR)
    save(f"f1_f2_reconstruct_hllc_{tag}", sL=sL, bL=bL, sR=sR, bR=bR, rec_L=rec_L, rec_R=rec_R,
         rec_stop=rec_stop, flux=flux, very_small=np.array(VS))

    # friction
    st, bd = random_cells(rng, 768, real)
    man = rng.uniform(0.01, 0.08, 768).astype(real)
    dts = (10.0 ** rng.uniform(-4, 0.5, 768)).astype(real)
    fr = np.array([ref.friction(st[i], bd[i], man[i], dts[i]) for i in range(768)], real)
    save(f"f3_friction_{tag}", state=st, bed=bd, manning=man, dt=dts, out=fr)

    # limiter
    a, ba = random_cells(rng, 768, real); b, bb = random_cells(rng, 768, real); c, bc = random_cells(rng, 768, real)
    dup = rng.random(768) < 0.15          # zero left difference / sign changes
    b[dup] = a[dup]
    lim = np.array([ref.limiter(a[i], b[i], c[i], ba[i], bb[i], bc[i]) for i in range(768)], real)
    save(f"f4_limiter_{tag}", sL=a, sC=b, sR=c, bL=ba, bC=bb, bR=bc, out=lim)

    # MUSCL predictor + 2nd-order reconstruction
    m = 384
    sts = np.zeros((m, 5, 4), real); bds = np.zeros((m, 5), real)
    base, bbase = random_cells(rng, m, real)
    for k in range(5):
        pert, bpert = random_cells(rng, m, real)
        smooth = rng.random(m) < 0.7      # mostly smooth neighbourhoods so slopes are non-trivial
        bds[:, k] = np.where(smooth, bbase + 0.02 * (k - 2), bpert)
        depth = np.where(smooth, np.maximum(base[:, 0] - bbase, 0) * (1 + 0.05 * rng.standard_normal(m)), pert[:, 0] - bpert)
        depth = np.maximum(depth, 0)
        sts[:, k, 0] = bds[:, k] + depth
        sts[:, k, 1] = sts[:, k, 0]
        sts[:, k, 2] = np.where(smooth, base[:, 2] * (1 + 0.1 * rng.standard_normal(m)), pert[:, 2])
        sts[:, k, 3] = np.where(smooth, base[:, 3] * (1 + 0.1 * rng.standard_normal(m)), pert[:, 3])
    sts[rng.random(m) < 0.05, 2, 1] = -9999.0           # a disabled neighbour -> first-order fallback
    dts = (10.0 ** rng.uniform(-3, -1, m)).astype(real)
    faces = np.zeros((m, 4, 4), real); first = np.zeros(m, np.int32)
    for i in range(m):
        faces[i], first[i] = ref.mch_1st(dts[i], sts[i].reshape(-1), bds[i])
    r2L = np.zeros((4, m, 8), real); r2R = np.zeros((4, m, 8), real); r2stop = np.zeros((4, m), np.int32)
    perm = rng.permutation(m)
    for d in range(4):
        for i in range(m):
            j = perm[i]
            # left/right estimates: opposite faces of two predictor outputs (N<->S, E<->W)
            eL, eR = (faces[i, d], faces[j, (d + 2) % 4]) if d < 2 else (faces[j, (d + 2) % 4], faces[i, d])
            oL, oR, stop = ref.reconstruct2(d, sts[i, 0], bds[i, 0], sts[j, 0], bds[j, 0], eL, eR)
            r2L[d, i], r2R[d, i], r2stop[d, i] = oL, oR, stop
    save(f"f5_muscl_predict_{tag}", states=sts, beds=bds, dt=dts, faces=faces, first=first, perm=perm,
         r2L=r2L, r2R=r2R, r2stop=r2stop)


def trajectories(precision, tag, mad=False):
    real = np.float64 if precision == "f64" else np.float32
    out = {}
    st, bed, man = syn.s_rough(64, 64, dtype=real, manning=None)
    out.update(rough_state=st, rough_bed=bed, rough_manning=man)
    for scheme, sname in ((oracle.GODUNOV, "god"), (oracle.MUSCL, "mch")):
        for quirks, qname in ((oracle.QUIRKS_REFERENCE, "q"), (oracle.QUIRKS_REFERENCE & ~oracle.Q1_CFL_READS_PRIMARY, "noq1")):
            if scheme == oracle.MUSCL and qname == "noq1":
                continue
            sim = oracle.RefSim(64, 64, scheme=scheme, precision=precision, quirks=quirks, mad=mad)
            sim.upload(st, bed, man)
            sim.set_target(1e9)
            t1 = sim.run(1)
            s1 = sim.download()
            t2 = sim.run(199)
            out[f"{sname}_{qname}_state1"] = s1
            out[f"{sname}_{qname}_state200"] = sim.download()
            out[f"{sname}_{qname}_dt"] = np.concatenate([t1, t2])
            out[f"{sname}_{qname}_t"] = np.array(sim.scalars()["t"])
    # all-wet dam break with walls (the benchmark's shape, small)
    st, bed, man = syn.s_dam(96, 48, dtype=real)
    for scheme, sname in ((oracle.GODUNOV, "god"), (oracle.MUSCL, "mch")):
        sim = oracle.RefSim(96, 48, scheme=scheme, precision=precision, mad=mad)
        sim.upload(st, bed, man)
        sim.set_target(1e9)
        out[f"dam_{sname}_dt"] = sim.run(150)
        out[f"dam_{sname}_state150"] = sim.download()
    # dam break onto a dry bed
    st, bed, man = syn.s_dam(96, 48, dtype=real, wet_right=False)
    sim = oracle.RefSim(96, 48, scheme=oracle.GODUNOV, precision=precision, mad=mad)
    sim.upload(st, bed, man)
    sim.set_target(1e9)
    out["damdry_god_dt"] = sim.run(150)
    out["damdry_god_state150"] = sim.download()
    save(f"f6_f7_trajectories_{tag}", **out)


def fp32_bracket():
    """The reference's OWN compiler-dependent spread in single precision: the F6/F7 rough-bed case through its fp32
    program built as shipped (-cl-mad-enable, COCLProgram.cpp:73).  MUSCL-Hancock on wet/dry terrain is chaotic at
    fp32 resolution (thresholds of 1e-10 m against a level ulp of 3e-8 m): strict and shipped builds part by
    RMSE 2.8e-4 m / max 1.2e-2 m after 200 iterations, Godunov by 5e-8 m."""
    st, bed, man = syn.s_rough(64, 64, dtype=np.float32, manning=None)
    out = {}
    for scheme, sname in ((oracle.GODUNOV, "god"), (oracle.MUSCL, "mch")):
        sim = oracle.RefSim(64, 64, scheme=scheme, precision="f32", mad=True)
        sim.upload(st, bed, man)
        sim.set_target(1e9)
        sim.run(200)
        out[f"{sname}_q_state200"] = sim.download()
        out[f"{sname}_q_t"] = np.array(sim.scalars()["t"])
    save("f6_f7_bracket_f32_mad", **out)


def time_control(precision, tag):
    """F8: tst_Advance_Normal / tst_UpdateTimestep scalar traces driven with prescribed wave speeds."""
    real = np.float64 if precision == "f64" else np.float32
    rng = np.random.default_rng(8)
    sim = oracle.RefSim(8, 8, precision=precision, end_time=20.0, dx=2.0)
    sim._configure()
    C = oracle.C
    P = oracle._ptr
    rows = []
    speeds = np.concatenate([[0.0, 0.0], 10 ** rng.uniform(-2, 1.3, 400)]).astype(real)
    sim.t_sync[0] = 5.0
    for i, v in enumerate(speeds):
        sim.scratch[:] = 0
        sim.scratch[i % sim.WORKERS] = v
        if i == 150:
            sim.t_sync[0] = 12.0        # host moves the sync point on (CSchemeGodunov.cpp:1166-1176)
        if i == 300:
            sim.t_sync[0] = 1e9
        sim.lib.ref_advance(P(sim.t), P(sim.dt), P(sim.t_hydro), P(sim.scratch), P(sim.primary), P(sim.bed),
                            P(sim.t_sync), P(sim.batch_dt), P(sim.ok), P(sim.skipped))
        rows.append([sim.t[0], sim.dt[0], sim.t_hydro[0], sim.t_sync[0], sim.batch_dt[0], sim.ok[0], sim.skipped[0]])
    adv = np.array(rows, np.float64)
    # tst_UpdateTimestep from a set of states
    upd_in, upd_out = [], []
    for _ in range(200):
        t = float(rng.uniform(0, 100)); dt = float(rng.uniform(-0.5, 0.5)); ts = float(t + rng.uniform(-0.1, 2.0))
        bdt = float(rng.uniform(0, 10)); v = float(10 ** rng.uniform(-2, 1.3))
        sim.t[0], sim.dt[0], sim.t_sync[0], sim.batch_dt[0] = t, dt, ts, bdt
        upd_in.append([sim.t[0], sim.dt[0], sim.t_sync[0], sim.batch_dt[0], real(v)])
        sim.scratch[:] = 0
        sim.scratch[3] = v
        sim.lib.ref_update_timestep(P(sim.t), P(sim.dt), P(sim.scratch), P(sim.t_sync), P(sim.batch_dt))
        upd_out.append([sim.dt[0], sim.batch_dt[0]])
    save(f"f8_time_control_{tag}", speeds=speeds, advance=adv, update_in=np.array(upd_in, np.float64),
         update_out=np.array(upd_out, np.float64), dx=np.array(2.0), end_time=np.array(20.0))


def rain(precision, tag):
    """F9: uniform rain + loss and gridded rain on 48x40 (not multiples of 8 -> quirk Q9 visible)."""
    real = np.float64 if precision == "f64" else np.float32
    cols, rows = 50, 43
    st, bed, man = syn.s_rough(cols, rows, dtype=real, pool_level=-10.0, amplitude=0.2, walls=False)   # dry
    st[..., 2:] = 0
    series = np.array([[0, 70.0], [3600, 70.0], [7200, 0.0], [10800, 0.0]], real)       # test/newcastle-centre rain CSV shape
    loss = np.array([[0, 12.0], [10800, 12.0]], real)
    rng = np.random.default_rng(7)
    grids = rng.uniform(0, 120, (3, 5, 6)).astype(real)
    out = dict(state=st, bed=bed, manning=man, series=series, loss=loss, grids=grids)
    for name, setup in (("uniform", lambda s: (s.add_uniform(oracle.UNIFORM_RAIN_INTENSITY, series, 3600.0, 10800.0),
                                                 s.add_uniform(oracle.UNIFORM_LOSS_RATE, loss, 10800.0, 10800.0))),
                        ("gridded", lambda s: s.add_gridded(oracle.GRIDDED_RAIN_INTENSITY, grids, 10.0, 0.0, 0.0, 20.0))):
        sim = oracle.RefSim(cols, rows, precision=precision, dx=1.0)
        sim.upload(st, bed, man)
        setup(sim)
        sim.set_target(1e9)
        out[f"{name}_dt"] = sim.run(420)
        out[f"{name}_state"] = sim.download()
        out[f"{name}_t"] = np.array(sim.scalars()["t"])
    save(f"f9_rain_{tag}", **out)


CELL_MODES = [("depth_q", 2, 1), ("fsl_vel", 1, 2), ("free_q", 0, 1), ("free_volume", 0, 3)]


def cell_boundary(precision, tag):
    """F11: bdy_Cell -- imposed depth / level / discharge / velocity / volume on a short line of cells."""
    real = np.float64 if precision == "f64" else np.float32
    cols, rows = 60, 44
    st, bed, man = syn.s_rough(cols, rows, dtype=real, pool_level=-10.0, amplitude=0.2, walls=True)
    st[..., 2:] = 0
    series = np.array([[0, 0.0, 0.0, 0.0], [5, 0.3, 0.4, 0.1], [10, 0.6, 0.8, 0.0], [15, 0.2, 0.1, -0.2], [20, 0.0, 0.0, 0.0]], real)
    cells = np.array([10 * cols + 5, 11 * cols + 5, 12 * cols + 5, 30 * cols + 40], np.uint64)
    out = dict(state=st, bed=bed, manning=man, series=series, cells=cells)
    for name, dd, qd in CELL_MODES:
        sim = oracle.RefSim(cols, rows, precision=precision)
        sim.upload(st, bed, man)
        sim.add_cell(dd, qd, cells, series, 5.0, 20.0)
        sim.set_target(1e9)
        out[f"{name}_dt"] = sim.run(300)
        out[f"{name}_state"] = sim.download()
        out[f"{name}_t"] = np.array(sim.scalars()["t"])
    save(f"f11_cell_boundary_{tag}", **out)


def newcastle(precision, tag, mad=False, libm_pow=False):
    """F10: config C1 -- the reference's own example (342x195 @ 2 m DEM from its data file, rain 70 mm/h + drainage
    12 mm/h, closed edges), Godunov fp64, first 900 iterations on the reference's kernels.  The -cl-mad-enable build
    (what the reference ships) and the build whose pow() is the host libm's (another conforming OpenCL platform) store
    only the final levels and time: together they bracket the reference's own platform-dependent spread."""
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from hipims_mi import frontend
    from model_dir import make_newcastle
    with tempfile.TemporaryDirectory() as tmp:
        cfg = frontend.parse_configuration(make_newcastle(tmp))
        st, bed, man, res = frontend.build_domain(cfg)
    sim = oracle.RefSim(342, 195, precision=precision, dx=res, end_time=cfg.duration, mad=mad, libm_pow=libm_pow)
    sim.upload(st, bed, man)
    frontend.attach_boundaries(cfg, sim, 342)
    sim.set_target(1e9)
    dt = sim.run(900)
    final = sim.download()
    if mad or libm_pow:
        save(f"f10_newcastle_{tag}", z=final[..., 0], t=np.array(sim.scalars()["t"]))
        return
    depth = np.maximum(0, final[..., 0] - bed)
    save(f"f10_newcastle_{tag}", dt=dt, t=np.array(sim.scalars()["t"]), depth=depth.astype(np.float32),
         z=final[..., 0], qx=final[..., 2].astype(np.float32), qy=final[..., 3].astype(np.float32))


C1_BATCH = 512          # the reference's queueMode="fixed" with this queue size: batch boundaries (and with them the number of
                        # suspended iterations behind every output time, quirk Q1's buffer parity) are part of the run


def newcastle_full(tag, mad=False, libm_pow=False, threads=8):
    """F16: config C1 TO ITS END -- test/newcastle-centre.xml is 7200 s with rasters every 600 s (CModel.cpp:723-770: the
    target time is clipped to every output time; CDomainCartesian.cpp:804-829), 1.0e5 iterations of the reference's kernels
    under the front end's CModel loop (hipims_mi.model.Model with the reference kernels as its `sim`), fixed batches of
    C1_BATCH.  Stored for the strict build: the simulation time, iteration counts and the SHA-256 of the float64 depth and
    maxdepth rasters (as derive_output computes them) at all twelve output times -- what a bit-for-bit comparison needs -- and
    the rasters themselves at four of them (600, 2400, 4800, 7200 s; the tolerance curve of the FAST mode).  The
    -cl-mad-enable build and the libm-pow build store depth at 2400 s and 7200 s (the bracket)."""
    import hashlib
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from model_dir import make_newcastle
    from hipims_mi.model import Model

    def make_sim(cfg, cols, rows, res):
        return oracle.RefSim(cols, rows, precision="f64", dx=res, end_time=cfg.duration, mad=mad, libm_pow=libm_pow, threads=threads)
    with tempfile.TemporaryDirectory() as tmp:
        m = Model(make_newcastle(tmp), make_sim=make_sim, output_format=None)
        m.scheme.automatic_queue = False
        m.scheme.queue_addition_size = C1_BATCH
        outs = m.run()
        iterations, ok = m.scheme.iterations, m.scheme.batch_successful
    times = np.array([t for t, _ in outs])
    assert len(outs) == 12 and abs(times[-1] - 7200.0) < 1e-5
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a, dtype=np.float64).tobytes()).hexdigest()
    if mad or libm_pow:
        save(f"f16_newcastle_full_{tag}", t=times[[3, 11]], depth=np.stack([outs[k][1]["depth"] for k in (3, 11)]),
             iterations=np.array(iterations), successful=np.array(ok))
        return
    keep = (0, 3, 7, 11)
    save(f"f16_newcastle_full_{tag}", t=times, kept=np.array(keep), depth=np.stack([outs[k][1]["depth"] for k in keep]),
         maxdepth=outs[-1][1]["maxdepth"], depth_sha256=np.array([sha(o["depth"]) for _, o in outs]),
         maxdepth_sha256=np.array([sha(o["maxdepth"]) for _, o in outs]),
         iterations=np.array(iterations), successful=np.array(ok), batch=np.array(C1_BATCH))


def libm_twins(tag="f64"):
    """F17: the stand-in that carries arithmetic in the reference build is pow (oracle/ref_build/shim.cpp: the correctly rounded
    cube root of hp_crmath.h).  These are the F6 / F7 rough-bed trajectories (Godunov, MUSCL-Hancock) and F12's rough-bed
    trajectory (partial-inertial) on the SAME strict program with the host libm's pow instead -- another conforming OpenCL
    platform.  Final states only: the distance between the two builds is the independent bound on that stand-in, beyond C1."""
    out = {}
    st, bed, man = syn.s_rough(64, 64, manning=None)
    for scheme, sname in ((oracle.GODUNOV, "god"), (oracle.MUSCL, "mch")):
        sim = oracle.RefSim(64, 64, scheme=scheme, libm_pow=True)
        sim.upload(st, bed, man)
        sim.set_target(1e9)
        sim.run(200)
        out[f"{sname}_q_state200"] = sim.download()
        out[f"{sname}_q_t"] = np.array(sim.scalars()["t"])
    st2 = st.copy()
    st2[..., 2:] = 0
    sim = oracle.RefSim(64, 64, scheme=oracle.INERTIAL, libm_pow=True)
    sim.upload(st2, bed, man)
    sim.set_target(1e9)
    sim.run(200)
    out["ine_rough_q_state"] = sim.download()
    out["ine_rough_q_t"] = np.array(sim.scalars()["t"])
    save(f"f17_libm_twins_{tag}", **out)


def inertial(precision, tag, mad=False):
    """F12: calculateInertialFlux table + ine_cacheDisabled trajectories (CLSchemeInertial.clc)."""
    real = np.float64 if precision == "f64" else np.float32
    out = {}
    if not mad:
        ref = oracle.RefFunctions(precision=precision)
        rng = np.random.default_rng(12)
        n = 4096
        bu, bd = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
        kind = rng.integers(0, 5, (2, n))
        depth = lambda k: np.where(k == 0, 0.0, np.where(k == 1, rng.uniform(0, 2e-10, n),
                          np.where(k == 2, rng.uniform(1e-6, 1e-3, n), rng.uniform(0.01, 3.0, n))))
        args = np.stack([rng.choice([0.0, 0.03, 0.06], n) + rng.uniform(0, 0.02, n) * (rng.random(n) < 0.5),   # n
                         rng.choice([0.001, 0.05, 0.1], n) * rng.uniform(0.5, 1.0, n),                         # dt
                         rng.normal(0, 1.5, n) * (rng.random(n) > 0.25),                                       # q_prev
                         bu + depth(kind[0]), bu, bd + depth(kind[1]), bd], axis=1).astype(real)
        res = np.array([ref.inertial_flux(*a) for a in args], real)
        out.update(flux_args=args, flux_out=res)

    def run(name, cols, rows, st, bed, man, steps, quirks=oracle.QUIRKS_REFERENCE, first=False, setup=None):
        sim = oracle.RefSim(cols, rows, scheme=oracle.INERTIAL, precision=precision, quirks=quirks, mad=mad)
        sim.upload(st, bed, man)
        sim.set_target(1e9)
        if setup:
            setup(sim)
        dts = []
        if first:
            dts.append(sim.run(1))
            out[f"{name}_state1"] = sim.download()
            steps -= 1
        dts.append(sim.run(steps))
        out[f"{name}_state"] = sim.download()
        out[f"{name}_dt"] = np.concatenate(dts)
        out[f"{name}_t"] = np.array(sim.scalars()["t"])
        return sim

    st, bed, man = syn.s_rough(64, 64, dtype=real, manning=None)            # spatially varying n: four fluxes per cell
    st[..., 2:] = 0
    out.update(rough_state=st, rough_bed=bed, rough_manning=man)
    run("rough_q", 64, 64, st, bed, man, 200, first=True)
    run("rough_noq1", 64, 64, st, bed, man, 200, quirks=oracle.QUIRKS_REFERENCE & ~oracle.Q1_CFL_READS_PRIMARY)
    st, bed, man = syn.s_dam(96, 48, dtype=real)                             # uniform n: shared faces
    run("dam", 96, 48, st, bed, man, 150)                                    # 10 m -> 1 m: far outside the scheme's range
    st, bed, man = syn.s_dam(96, 48, dtype=real, levels=(2.0, 1.6))          # gentle step: what the scheme is for
    run("step", 96, 48, st, bed, man, 150)
    st, bed, man = syn.s_dam(96, 48, dtype=real, wet_right=False)
    run("damdry", 96, 48, st, bed, man, 150)
    # sync point: the target time is reached, iterations are skipped (dt <= 0: the kernel leaves dst untouched), then
    # the target moves on (Threaded_runBatch: new target -> tst_UpdateTimestep)
    st, bed, man = syn.s_rough(48, 40, dtype=real, manning=None)
    st[..., 2:] = 0
    out.update(sync_state=st, sync_bed=bed, sync_manning=man)
    sim = run("sync_a", 48, 40, st, bed, man, 40, setup=lambda s: s.set_target(2.0))
    sim.set_target(5.0)
    sim.update_timestep()
    out["sync_b_dt"] = sim.run(45)
    out["sync_b_state"] = sim.download()
    out["sync_b_t"] = np.array(sim.scalars()["t"])
    # uniform rain + gridded rain through the inherited boundary path
    st, bed, man = syn.s_rough(48, 40, dtype=real, pool_level=-10.0, amplitude=0.2, walls=False)
    st[..., 2:] = 0
    series = np.array([[0, 50.0], [10, 120.0], [20, 0.0], [30, 80.0]], real)
    grids = np.random.default_rng(5).uniform(0, 120, (3, 5, 6)).astype(real)
    out.update(rain_init=st, rain_bed=bed, rain_manning=man, rain_series=series, rain_grids=grids)

    def add_rain(s):
        s.add_uniform(oracle.UNIFORM_RAIN_INTENSITY, series, 10.0, 30.0)
        s.add_gridded(oracle.GRIDDED_RAIN_INTENSITY, grids, 10.0, 0.0, 0.0, 15.0)
    run("rain", 48, 40, st, bed, man, 260, setup=add_rain)
    save(f"f12_inertial_{tag}", **out)


def fixed_timestep(tag="f64"):
    """F13: the TIMESTEP_FIXED program (CSchemeGodunov.cpp:735-755, CLDynamicTimestep.clc:92-96): no reduction kernel,
    dt = the configured value, still subject to the sync clip, the early limit (0.1 s before t = 60 s) and the end time."""
    out = {}
    st, bed, man = syn.s_rough(64, 64, manning=None)
    out.update(state=st, bed=bed, manning=man)
    for name, dt, target, end in (("small", 0.02, 2.5, 1e30), ("clipped", 0.25, 1e9, 11.0)):
        sim = oracle.RefSim(64, 64, dynamic_dt=False, fixed_dt=dt, dt_initial=dt, end_time=end)
        sim.upload(st, bed, man)
        sim.set_target(target)
        out[f"{name}_dt"] = sim.run(160)
        out[f"{name}_state"] = sim.download()
        sc = sim.scalars()
        out[f"{name}_t"] = np.array(sc["t"]); out[f"{name}_ok"] = np.array(sc["batch_ok"]); out[f"{name}_skipped"] = np.array(sc["batch_skipped"])
        if name == "small":
            # resume after the sync point: new target -> tst_UpdateTimestep (whose dLclTimestep is uninitialised in this
            # program, CLDynamicTimestep.clc:268/:297: the host build returns |dt|)
            sim.set_target(4.0)
            sim.update_timestep()
            out["resume_dt"] = sim.run(100)
            out["resume_state"] = sim.download()
            out["resume_t"] = np.array(sim.scalars()["t"])
    save(f"f13_fixed_timestep_{tag}", **out)


def no_friction(tag="f64"):
    """F14: frictionEffects="no" (FRICTION_ENABLED undefined): no transcendental on the path, so the STRICT HIP
    kernels must reproduce these states bit for bit."""
    out = {}
    st, bed, man = syn.s_rough(72, 40, manning=None)
    out.update(state=st, bed=bed, manning=man)
    for scheme, name in ((oracle.GODUNOV, "god"), (oracle.MUSCL, "mch")):
        sim = oracle.RefSim(72, 40, scheme=scheme, friction=False)
        sim.upload(st, bed, man)
        sim.set_target(1e9)
        out[f"{name}_dt"] = sim.run(120)
        out[f"{name}_state"] = sim.download()
        out[f"{name}_t"] = np.array(sim.scalars()["t"])
    save(f"f14_no_friction_{tag}", **out)


def disabled_cells(tag="f64"):
    """F15: disabled cells -- the domain's nulls.  Three kinds in a wet/dry rough grid, all three schemes, through a sync point:
    mask-style nulls (a disabledCells raster: Zmax = -9999, the bed keeps its DEM value, CDomain.cpp:351-360; a block and a
    sprinkle), one cell with Z == -9999 (the kernels test that too), DEM-nodata nulls (bed = Z = Zmax = -9999) and one LIVE dry
    cell on a bed of -9998.5.  MUSCL-Hancock is stored for BOTH predictor variants of the reference: `mch` = its default
    configuration (kCachePrediction, CSchemeMUSCLHancock.cpp:46: mch_1st_cachePrediction run as real 16 x 16 work-groups,
    oracle/ref_build/shim.cpp; the neighbours' .y is their BED) and `mchnone` = mch_1st_cacheNone (their Zmax).  Next to the
    mask-style nulls the default stays second order where cacheNone drops to first; next to the -9998.5 bed it is the other way
    round."""
    out = {}
    st, bed, man = syn.s_rough(64, 48, manning=None)
    rng = np.random.default_rng(15)
    dis = np.zeros((48, 64), bool)
    dis[10:18, 20:30] = True
    dis |= rng.random((48, 64)) < 0.03
    dis[0] = dis[-1] = False; dis[:, 0] = dis[:, -1] = False
    st[dis, 1] = -9999.0
    st[5, 5, 0] = -9999.0
    nod = np.zeros_like(dis)
    nod[30:34, 40:44] = True
    bed[nod] = -9999.0; st[nod, 0] = -9999.0; st[nod, 1] = -9999.0
    dis |= nod
    bed[40, 12] = -9998.5; st[40, 12, 0] = -9998.5; st[40, 12, 1] = -9998.5
    out.update(state=st, bed=bed, manning=man, disabled=dis)
    runs = ((oracle.GODUNOV, "god", oracle.QUIRKS_REFERENCE), (oracle.MUSCL, "mch", oracle.QUIRKS_REFERENCE),
            (oracle.MUSCL, "mchnone", oracle.QUIRKS_REFERENCE & ~oracle.Q11_MUSCL_NB_Y_IS_BED),
            (oracle.INERTIAL, "ine", oracle.QUIRKS_REFERENCE))
    for scheme, name, quirks in runs:
        sim = oracle.RefSim(64, 48, scheme=scheme, quirks=quirks)
        sim.upload(st, bed, man)
        sim.set_target(2.5)
        out[f"{name}_dt"] = sim.run(150)
        out[f"{name}_state"] = sim.download()
        out[f"{name}_t"] = np.array(sim.scalars()["t"])
    assert not np.array_equal(out["mch_state"], out["mchnone_state"])
    save(f"f15_disabled_cells_{tag}", **out)


JOBS = [
    ("f1", lambda: [function_level(p, p) for p in ("f64", "f32")]),      # f1..f5
    ("f6", lambda: [trajectories(p, p) for p in ("f64", "f32")] + [trajectories("f64", "f64_mad", mad=True)]),
    ("f6b", fp32_bracket),
    ("f8", lambda: [time_control(p, p) for p in ("f64", "f32")]),
    ("f9", lambda: [rain(p, p) for p in ("f64", "f32")]),
    ("f11", lambda: [cell_boundary(p, p) for p in ("f64", "f32")]),
    ("f10", lambda: [newcastle("f64", "f64"), newcastle("f64", "f64_mad", mad=True),
                    newcastle("f64", "f64_libm", libm_pow=True)]),
    ("f16", lambda: [newcastle_full("f64"), newcastle_full("f64_mad", mad=True), newcastle_full("f64_libm", libm_pow=True)]),
    ("f17", libm_twins),
    ("f13", fixed_timestep),
    ("f14", no_friction),
    ("f15", disabled_cells),
    ("f12", lambda: [inertial(p, p) for p in ("f64", "f32")] + [inertial("f64", "f64_mad", mad=True)]),
]

if __name__ == "__main__":
    oracle.build(ref=True)
    want = sys.argv[1:]
    for key, job in JOBS:
        if not want or key in want:
            job()
