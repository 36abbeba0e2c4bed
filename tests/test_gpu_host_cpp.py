"""The C++ host side (hipims-ocl_amd/host: CScheme-surface mirror + a CModel-shaped driver) against the Python
binding driving the same protocol through the same C ABI.  GPU only."""
import os
import subprocess

import numpy as np
import pytest

import hipims_mi as hp
from conftest import PKG
from hipims_mi import synthetic as syn

pytestmark = pytest.mark.gpu
EXE = os.path.join(PKG, "lib", "run_dambreak")


def python_protocol(cols, rows, duration, freq, scheme, batch, inflow=False):
    """CSchemeGodunov::runSimulation + Threaded_runBatch (CSchemeGodunov.cpp:1374-1453, :1147-1372) as the C++ mirror
    implements them, with a fixed queue size."""
    st, bed, man = syn.s_dam(cols, rows)
    dom = hp.Domain(cols, rows, scheme=scheme, t_end=duration)
    dom.upload(st, bed, man)
    if inflow:                                                      # run_dambreak's "inflow" option
        cells = [y * cols + 1 for y in range(rows // 4, 3 * rows // 4)]
        series = [[0.0, 0.0, 40.0, 0.0], [1.0, 0.0, 120.0, 0.0], [2.0, 0.0, 120.0, 0.0]]
        dom.add_cell(hp.DEPTH_IGNORE, hp.DISCHARGE_IS_VOLUME, cells, series, 1.0, 2.0)
    out, target, cur_target = [], freq, 0.0
    sc = dom.read_scalars()
    while sc["time"] < duration - 1e-9:
        if cur_target != target:
            cur_target = target
            dom.set_target_time(target)
            if sc["timestep"] <= 0.0:
                dom.update_timestep()
            sc2 = sc if sc["timestep"] > 0 else dom.read_scalars()
            if sc["time"] + sc["timestep"] > target + 1e-5:
                dom.force_timestep(target - sc["time"])
        if sc["time"] < target:
            dom.step_batch(batch)
        sc = dom.read_scalars()
        if target - sc["time"] <= 1e-5:
            z = dom.download()[..., 0]
            out.append((sc["time"], float(z.astype(np.float64).sum())))
            target = min(duration, target + freq)
    dom.close()
    return out


@pytest.mark.parametrize("scheme_name,scheme", [("godunov", hp.SCHEME_GODUNOV), ("muscl", hp.SCHEME_MUSCL_HANCOCK)])
def test_cpp_host_driver_matches_python_binding(scheme_name, scheme):
    assert os.path.exists(EXE), "run_dambreak not built (make -C hipims-ocl_amd/csrc)"
    cols, rows, duration, freq, batch = 320, 160, 1.5, 0.5, 25
    res = subprocess.run([EXE, str(cols), str(rows), str(duration), str(freq), scheme_name, str(batch)],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    lines = [l.split() for l in res.stdout.strip().splitlines()]
    assert len(lines) == 3
    ref = python_protocol(cols, rows, duration, freq, scheme, batch)
    assert len(ref) == 3
    for (t_py, sum_py), l in zip(ref, lines):
        t_cpp, sum_cpp = float(l[0]), float(l[4])
        assert abs(t_cpp - t_py) < 1e-9 and abs(t_cpp - round(t_cpp / freq) * freq) < 1e-9     # lands on the output times
        assert abs(sum_cpp - sum_py) <= 1e-9 * abs(sum_py)
    vols = [float(l[3]) for l in lines]
    if scheme == hp.SCHEME_GODUNOV:
        assert max(vols) - min(vols) < 1e-6 * vols[0]                 # closed basin
    assert "Mcell-steps/s" in res.stderr


def test_cpp_host_cell_boundary_matches_python_binding():
    """CSchemeMI::addBoundaryCell (CBoundaryCell's place) -> hp_boundary_add_cell: an inflow hydrograph raises the volume,
    and the C++ and Python hosts agree on every output time."""
    cols, rows, duration, freq, batch = 320, 160, 1.5, 0.5, 25
    res = subprocess.run([EXE, str(cols), str(rows), str(duration), str(freq), "godunov", str(batch), "inflow"],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    lines = [l.split() for l in res.stdout.strip().splitlines()]
    ref = python_protocol(cols, rows, duration, freq, hp.SCHEME_GODUNOV, batch, inflow=True)
    assert len(lines) == len(ref) == 3
    for (t_py, sum_py), l in zip(ref, lines):
        assert abs(float(l[0]) - t_py) < 1e-9
        assert abs(float(l[4]) - sum_py) <= 1e-9 * abs(sum_py)
    vols = [float(l[3]) for l in lines]
    assert vols[0] < vols[1] < vols[2]                              # water is coming in
    closed = subprocess.run([EXE, str(cols), str(rows), str(duration), str(freq), "godunov", str(batch)],
                            capture_output=True, text=True, timeout=300)
    v_closed = float(closed.stdout.strip().splitlines()[-1].split()[3])
    # "volume" adds Q*dt/dx^2 of depth to EACH listed cell (CLBoundaries.clc:67-98): the hydrograph integral to t=1.5 is
    # 0.5*(40+120)*1 + 120*0.5 = 140 m3 per cell, sampled at each step's start time (rectangle rule -> within 1 %)
    expected = 140.0 * len(range(rows // 4, 3 * rows // 4))
    assert abs((vols[-1] - v_closed) - expected) < 0.01 * expected


def test_cpp_host_automatic_queue_runs():
    res = subprocess.run([EXE, "512", "256", "1.0", "0.25"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    times = [float(l.split()[0]) for l in res.stdout.strip().splitlines()]
    assert np.allclose(times, [0.25, 0.5, 0.75, 1.0], atol=1e-9)


@pytest.mark.parametrize("scheme_name,world,peer_max", [("godunov", 2, 2), ("godunov", 3, 2), ("muscl", 3, 2), ("godunov", 3, 1), ("godunov", 3, 0)])
def test_cpp_host_strip_mode_matches_the_single_domain(scheme_name, world, peer_max):
    """CSchemeMI in strip mode (setStrip -> hp_strip_comm_init / hp_strip_step_batch / hp_strip_update_timestep): the C++
    host runs one scheme per row strip -- ranks as threads on the one GPU, collective library = the tests' in-process
    double -- through CModel's output-time loop; times, iteration counts, volume and level checksum of the strips
    together equal the single domain's."""
    exe = os.path.join(PKG, "lib", "run_strips")
    fake = os.path.join(os.path.dirname(__file__), "fake_rccl", "libfake_rccl.so")
    assert os.path.exists(exe), "run_strips not built (make -C hipims-ocl_amd/csrc)"
    if not os.path.exists(fake):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", "-o", fake,
                               os.path.join(os.path.dirname(fake), "fake_rccl.cpp")])
    cols, rows, duration, freq, batch = 320, 161, 1.5, 0.5, 25
    strips_run = subprocess.run([exe, fake, str(world), str(cols), str(rows), str(duration), str(freq), scheme_name, str(batch)],
                                capture_output=True, text=True, timeout=300,
                                env=dict(os.environ, HIPIMS_MI_PEER_MAX=str(min(peer_max, 1)), HP_PEER_DIRECT=str(int(peer_max == 2)), GPU_MAX_HW_QUEUES="16"))
    assert strips_run.returncode == 0, strips_run.stdout + strips_run.stderr
    # the maximum over the strips: through the peer-written mailboxes (CSchemeMI::getPeerTicket / connectPeers) or the all-reduce
    assert ("maximum over the strips: peer-written mailboxes" in strips_run.stderr) == bool(peer_max), strips_run.stderr
    assert ("ghost rows: written by the strips" in strips_run.stderr) == (peer_max == 2), strips_run.stderr
    single = subprocess.run([EXE, str(cols), str(rows), str(duration), str(freq), scheme_name, str(batch)],
                            capture_output=True, text=True, timeout=300)
    assert single.returncode == 0, single.stderr
    a = [l.split() for l in strips_run.stdout.strip().splitlines()]
    b = [l.split() for l in single.stdout.strip().splitlines()]
    assert len(a) == len(b) == 3
    for la, lb in zip(a, b):
        assert float(la[0]) == float(lb[0]) and int(la[1]) == int(lb[1])              # time, successful iterations
        assert abs(float(la[2]) - float(lb[3])) <= 1e-11 * float(lb[3])                 # volume (summation order differs)
        assert abs(float(la[3]) - float(lb[4])) <= 1e-12 * abs(float(lb[4]))            # checksum of Z


@pytest.mark.parametrize("scheme_name,world,peer_max,batch", [("godunov", 2, 2, 25), ("godunov", 3, 2, 25), ("muscl", 3, 2, 25), ("muscl", 2, 0, 25),
                                                              ("godunov", 3, 1, 25), ("godunov", 3, 0, 25), ("godunov", 3, 2, 0)])
def test_cpp_single_threaded_model_loop_drives_the_strips(scheme_name, world, peer_max, batch):
    """the reference runs each domain's batch on the scheme's own worker (CSchemeGodunov.cpp:1116-1139) so that
    CModel's ONE main thread can schedule every idle domain in turn and poll isRunning() (CModel.cpp:906-955, :1069-1108).
    CSchemeMI now has that worker: host/run_model_strips.cpp is that management loop -- assess, sync / outputs, schedule every
    idle strip, poll -- on a single thread over 2-3 strips of one grid, both transports (the strips' own stores and mailboxes;
    the collective library's send / receive and all-reduce), fixed batches and the several-domains automatic queue.  A blocking
    runSimulation would deadlock here (strip 0's batch waits for strip 1's, which the same thread has not scheduled yet).
    Times, iteration counts, volume and level checksum equal the single domain's."""
    exe = os.path.join(PKG, "lib", "run_model_strips")
    fake = os.path.join(os.path.dirname(__file__), "fake_rccl", "libfake_rccl.so")
    assert os.path.exists(exe), "run_model_strips not built (make -C hipims-ocl_amd/csrc)"
    if not os.path.exists(fake):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", "-o", fake,
                               os.path.join(os.path.dirname(fake), "fake_rccl.cpp")])
    cols, rows, duration, freq = 320, 161, 1.5, 0.5
    run = subprocess.run([exe, fake, str(world), str(cols), str(rows), str(duration), str(freq), scheme_name, str(batch)],
                         capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, HIPIMS_MI_PEER_MAX=str(min(peer_max, 1)), HP_PEER_DIRECT=str(int(peer_max == 2)), GPU_MAX_HW_QUEUES="16"))
    assert run.returncode == 0, run.stdout + run.stderr
    assert ("maximum over the strips: peer-written mailboxes" in run.stderr) == bool(peer_max), run.stderr
    assert ("ghost rows: written by the strips" in run.stderr) == (peer_max == 2), run.stderr
    assert "no strip driven from a thread of its own" in run.stderr
    a = [l.split() for l in run.stdout.strip().splitlines()]
    if batch:
        single = subprocess.run([EXE, str(cols), str(rows), str(duration), str(freq), scheme_name, str(batch)],
                                capture_output=True, text=True, timeout=300)
        assert single.returncode == 0, single.stderr
        b = [[l.split()[0], l.split()[1], l.split()[3], l.split()[4]] for l in single.stdout.strip().splitlines()]
    else:       # the several-domains queue formula has no single-domain twin: the thread-per-strip driver is the comparison
        other = subprocess.run([os.path.join(PKG, "lib", "run_strips"), fake, str(world), str(cols), str(rows), str(duration), str(freq), scheme_name, "0"],
                               capture_output=True, text=True, timeout=300, env=dict(os.environ, GPU_MAX_HW_QUEUES="16"))
        assert other.returncode == 0, other.stdout + other.stderr
        b = [l.split() for l in other.stdout.strip().splitlines()]
    assert len(a) == len(b) == 3
    for la, lb in zip(a, b):
        assert float(la[0]) == float(lb[0]) and int(la[1]) == int(lb[1])              # time, successful iterations
        assert abs(float(la[2]) - float(lb[2])) <= 1e-11 * float(lb[2])                 # volume (summation order differs)
        assert abs(float(la[3]) - float(lb[3])) <= 1e-12 * abs(float(lb[3]))            # checksum of Z


def test_cpp_host_strip_mode_automatic_queue_is_rank_consistent():
    """The automatic batch size of a strip is the reference's several-domains formula (CSchemeGodunov.cpp:1428-1429: the
    iterations left to the target at the batch's mean timestep) -- simulated quantities only, so every rank queues the same
    batches (a disagreement would hang the collectives: the run must simply finish) and the result does not depend on the
    number of strips."""
    exe = os.path.join(PKG, "lib", "run_strips")
    fake = os.path.join(os.path.dirname(__file__), "fake_rccl", "libfake_rccl.so")
    if not os.path.exists(fake):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", "-o", fake,
                               os.path.join(os.path.dirname(fake), "fake_rccl.cpp")])
    outs = []
    for world in (2, 3):
        r = subprocess.run([exe, fake, str(world), "320", "161", "1.5", "0.5", "godunov", "0"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append([l.split() for l in r.stdout.strip().splitlines()])
    assert len(outs[0]) == len(outs[1]) == 3
    for la, lb in zip(*outs):
        assert float(la[0]) == float(lb[0]) and int(la[1]) == int(lb[1]) and int(la[1]) > 0
        assert abs(float(la[3]) - float(lb[3])) <= 1e-12 * abs(float(lb[3]))
