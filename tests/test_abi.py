"""The C-ABI library: loads without a GPU, exports exactly what include/hipims_mi.h declares, and refuses
to run on the CPU (no compute call is made here)."""
import ctypes as C
import os
import re

import pytest

import hipims_mi
from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "hipims_mi.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hp_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = hipims_mi.load_library()
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in hipims_mi.h but not exported"
    assert sorted(hipims_mi.EXPORTS) == names


def test_abi_version_and_defaults():
    lib = hipims_mi.load_library()
    assert lib.hp_abi_version() == 2
    d = hipims_mi.DomainDesc()
    lib.hp_domain_desc_default(C.byref(d))
    # reference defaults: CScheme.cpp:46-55, CSchemeGodunov.cpp:56
    assert d.struct_size == C.sizeof(hipims_mi.DomainDesc)
    assert (d.courant, d.dt_initial, d.dry_threshold) == (0.5, 0.001, 1e-10)
    assert d.friction == 1 and d.dynamic_dt == 1 and d.precision == 8
    assert d.quirks == hipims_mi.QUIRKS_REFERENCE


def test_no_silent_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(hipims_mi.HipimsError, match="no HIP device|failed"):
        hipims_mi.Domain(16, 16)


def test_descriptor_validation_precedes_device_use():
    lib = hipims_mi.load_library()
    d = hipims_mi.DomainDesc()
    lib.hp_domain_desc_default(C.byref(d))
    d.struct_size = 4
    h = C.c_void_p()
    assert lib.hp_domain_create(C.byref(d), C.byref(h)) == -1
    assert b"size mismatch" in lib.hp_last_error()


def test_log_sink_receives_failures_with_the_reference_error_level():
    # model::doError(..., kLevelModelStop) is where the reference sends such failures (main.cpp:631-652)
    lib = hipims_mi.load_library()
    seen = []
    hipims_mi.set_log_sink(lambda level, text: seen.append((level, text)))
    try:
        d = hipims_mi.DomainDesc()
        lib.hp_domain_desc_default(C.byref(d))
        d.struct_size = 4
        h = C.c_void_p()
        assert lib.hp_domain_create(C.byref(d), C.byref(h)) == -1
    finally:
        hipims_mi.set_log_sink(None)
    assert len(seen) == 1 and seen[0][0] == 2 and "size mismatch" in seen[0][1]
    assert lib.hp_domain_create(C.byref(d), C.byref(h)) == -1          # sink removed: no further callbacks
    assert len(seen) == 1


def test_comm_load_failure_is_an_error_code_not_a_crash():
    # dlerror() was called twice (the second call returns NULL) and the message was built from NULL.
    # Run in a child process: a regression here is a segfault, which must fail this test, not end the session.
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import hipims_mi as hp; lib = hp.load_library();"
            "rc = lib.hp_comm_load(b'/nonexistent/dir/librccl.so'); msg = lib.hp_last_error().decode();"
            "print(rc, msg); sys.exit(0 if rc == -4 and 'cannot load the RCCL library' in msg and 'nonexistent' in msg else 1)"
            % os.path.join(ROOT, "hipims-ocl_amd"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, HIPIMS_MI_NO_TORCH="1"))
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr[-2000:])


def test_strip_info_without_a_domain_reports_no_communicator():
    lib = hipims_mi.load_library()
    info = hipims_mi.StripInfo()
    assert lib.hp_strip_info(None, C.byref(info)) == 0 and info.comm_ranks == -1
    assert lib.hp_strip_info(None, None) == -1


def test_peer_transport_entry_points_reject_missing_domains_and_match_the_header():
    """hp_strip_peer_*: argument errors are error codes (no GPU needed), and the binding's ticket size and strip-info layout
    are the header's."""
    import re
    lib = hipims_mi.load_library()
    buf = C.create_string_buffer(hipims_mi.PEER_TICKET_BYTES)
    active, out = C.c_int(7), C.c_double(0.0)
    assert lib.hp_strip_peer_ticket(None, buf) < 0
    assert lib.hp_strip_peer_connect(None, buf, 1, 0, C.byref(active)) < 0
    assert lib.hp_strip_peer_round(None, 1.0, C.byref(out)) < 0
    assert lib.hp_strip_peer_disconnect(None) == 0                      # like hp_strip_comm_destroy(NULL): nothing to do
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "hipims_mi.h")).read()
    assert int(re.search(r"#define HP_PEER_TICKET_BYTES (\d+)", header).group(1)) == hipims_mi.PEER_TICKET_BYTES
    assert C.sizeof(hipims_mi.StripInfo) == 256 + 6 * 4
    assert [f[0] for f in hipims_mi.StripInfo._fields_][-2:] == ["peer_max", "peer_halo"]


def test_bench_counts_gpus_without_the_hip_runtime(monkeypatch):
    """the launcher must stay GPU-free; torch.cuda.device_count() may open /dev/kfd.  visible_gpus() reads the KFD
    topology and the *_VISIBLE_DEVICES variables only (and never imports torch)."""
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    had_torch = "torch" in sys.modules
    n = bench.visible_gpus()
    assert isinstance(n, int) and n >= 0 and (("torch" in sys.modules) == had_torch)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpus() == 0
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0")
    assert bench.visible_gpus() <= 1


def test_bench_never_runs_a_smaller_job_than_asked_for():
    """`python bench.py --gpus 8` without a launcher's environment used to run N = 1 and print
    "n_gpus": 1.  It now starts the ranks itself -- and where the GPUs are not there it must refuse."""
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs are present: the self-launch would really run")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "HIPIMS_MI_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0
    assert '"n_gpus"' not in r.stdout and "refusing" in r.stderr
    # a launcher's environment that disagrees with --gpus is an error too, before anything touches a GPU
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and '"n_gpus"' not in r.stdout and "WORLD_SIZE=1" in r.stderr
