"""Replay one case of tests/test_gpu_fuzz_strict.py with different batch sizes and report where engine and oracle part."""
import importlib.util, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hipims-ocl_amd"), os.path.join(ROOT, "tests")]
os.environ.setdefault("HIPIMS_MI_NO_TORCH", "1")
import hipims_mi as hp, oracle
spec = importlib.util.spec_from_file_location("fz", os.path.join(ROOT, "tests", "test_gpu_fuzz_strict.py"))
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
seed = int(sys.argv[1])
for batch in (1, 2, 5, 50):
    c = m.make_case(seed)
    oq = c["quirks"] & ~(oracle.Q6_MUSCL_SERIAL if c["scheme"] == hp.SCHEME_MUSCL_HANCOCK else 0)
    ref = oracle.OracleSim(c["cols"], c["rows"], dx=c["dx"], scheme=c["scheme"], precision=c["precision"], quirks=oq, friction=c["kw"]["friction"],
                           dynamic_dt=c["kw"]["dynamic_dt"], fixed_dt=c["fixed_dt"], dt_initial=c["fixed_dt"] if not c["kw"]["dynamic_dt"] else 0.001)
    dom = hp.Domain(c["cols"], c["rows"], dx=c["dx"], scheme=c["scheme"], precision=c["precision"], quirks=oracle.quirks_to_engine(c["quirks"]), friction=c["kw"]["friction"],
                    dynamic_dt=c["kw"]["dynamic_dt"], dt_fixed=c["fixed_dt"], dt_initial=c["fixed_dt"] if not c["kw"]["dynamic_dt"] else 0.001, math_mode=hp.MATH_STRICT)
    for s in (ref, dom):
        s.upload(c["st"], c["bed"], c["man"]); m.attach(s, c["bdy"])
    dom.set_target_time(c["target"]); ref.set_target(c["target"])
    done, first = 0, None
    while done < sum(c["cuts"]) and first is None:
        ref.run(batch); dom.step_batch(batch); done += batch
        a, b = dom.download(), ref.download()
        if not np.array_equal(a, b):
            bad = np.argwhere((a != b).any(axis=-1))
            sr, sc = ref.scalars(), dom.read_scalars()
            first = (done, len(bad), bad[:4].tolist(), [(a[y, x].tolist(), b[y, x].tolist()) for y, x in bad[:2]], sr["t"], sr["t_hydro"], sc["time_hydrological"])
    print("fused", dom.boundaries_fused(), "batch", batch, "first mismatch:", first, flush=True)
    dom.close()
