"""Launched by torch.distributed.run from tests/test_gpu_strips.py: one rank of a strip-decomposed run with the HIP
engine, several ranks sharing GPU 0, exchange staged through host memory over gloo (rehearsal transport)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "hipims-ocl_amd"))

from hipims_mi import strips, synthetic as syn  # noqa: E402


def main():
    out, scheme, cols, rows, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    st, bed, man = syn.s_rough(cols, rows, manning=None)
    r = strips.StripRunner(cols, rows, scheme=scheme, rank=rank, world=world, device=0, backend="gloo")
    r.upload_global(st, bed, man)
    r.set_target_time(2.0)                     # a sync point inside the run: skipped iterations on every rank
    r.step(steps)
    r.barrier()
    full = r.gather_owned()
    sc = r.engine.scalars()
    # bench.py's guard for N > 1 runs (StripRunner.verify_ghost_rows): clean after the run, and it notices ONE cell of one
    # ghost row that is not what its owner holds
    clean = r.verify_ghost_rows()
    if rank == 0:
        n = r.local_rows_total
        row = r.engine.domain.download(row0=n - 1, nrows=1)
        row[0, cols // 2, 0] += 1e-9
        r.engine.domain.upload_rows(row, n - 1)
    spoiled = r.verify_ghost_rows()
    if rank == 0:
        np.savez(out, state=full, t=sc["t"], dt=sc["dt"], ok=sc["batch_ok"], skipped=sc["batch_skipped"], clean=clean, spoiled=spoiled)
    r.close()


if __name__ == "__main__":
    main()
