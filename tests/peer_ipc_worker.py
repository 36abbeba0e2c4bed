"""Worker of tests/test_gpu_strips.py::test_peer_mailboxes_between_processes: RANKS THAT ARE PROCESSES (sharing the one GPU
of the box) connect their mailboxes through IPC handles and run reductions of known values through them -- the part of the
peer-written maximum that the thread-rank tests cannot reach (there the ranks share an address space).
usage: peer_ipc_worker.py <rank> <world> <directory the tickets travel through>"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hipims-ocl_amd")]
os.environ["HIPIMS_MI_NO_TORCH"] = "1"
import hipims_mi as hp  # noqa: E402

rank, world, where = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
dom = hp.Domain(64, 64)
ticket = dom.strip_peer_ticket()
with open(os.path.join(where, f"ticket.{rank}.tmp"), "wb") as f:
    f.write(ticket)
os.rename(os.path.join(where, f"ticket.{rank}.tmp"), os.path.join(where, f"ticket.{rank}"))
deadline = time.time() + 120
tickets = []
for r in range(world):
    path = os.path.join(where, f"ticket.{r}")
    while not os.path.exists(path):
        if time.time() > deadline:
            print("FAILED: no ticket from rank", r, flush=True); sys.exit(3)
        time.sleep(0.01)
    tickets.append(open(path, "rb").read())
assert tickets[rank] == ticket and len(set(tickets)) == world
active = dom.strip_peer_connect(tickets, rank)
if not active:
    print("FAILED: mailboxes not connected on rank", rank, flush=True); sys.exit(4)
bad = 0
for i in range(200):                                      # both mailbox sets, many times over; the largest value moves around the ranks
    mine = float((rank * 7 + i * 3) % 11) + 0.125 * rank - 3.0
    want = max(float((r * 7 + i * 3) % 11) + 0.125 * r - 3.0 for r in range(world))
    got = dom.strip_peer_round(mine)
    bad += got != want
t0 = time.perf_counter()
for i in range(500):
    dom.strip_peer_round(1.0)
per_round_us = (time.perf_counter() - t0) / 500 * 1e6     # launch + wait + 16-byte read-back, host-synchronous
dom.strip_peer_disconnect()
dom.close()
print("rank", rank, "of", world, "wrong maxima", bad, "host-synchronous round %.1f us" % per_round_us, flush=True)
sys.exit(0 if bad == 0 else 1)
