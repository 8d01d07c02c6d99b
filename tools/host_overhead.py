"""Host time the slab driver spends per cycle (what bounds a rank when its GPU share is small).

Rank g of R is built with a communicator that moves nothing (tools/rank_share.py), on a grid
small enough that the device finishes each launch long before the host has enqueued the next
one: the wall time per cycle is then the host's.  A second figure stubs the launches out
altogether (Python only).

   python tools/host_overhead.py [--size 4097] [--ranks 8] [--rank 3] [--cycles 200]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from multigrid_amd import slab
from rank_share import QuietComm


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=4097)
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--rank", type=int, default=3)
    ap.add_argument("--cycles", type=int, default=200)
    ap.add_argument("--tail-m", type=int, default=257)
    args = ap.parse_args()
    M = args.size
    slab._Comm = lambda group: QuietComm(args.rank, args.ranks)
    S = slab.SlabSolver(M, 1.0 / (M - 1), 2, tail_m=args.tail_m)
    S.set_random_rhs()
    S.tune(2, 2, 0.6)
    S.cycles(3, 2, 2, 0.6)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = S.cycles(args.cycles, 2, 2, 0.6)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / args.cycles * 1e6
    dev = float(out.Times[-1]) / args.cycles * 1e6
    nlaunch = [0]
    real = S.be.launch

    def stub(op):
        nlaunch[0] += 1

    S.be.launch = stub
    t0 = time.perf_counter()
    S.cycles(args.cycles, 2, 2, 0.6)
    torch.cuda.synchronize()
    py = (time.perf_counter() - t0) / args.cycles * 1e6
    S.be.launch = real
    print(json.dumps({"M": M, "ranks": args.ranks, "rank": args.rank, "ndist": S.ndist,
                      "wall_us_per_cycle": wall, "device_us_per_cycle": dev,
                      "python_only_us_per_cycle": py, "launches_per_cycle": nlaunch[0] / args.cycles}, indent=1))


if __name__ == "__main__":
    main()
