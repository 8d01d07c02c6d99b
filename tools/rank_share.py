"""What ONE rank of an R-rank run computes per cycle, measured on one GPU.

The slab driver (multigrid_amd/slab.py) is built for rank `g` of `R` with a stand-in
communicator that moves nothing, so every launch of that rank's share -- the level-0 edge
and interior launches, the distributed coarser levels with their recomputed halo columns,
the replicated tail -- runs with the real geometry while nothing waits for a neighbour.
The numbers are therefore the compute part of a cycle at R GPUs (values are meaningless:
the halo columns never arrive).

   python tools/rank_share.py [--size 16385] [--ranks 8] [--rank 3] [--cycles 20] [--tail-m 0]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from multigrid_amd import slab


class QuietComm:
    """Rank g of R that exchanges nothing."""

    def __init__(self, rank, size):
        self.rank, self.size = rank, size
        self.on, self.backend, self.solo, self.group = False, "none", False, None
        self.nexchanges = self.ngathers = 0

    def exchange(self, items):
        self.nexchanges += 1

    def exchange_start(self, items):
        self.nexchanges += 1
        return None

    def exchange_finish(self, handle):
        pass

    def allgather(self, t, out=None):
        self.ngathers += 1
        return out if out is not None else t

    def allreduce_max_(self, t):
        return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=16385)
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--rank", type=int, default=3)
    ap.add_argument("--cycles", type=int, default=20)
    ap.add_argument("--tail-m", type=int, default=0)
    ap.add_argument("--nd", type=int, default=2)
    ap.add_argument("--nu", type=int, default=2)
    args = ap.parse_args()
    M = args.size
    slab._Comm = lambda group: QuietComm(args.rank, args.ranks)
    S = slab.SlabSolver(M, 1.0 / (M - 1), 2, tail_m=args.tail_m or None)
    S.set_random_rhs()
    S.tune(args.nd, args.nu, 0.6)
    S.cycles(3, args.nd, args.nu, 0.6)
    S.profile(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = S.cycles(args.cycles, args.nd, args.nu, 0.6)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / args.cycles * 1e3
    prof = S.profile_totals()
    pts = S.profile_points()
    L0 = S.levels[0]
    res = {"M": M, "ranks": args.ranks, "rank": args.rank, "ndist": S.ndist, "tail_m": S.dims[S.ndist],
           "own_cols": L0.b1 - L0.b0, "edge_cols": None, "wall_ms_per_cycle": wall,
           "device_ms_per_cycle": float(out.Times[-1]) / args.cycles * 1e3, "level0": {}}
    l0 = 0.0
    for k, (cnt, ms) in sorted(prof.items()):
        p = pts.get(k, 0) / max(cnt, 1)
        us = ms / cnt * 1e3
        # bytes a launch must move per point: up+down (and its edge band) 28, down and up 26, norm 16
        bpp = {"k_stream:norm": 16.0, "k_stream:down": 26.0, "k_stream:up": 26.0}.get(k, 28.0)
        res["level0"][k] = {"launches": cnt, "avg_us": us, "points": p, "bytes_per_point": bpp,
                            "frac_of_peak": bpp * p / (us * 1e-6) / 8e12 if us > 0 else None}
        l0 += ms
    res["level0_ms_per_cycle"] = l0 / args.cycles
    res["rest_ms_per_cycle"] = res["device_ms_per_cycle"] - res["level0_ms_per_cycle"]
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
