"""Host cost per V(2,2) cycle of the three drivers on a grid small enough that the GPU is
mostly idle (default 2049^2): the single-device plan, the partitioned plan in C++
(mgx_mplan_*, R slabs sharing this GPU, one host thread each) and the Python slab driver
(one rank).  Wall time per cycle of a 200-cycle call next to the device time the same call
reports; when wall >> device the driver is host-bound.
   python tools/host_overhead.py [M] [cycles]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import multigrid_amd as mg
from multigrid_amd import rhs, slab

M = int(sys.argv[1]) if len(sys.argv) > 1 else 2049
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dx = 1.0 / (M - 1)


def timed(run):
    run(5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = run(K)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / K
    return wall * 1e3, out.Times[-1] / K * 1e3


plan = mg.Plan(M, dx, 2)
plan.set(rhs.random_torch(M), mg.colmajor(M, M))
w, d = timed(lambda n: plan.cycles(n, 2, 2, 0.6))
print("M=%d  single-device plan        : wall %.3f ms/cycle, device %.3f ms/cycle" % (M, w, d), flush=True)
plan.close()
for R in (1, 2, 4, 8):
    try:
        P = mg.MultiPlan(M, dx, 2, nslabs=R, devices=[0] * R, tail_m=max(257, (M - 1) // 8 + 1))
    except mg.MgxError as e:
        print("R=%d: %s" % (R, e))
        continue
    P.set_random_rhs()
    w, d = timed(lambda n: P.cycles(n, 2, 2, 0.6))
    enq = P.stats()[3] / K * 1e3
    print("M=%d  C++ partitioned plan R=%d  : wall %.3f ms/cycle, device(slab 0) %.3f ms/cycle, host enqueue "
          "(slab 0's thread) %.3f ms/cycle, tail from %d^2 (on ONE GPU every slab's copy of the tail runs here)"
          % (M, R, w, d, enq, P.tail_m), flush=True)
    P.close()
S = slab.SlabSolver(M, dx, 2, tail_m=max(257, (M - 1) // 8 + 1))
S.set_random_rhs()
w, d = timed(lambda n: S.cycles(n, 2, 2, 0.6))
print("M=%d  Python slab driver, 1 rank : wall %.3f ms/cycle, device %.3f ms/cycle" % (M, w, d), flush=True)
S.close()
