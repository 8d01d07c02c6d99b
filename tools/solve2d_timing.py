"""End-to-end time of mgx_solve2d on ordinary host arrays (the call main.cpp:445 becomes):
16385^2, 10 V(2,2) cycles, with and without the zero-guess flag, next to the device time of
the cycles alone.   python tools/solve2d_timing.py [M] [Nv]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import multigrid_amd as mg
from multigrid_amd import rhs

M = int(sys.argv[1]) if len(sys.argv) > 1 else 16385
Nv = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dx = 1.0 / (M - 1)
f = rhs.random_np(M)
ref = None
for label, kw in (("first call (page-locks the staging buffers)", {"zero_guess": True}), ("default", {}),
                  ("zero-guess flag", {"zero_guess": True}), ("default", {}), ("zero-guess flag", {"zero_guess": True})):
    v = np.zeros((M, M), order="F")
    t0 = time.perf_counter()
    out = mg.solve_host(f, v, dx, Nv, 2, 2, 0.6, 2, **kw)
    dt = time.perf_counter() - t0
    if ref is None:
        ref = v.copy(order="F")
    same = np.array_equal(v.view(np.uint64), ref.view(np.uint64))
    print("%-55s total %7.1f ms, cycles %6.1f ms, same bits: %s" % (label, dt * 1e3, out.Times[-1] * 1e3, same), flush=True)
