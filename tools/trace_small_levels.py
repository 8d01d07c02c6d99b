"""Timeline INSIDE the launches of the small levels (k_tail, k_mid_down, k_mid_up).

Needs a build of the library with -DMGX_TAIL_TRACE (mgx_tail.hip) and -DMGX_MID_TRACE
(mgx_mid.hip): thread 0 of one workgroup then leaves the time of a 100 MHz clock at every
step of the launch (tools/build_trace.sh builds variants/libmgx_trace.so).

   MGX_LIB=$PWD/variants/libmgx_trace.so python tools/trace_small_levels.py [M] [rows of a grouped launch's level 0 ...]

M: size of the problem (default 2049); the further arguments pick the grouped launches by the
size of their finest level (default 1025 513 129)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import multigrid_amd as mg
from multigrid_amd import _lib, rhs


def show(title, marks, names):
    ev = sorted((int(marks[k]), k) for k in names if marks[k] != 0)
    if not ev:
        print("%s: no marks" % title)
        return
    t0 = prev = ev[0][0]
    print(title)
    for ts, k in ev:
        print("   %-28s +%6.2f us   (%5.2f)" % (names[k], (ts - t0) / 100.0, (ts - prev) / 100.0))
        prev = ts


def main():
    L = _lib.load()
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 2049
    tops = [int(a) for a in sys.argv[2:]] or [1025, 513, 129]
    plan = mg.Plan(M, 1.0 / (M - 1), 2)
    plan.set(rhs.random_torch(M), mg.colmajor(M, M))
    plan.cycles(5, 2, 2, 0.6)
    torch.cuda.synchronize()
    if hasattr(L, "mgx_debug_tail_trace"):
        buf = (ctypes.c_ulonglong * 128)()
        L.mgx_debug_tail_trace.argtypes = [ctypes.c_void_p]
        L.mgx_debug_tail_trace(buf)
        names = {0: "entry", 1: "LDS cleared", 2: "right-hand side in LDS", 3: "levels done", 4: "stored"}
        for l in range(6):
            for k, what in enumerate(["enter", "down sweeps", "residual", "restriction", "back from below", "prolongation",
                                      "up sweeps"]):
                names[10 * l + 10 + k] = "level %d %s" % (l, what)
        show("k_tail", np.array(list(buf), dtype=np.int64), names)
    if hasattr(L, "mgx_debug_mid_trace"):
        L.mgx_debug_mid_trace.argtypes = [ctypes.c_int, ctypes.c_void_p]
        for top in tops:
            L.mgx_debug_mid_trace(top, None)
            plan.cycles(3, 2, 2, 0.6)
            torch.cuda.synchronize()
            buf = (ctypes.c_ulonglong * 128)()
            L.mgx_debug_mid_trace(-1, buf)
            t = np.array(list(buf), dtype=np.int64).reshape(2, 64)
            dn = {0: "entry"}
            up = {0: "entry", 1: "loads issued, coarse v in LDS", 60: "stored"}
            for k in range(5):
                for j, what in enumerate(["right-hand side in registers", "first sweep stored", "sweeps", "residual",
                                          "restriction"]):
                    dn[10 * k + 1 + j] = "level %d %s" % (k, what)
                for j, what in enumerate(["begin", "first sweep (+ correction)", "further down sweeps", "up sweeps"]):
                    up[10 * k + 2 + j] = "level %d %s" % (k, what)
            show("k_mid_down from %d" % top, t[0], dn)
            show("k_mid_up to %d" % top, t[1], up)


if __name__ == "__main__":
    main()
