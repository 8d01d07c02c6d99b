"""Host-side enqueue time of one slab-driver cycle (the launches are asynchronous, so the
driver's per-cycle wall clock is the time the host needs to issue a cycle)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multigrid_amd as mg
from multigrid_amd import slab

M = int(sys.argv[1]) if len(sys.argv) > 1 else 16385
tail_m = int(sys.argv[2]) if len(sys.argv) > 2 else 1025
S = slab.SlabSolver(M, 1.0 / (M - 1), 2, tail_m=tail_m)
S.set_random_rhs()
S.cycles(3, 2, 2, 0.6)
torch.cuda.synchronize()
t0 = time.perf_counter()
out = S.cycles(50, 2, 2, 0.6)
wall = time.perf_counter() - t0
print("M=%d ndist=%d tail=%d: wall %.3f ms/cycle, host enqueue %.3f ms/cycle" % (M, S.ndist, S.dims[S.ndist], wall * 1e3 / 50, out.Times[-1] * 1e3 / 50))
