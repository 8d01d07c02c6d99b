"""65537^2 (4.3 G points: every flat index needs 64 bits; 183 GB of HBM): the fused and the
one-kernel-per-operator schedules must still agree bit for bit.  Not part of the test suite.
   python -u tools/huge_grid_check.py [M]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import multigrid_amd as mg
from multigrid_amd import _lib, rhs
from multigrid_amd.slab import _device_view

M = int(sys.argv[1]) if len(sys.argv) > 1 else 65537
L = _lib.load()
res = []
for sched in (mg.SCHEDULE_FUSED, mg.SCHEDULE_STEPWISE):
    plan = mg.Plan(M, 1.0 / (M - 1), 2, schedule=sched)
    print("schedule", sched, "plan holds %.1f GB" % (L.mgx_plan_bytes(plan._h) / 1e9), flush=True)
    f = _device_view(L.mgx_plan_f_dev(plan._h), M, torch.device("cuda"))
    for j0 in range(0, M, 2048):
        j1 = min(M, j0 + 2048)
        f[j0:j1] = rhs.random_torch_cols(M, j0, j1)
    torch.cuda.synchronize()
    _lib.check(L.mgx_plan_commit(plan._h))
    out = plan.cycles(3, 2, 2, 0.6)
    v = _device_view(L.mgx_plan_v_dev(plan._h), M, torch.device("cuda"))
    # order-independent 64-bit fingerprints of the solution, column block by column block
    fp = []
    for j0 in range(0, M, 8192):
        blk = v[j0:min(M, j0 + 8192)].view(torch.int64)
        fp.append((int(blk.sum().item()), int((blk >> 7).sum().item())))
    print("  ResNorms", out.ResidualNorms.tolist(), flush=True)
    res.append((out.ResidualNorms, fp))
    plan.close()
    del f, v
    torch.cuda.empty_cache()
ok = np.array_equal(res[0][0].view(np.uint64), res[1][0].view(np.uint64)) and res[0][1] == res[1][1]
print("M=%d: schedules agree bit for bit: %s; norms contract: %s" % (M, ok, bool(res[0][0][3] < 0.5 * res[0][0][1])))
sys.exit(0 if ok else 1)
