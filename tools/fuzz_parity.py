"""Randomised parity sweep (not part of the test suite): random hierarchy shapes, sweep
counts, omega, schedules, boundary data and cycle counts against the CPU oracle, bit for bit.
   python tools/fuzz_parity.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multigrid_amd as mg
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
n = 0
while time.time() < t_end:
    k = int(rng.integers(1, 12))
    d = int(rng.integers(1, 9))
    M = k * 2 ** d + 1
    if M < 5 or M > 1800 or M % 2 == 0:
        continue
    coarsest = int(rng.choice([1, 2, 2, 2, 3, 4, 6, 11, k, 2 * k]))
    nd, nu = int(rng.integers(0, 8)), int(rng.integers(0, 8))
    omega = float(rng.choice([0.5, 0.6, 0.7, 0.8, 0.9, 1.0]))
    Nv = int(rng.integers(0, 5))
    sched = int(rng.choice([1, 2]))
    pj = bool(rng.integers(0, 8) == 0)
    f = np.asfortranarray(rng.standard_normal((M, M)))
    v0 = np.asfortranarray(rng.standard_normal((M, M)) * float(rng.choice([0.0, 1.0])))
    dx = float(rng.choice([1.0, 1.0 / (M - 1), 0.37]))
    try:
        sol, rn, _ = O.multigrid(f, v0, dx, Nv, nd, nu, omega, coarsest, pure_jacobi=pj)
    except ValueError:
        try:
            mg.solve_host(f, v0.copy(order="F"), dx, Nv, nd, nu, omega, coarsest, pureJacobi=pj, schedule=sched)
        except mg.MgxError:
            continue
        print("MISMATCH: oracle refused, library accepted", M, coarsest)
        sys.exit(1)
    v = v0.copy(order="F")
    out = mg.solve_host(f, v, dx, Nv, nd, nu, omega, coarsest, pureJacobi=pj, schedule=sched)
    same = np.array_equal(out.ResidualNorms.view(np.uint64), rn.view(np.uint64)) and \
        np.array_equal(v.view(np.uint64), sol.view(np.uint64))
    n += 1
    if not same:
        print("MISMATCH", dict(M=M, coarsest=coarsest, nd=nd, nu=nu, omega=omega, Nv=Nv, sched=sched, pj=pj, dx=dx))
        print("resnorms gpu   ", out.ResidualNorms.tolist())
        print("resnorms oracle", rn.tolist())
        bad = np.argwhere(v.view(np.uint64) != sol.view(np.uint64))
        print("sol mismatches", len(bad), bad[:8].tolist())
        for (i, j) in bad[:4]:
            print("  ", i, j, repr(v[i, j]), repr(sol[i, j]))
        print("v0 zero?", not v0.any(), "max|v0|", np.abs(v0).max())
        np.savez("gpurun_out/fuzz_case.npz", f=f, v0=v0)
        sys.exit(1)
print("fuzz: %d random cases bit-identical to the oracle" % n)
