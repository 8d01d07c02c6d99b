"""Randomised parity sweep: random hierarchy shapes, sweep counts, omega, schedules, boundary
data and cycle counts against the CPU oracle, bit for bit.  The test suite runs a small budget
of the partitioned-plan cases (tests/test_gpu_mplan.py); long sweeps are run by hand.
   python tools/fuzz_parity.py [seconds] [seed] [largest size]
   FUZZ_MODES=3,4  only these kinds of case (0 full multigrid, 1 slab driver, 2 exact coarsest
                   solve, 3/4 partitioned plan, 5 a series of short calls on one plan, others: one call)
   FUZZ_CASES=200  stop after this many cases"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multigrid_amd as mg
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
MAXM = int(sys.argv[3]) if len(sys.argv) > 3 else 1800
t_end = time.time() + budget
t_say = time.time() + 30
n = 0
only = [int(x) for x in os.environ["FUZZ_MODES"].split(",")] if os.environ.get("FUZZ_MODES") else None
most = int(os.environ.get("FUZZ_CASES", "0"))
while time.time() < t_end and not (most and n >= most):
    if time.time() > t_say:
        print("... %d cases so far" % n, flush=True)
        t_say = time.time() + 30
    k = int(rng.integers(1, 12))
    d = int(rng.integers(1, 13 if MAXM > 2000 else 9))
    M = k * 2 ** d + 1
    if M < 5 or M > MAXM or M % 2 == 0:
        continue
    coarsest = int(rng.choice([1, 2, 2, 2, 3, 4, 6, 11, k, 2 * k]))
    nd, nu = int(rng.integers(0, 8)), int(rng.integers(0, 8))
    if os.environ.get("FUZZ_LEAN"):  # the schedules with at most four sweeps per cycle (grouped launches, up+down fusion)
        nd = int(rng.integers(0, 5))
        nu = int(rng.integers(0, 5 - nd))
    omega = float(rng.choice([0.5, 0.6, 0.7, 0.8, 0.9, 1.0]))
    Nv = int(rng.integers(0, 5))
    sched = int(rng.choice([1, 2]))
    pj = bool(rng.integers(0, 8) == 0)
    f = np.asfortranarray(rng.standard_normal((M, M)))
    v0 = np.asfortranarray(rng.standard_normal((M, M)) * float(rng.choice([0.0, 1.0])))
    dx = float(rng.choice([1.0, 1.0 / (M - 1), 0.37]))
    mode = int(rng.integers(0, 10))
    if only is not None:
        mode = int(rng.choice(only))
    if os.environ.get("FUZZ_LOG"):  # one line per case, so that a hang names its case
        with open(os.environ["FUZZ_LOG"], "a") as fh:
            fh.write("case %d: %r\n" % (n, dict(M=M, coarsest=coarsest, nd=nd, nu=nu, omega=omega, Nv=Nv, sched=sched,
                                                pj=pj, dx=dx, mode=mode)))
    if mode == 0 and not pj:  # full multigrid
        try:
            sol, rn = O.fmg(f, v0, dx, Nv, nd, nu, omega, coarsest)
        except ValueError:
            continue
        prob = mg.gridtype(mg.to_device(f), mg.to_device(v0), dx)
        out = mg.FullMultiGrid(prob, Nv, nd, nu, omega, coarsest, schedule=sched)
        v = mg.to_numpy(prob.v)
        n += 1
        if not (np.array_equal(out.ResidualNorms.view(np.uint64), rn.view(np.uint64)) and
                np.array_equal(v.view(np.uint64), sol.view(np.uint64))):
            print("FMG MISMATCH", dict(M=M, coarsest=coarsest, nd=nd, nu=nu, omega=omega, ncyc=Nv, sched=sched, dx=dx))
            sys.exit(1)
        continue
    if mode == 1 and M >= 65:  # the slab driver on one rank, V-cycles or full multigrid
        from multigrid_amd import slab
        tail_m = int(rng.choice([17, 33, 65, 129, 257, 513]))
        use_fmg = bool(rng.integers(0, 2)) and not pj
        try:
            if use_fmg:
                sol, rn = O.fmg(f, v0, dx, Nv, nd, nu, omega, coarsest)
            else:
                sol, rn, _ = O.multigrid(f, v0, dx, Nv, nd, nu, omega, coarsest, pure_jacobi=pj)
        except ValueError:
            continue
        try:
            S = slab.SlabSolver(M, dx, coarsest, schedule=sched, tail_m=tail_m)
        except ValueError:
            continue
        S.set_problem(f, v0)
        out = S.fmg(Nv, nd, nu, omega) if use_fmg else S.cycles(Nv, nd, nu, omega, pureJacobi=pj)
        v = S.gather_solution()
        S.close()
        n += 1
        if not (np.array_equal(out.ResidualNorms.view(np.uint64), rn.view(np.uint64)) and
                np.array_equal(np.ascontiguousarray(v).view(np.uint64), np.ascontiguousarray(sol).view(np.uint64))):
            print("SLAB MISMATCH", dict(M=M, coarsest=coarsest, nd=nd, nu=nu, omega=omega, Nv=Nv, sched=sched, pj=pj, dx=dx,
                                        tail_m=tail_m, fmg=use_fmg, ndist=S.ndist))
            sys.exit(1)
        continue
    if mode in (3, 4) and M >= 65:  # the partitioned plan in one process: random slab count, split calls
        R = int(rng.integers(1, 9))
        tail_m = int(rng.choice([17, 33, 65, 129, 257, 513]))
        use_fmg = bool(rng.integers(0, 3) == 0) and not pj
        try:
            if use_fmg:
                sol, rn = O.fmg(f, v0, dx, Nv, nd, nu, omega, coarsest)
            else:
                sol, rn, _ = O.multigrid(f, v0, dx, Nv, nd, nu, omega, coarsest, pure_jacobi=pj)
        except ValueError:
            continue
        try:
            P = mg.MultiPlan(M, dx, coarsest, nslabs=R, devices=[0] * R, tail_m=tail_m)
        except mg.MgxError:
            continue  # too few columns per slab
        P.upload(f, v0)
        if use_fmg:
            if rng.integers(0, 2):
                P.cycles(1, nd, nu, omega)   # counters and staging parities not at their initial values
                P.upload(f, v0)
            out = P.fmg(Nv, nd, nu, omega)
            v = P.download()
            P.close()
            n += 1
            if not (np.array_equal(out.ResidualNorms.view(np.uint64), rn.view(np.uint64)) and
                    np.array_equal(v.view(np.uint64), sol.view(np.uint64))):
                print("MPLAN FMG MISMATCH", dict(M=M, coarsest=coarsest, nd=nd, nu=nu, omega=omega, Nv=Nv, dx=dx,
                                                 tail_m=tail_m, R=R))
                sys.exit(1)
            continue
        k1 = int(rng.integers(0, Nv + 1))
        if os.environ.get("FUZZ_LOG"):
            with open(os.environ["FUZZ_LOG"], "a") as fh:
                fh.write("   mplan R=%d tail_m=%d k1=%d ndist=%d\n" % (R, tail_m, k1, P.ndist))
        a = P.cycles(k1, nd, nu, omega, pureJacobi=pj)
        b = P.cycles(Nv - k1, nd, nu, omega, pureJacobi=pj)
        v = P.download()
        ndist = P.ndist
        P.close()
        got = np.concatenate([a.ResidualNorms, b.ResidualNorms[1:]])
        n += 1
        if not (np.array_equal(got.view(np.uint64), rn.view(np.uint64)) and
                np.array_equal(v.view(np.uint64), sol.view(np.uint64))):
            print("MPLAN MISMATCH", dict(M=M, coarsest=coarsest, nd=nd, nu=nu, omega=omega, Nv=Nv, k1=k1, pj=pj, dx=dx,
                                         tail_m=tail_m, R=R, ndist=ndist))
            sys.exit(1)
        continue
    if mode == 5 and not pj:  # one plan driven by a series of short calls (what a call leaves for the next one)
        try:
            O.multigrid(f, v0, dx, 0, nd, nu, omega, coarsest)
        except ValueError:
            continue
        try:
            plan = mg.Plan(M, dx, coarsest, schedule=sched)
        except mg.MgxError:
            continue
        plan.set(mg.to_device(f), mg.to_device(v0))
        ref_v, ok, log = v0, True, []
        cur = (nd, nu, omega)
        for call in range(int(rng.integers(3, 12))):
            r = rng.random()
            if r < 0.15:
                cur = (int(rng.integers(0, 5)), int(rng.integers(0, 4)), float(rng.choice([0.6, 0.8])))
            ncyc = int(rng.choice([0, 1, 1, 1, 2, 3]))
            out = plan.cycles(ncyc, cur[0], cur[1], cur[2])
            sol, rn, _ = O.multigrid(f, ref_v, dx, ncyc, cur[0], cur[1], cur[2], coarsest)
            ref_v = sol
            log.append((ncyc,) + cur)
            ok = ok and np.array_equal(out.ResidualNorms.view(np.uint64), rn.view(np.uint64))
            r = rng.random()
            if r < 0.3:
                ok = ok and np.array_equal(mg.to_numpy(plan.get()).view(np.uint64), ref_v.view(np.uint64))
                log.append("get")
            elif r < 0.4:
                plan.set(mg.to_device(f), mg.to_device(ref_v))
                log.append("set")
            if not ok:
                break
        ok = ok and np.array_equal(mg.to_numpy(plan.get()).view(np.uint64), ref_v.view(np.uint64))
        plan.close()
        n += 1
        if not ok:
            print("SERIES MISMATCH", dict(M=M, coarsest=coarsest, sched=sched, dx=dx), log)
            sys.exit(1)
        continue
    if mode == 2 and not pj:  # the exact coarsest solve (an extension): agreement to rounding, not bitwise
        try:
            sol, rn, _ = O.multigrid(f, v0, dx, Nv, nd, nu, omega, coarsest, coarse_exact=True)
        except ValueError:
            continue
        v = v0.copy(order="F")
        try:
            out = mg.solve_host(f, v, dx, Nv, nd, nu, omega, coarsest, schedule=sched, coarse_solve=mg.COARSE_EXACT)
        except mg.MgxError as e:
            if "exact coarsest-grid solve takes at most" in str(e):
                continue
            raise
        n += 1
        scale = max(np.abs(sol).max(), 1e-300)
        if not (np.abs(v - sol).max() <= 1e-8 * scale and np.allclose(out.ResidualNorms, rn, rtol=1e-5, atol=1e-8 * max(rn[0], 1e-300))):
            print("EXACT-COARSE MISMATCH", dict(M=M, coarsest=coarsest, nd=nd, nu=nu, omega=omega, Nv=Nv, sched=sched, dx=dx),
                  np.abs(v - sol).max() / scale, out.ResidualNorms.tolist(), rn.tolist())
            sys.exit(1)
        continue
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
