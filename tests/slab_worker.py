"""One rank of a multi-process SlabSolver test (launched by test_slab_cpu.py and
test_gpu_slab.py with RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set).

  python slab_worker.py <backend: numpy|hip|hip-rccl> <M> <Nv> <Ndown> <Nup> <omega> <tail_m> <pure_jacobi> [coarsest] [exact_coarse] [fmg]

Every rank builds the same global problem, solves it with the slab driver over
gloo and compares the gathered solution and the residual history with the CPU
oracle bit for bit.  Exit code 0 = identical.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    backend, M, Nv, nd, nu, omega, tail_m, pure = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]),
                                                   int(sys.argv[5]), float(sys.argv[6]), int(sys.argv[7]),
                                                   int(sys.argv[8]))
    coarsest = int(sys.argv[9]) if len(sys.argv) > 9 else 2
    exact = bool(int(sys.argv[10])) if len(sys.argv) > 10 else False
    fmg = bool(int(sys.argv[11])) if len(sys.argv) > 11 else False  # full multigrid, Nv cycles per level
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rccl = backend == "hip-rccl"  # one GPU per rank over RCCL (a group of ONE rank on a one-GPU box)
    grouped = world > 1 or rccl
    if rccl:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dev = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=int(os.environ.get("RANK", "0")), world_size=world,
                                device_id=torch.device("cuda", dev))
    elif world > 1:
        dist.init_process_group("gloo")
    rank = dist.get_rank() if grouped else 0

    from oracle import oracle as O
    from multigrid_amd import rhs, slab
    if backend == "numpy":
        from slab_numpy_backend import NumpyBackend
        be = NumpyBackend(O, coarse_exact=exact)
    elif rccl:
        be = slab.HipBackend(coarse_solve=int(exact))
    else:
        torch.cuda.set_device(0)  # every rank shares the one GPU of the test box
        be = slab.HipBackend(coarse_solve=int(exact))

    dx = 1.0 / (M - 1)
    f = rhs.random_np(M)
    rng = np.random.default_rng(M)
    v0 = np.zeros((M, M), order="F")
    v0[0, :], v0[-1, :], v0[:, 0], v0[:, -1] = rng.standard_normal((4, M))   # non-zero Dirichlet data
    big = os.environ.get("MGX_WORKER_LEAN_INPUT") == "1"  # large grids: zero interior guess, threaded oracle
    if not big:
        v0[1:-1, 1:-1] = rng.standard_normal((M - 2, M - 2)) * 0.01          # and a non-zero guess
    v0[0, 3] = v0[5, 0] = v0[-1, -1] = v0[-1, 7] = v0[M // 2, -1] = -0.0     # main.cpp:341 turns these into +0.0

    S = slab.SlabSolver(M, dx, coarsest, backend=be, tail_m=tail_m, halo=8)
    two_range = [0]
    if os.environ.get("MGX_WORKER_COUNT_TWO_RANGE") == "1":  # count the launches that produce two column ranges
        plain_launch = be.launch

        def counting_launch(op):
            if op.jhi2 and op.jhi2 > op.jlo2:
                two_range[0] += 1
            plain_launch(op)

        be.launch = counting_launch
    S.set_problem(f, v0)
    out = S.fmg(Nv, nd, nu, omega) if fmg else S.cycles(Nv, nd, nu, omega, pureJacobi=bool(pure))
    sol = S.gather_solution()
    S.close()

    ok = True
    if os.environ.get("MGX_WORKER_COUNT_TWO_RANGE") == "1":
        print("slab_worker: rank %d two_range_launches=%d" % (rank, two_range[0]), flush=True)
    if rank == 0:
        if fmg:
            ref_sol, ref_rn = O.fmg(f, v0, dx, Nv, nd, nu, omega, coarsest, coarse_exact=exact)
        else:
            ref_sol, ref_rn, _ = O.multigrid(f, v0, dx, Nv, nd, nu, omega, coarsest, pure_jacobi=bool(pure),
                                             coarse_exact=exact, omp=big)
        if exact and backend != "numpy":
            # the GPU's sine-transform solve agrees with the oracle's Cholesky to rounding only
            same_rn = bool(np.allclose(out.ResidualNorms, ref_rn, rtol=1e-6, atol=1e-9 * ref_rn[0]))
            same_sol = bool(np.abs(sol - ref_sol).max() <= 1e-9 * np.abs(ref_sol).max())
        else:
            same_rn = np.array_equal(out.ResidualNorms.view(np.uint64), ref_rn.view(np.uint64))
            same_sol = np.array_equal(np.ascontiguousarray(sol).view(np.uint64),
                                      np.ascontiguousarray(ref_sol).view(np.uint64))
        ok = same_rn and same_sol
        print("slab_worker: world=%d backend=%s M=%d ndist=%d resnorms_equal=%s sol_equal=%s"
              % (world, backend, M, S.ndist, same_rn, same_sol), flush=True)
        if not same_rn:
            print(out.ResidualNorms, ref_rn, flush=True)
        if not same_sol:
            bad = np.argwhere(sol != ref_sol)
            print("first mismatches [i, j]:", bad[:5].tolist(), "of", len(bad), flush=True)
    if grouped:
        flag = torch.tensor([0 if ok else 1], device="cuda" if rccl else "cpu")
        dist.all_reduce(flag)
        ok = int(flag.item()) == 0
        dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
