"""GPU tests of the multi-rank slab path with the HIP backend: one process with
a single slab, and 2-3 processes that share the test box's one GPU and exchange
halos over gloo (RCCL needs one GPU per rank, which the test box does not have;
the launch sequence, ownership and kernels are the same)."""
import numpy as np
import pytest

from conftest import assert_bits
from test_slab_cpu import run_ranks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,tail_m,nd,nu", [(257, 33, 2, 2), (513, 65, 2, 2), (257, 33, 5, 6), (257, 65, 0, 2),
                                            (257, 65, 3, 0), (1025, 129, 2, 2)])
def test_single_slab_hip_matches_oracle(mg, O, M, tail_m, nd, nu):
    from multigrid_amd import rhs, slab
    dx = 1.0 / (M - 1)
    f = rhs.random_np(M)
    rng = np.random.default_rng(M)
    v0 = np.zeros((M, M), order="F")
    v0[0, :], v0[-1, :], v0[:, 0], v0[:, -1] = rng.standard_normal((4, M))
    S = slab.SlabSolver(M, dx, 2, tail_m=tail_m)
    S.set_problem(f, v0)
    out = S.cycles(3, nd, nu, 0.6)
    sol = S.gather_solution()
    S.close()
    ref_sol, ref_rn, _ = O.multigrid(f, v0, dx, 3, nd, nu, 0.6, 2)
    assert_bits(out.ResidualNorms, ref_rn, "ResNorms")
    assert_bits(sol, ref_sol, "Sol")


def test_slab_random_rhs_generator(mg):
    from multigrid_amd import rhs
    full = rhs.random_np(129)
    part = rhs.random_torch_cols(129, 40, 77).cpu().numpy()
    assert_bits(part, np.ascontiguousarray(full[:, 40:77].T), "random_torch_cols")
    edge = rhs.random_torch_cols(129, 0, 129).cpu().numpy()
    assert_bits(edge, np.ascontiguousarray(full.T), "random_torch_cols full")


@pytest.mark.parametrize("world,M,tail_m", [(2, 513, 65), (3, 1025, 129), (4, 2049, 257), (2, 1025, 65)])
def test_ranks_share_gpu_over_gloo(O, world, M, tail_m):
    codes, outs = run_ranks(world, ["hip", M, 3, 2, 2, 0.6, tail_m, 0], timeout=400)
    assert codes == [0] * world, "\n".join(outs)
    assert "resnorms_equal=True sol_equal=True" in outs[0]


def test_ranks_share_gpu_exact_coarse_even_coarsest(O):
    """705 = 11*2^6+1, coarsest 11: levels ... 23, 12; the tail solves the 12x12 grid exactly."""
    codes, outs = run_ranks(2, ["hip", 705, 4, 2, 2, 0.6, 89, 0, 11, 1], timeout=400)
    assert codes == [0, 0], "\n".join(outs)
    assert "resnorms_equal=True sol_equal=True" in outs[0]


@pytest.mark.parametrize("world,M,tail_m,ncyc", [(1, 513, 65, 1), (2, 513, 65, 1), (3, 1025, 129, 2), (2, 2049, 257, 1)])
def test_full_multigrid_over_slabs(O, world, M, tail_m, ncyc):
    """SlabSolver.fmg with the HIP backend (ranks share the GPU, gloo) against the oracle's composition."""
    codes, outs = run_ranks(world, ["hip", M, ncyc, 2, 2, 0.6, tail_m, 0, 2, 0, 1], timeout=400)
    assert codes == [0] * world, "\n".join(outs)
    assert "resnorms_equal=True sol_equal=True" in outs[0]


def test_rccl_group_of_one_rank_full_multigrid(O):
    codes, outs = run_ranks(1, ["hip-rccl", 1025, 1, 2, 2, 0.6, 129, 0, 2, 0, 1], timeout=400,
                            extra_env={"MGX_COMM_FORCE": "1"})
    assert codes == [0], "\n".join(outs)
    assert "resnorms_equal=True sol_equal=True" in outs[0]


def test_rccl_group_of_one_rank(O):
    """The RCCL calls of the multi-GPU path (process group bound to the device, all-gather
    into the tail plan's own buffer, max-reduction of the norm slots) on the one GPU there
    is: a group of a single rank with the collectives forced on."""
    codes, outs = run_ranks(1, ["hip-rccl", 1025, 3, 2, 2, 0.6, 129, 0], timeout=400, extra_env={"MGX_COMM_FORCE": "1"})
    assert codes == [0], "\n".join(outs)
    assert "resnorms_equal=True sol_equal=True" in outs[0]


def test_slab_launch_rejects_missing_halo(mg):
    import ctypes
    from multigrid_amd import _lib
    L = _lib.load()
    t = mg.colmajor(65, 20)
    o = _lib.mgx_slab_op()
    o.struct_size = ctypes.sizeof(_lib.mgx_slab_op)
    o.nsweeps, o.pro_stage, o.norm_stage, o.restrict_out, o.zero_in, o.store = 2, -1, -1, 0, 0, 1
    o.v_in, o.v_out, o.f = t.data_ptr(), mg.colmajor(65, 20).data_ptr(), t.data_ptr()
    o.m, o.ld, o.ncols = 65, 65, 20
    o.jlo, o.jhi, o.jfix_lo, o.jfix_hi = 1, 19, -1, -1   # needs 2 halo columns on each side, has 1
    o.dx, o.omega = 0.1, 0.6
    assert L.mgx_slab_launch(ctypes.byref(o), None) == -1
    assert b"halo columns needed" in L.mgx_last_error()
    o.jlo, o.jhi = 2, 18
    assert L.mgx_slab_launch(ctypes.byref(o), None) == 0


def test_slab_solver_under_a_stream_other_than_the_one_it_was_built_on(mg, O):
    """The tail plan follows torch's CURRENT stream at every call (mgx_plan_set_stream): a
    solver built on the default stream and run inside `torch.cuda.stream(side)` orders its
    tail after the gather that fills the tail's right-hand side and before the up leg that
    reads the tail's iterate -- same bits as the oracle, on either stream, back and forth."""
    import torch
    from multigrid_amd import rhs, slab
    M = 1025
    dx = 1.0 / (M - 1)
    f = rhs.random_np(M)
    v0 = np.zeros((M, M), order="F")
    ref_sol, ref_rn, _ = O.multigrid(f, v0, dx, 6, 2, 2, 0.6, 2)
    S = slab.SlabSolver(M, dx, 2, tail_m=129)      # built on the default stream
    S.set_problem(f, v0)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        a = S.cycles(2, 2, 2, 0.6)
    torch.cuda.current_stream().wait_stream(side)
    b = S.cycles(2, 2, 2, 0.6)                     # back on the default stream
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        c = S.cycles(2, 2, 2, 0.6)
        sol = S.gather_solution()
    side.synchronize()
    S.close()
    rn = np.concatenate([a.ResidualNorms, b.ResidualNorms[1:], c.ResidualNorms[1:]])
    assert_bits(rn, ref_rn, "ResNorms across streams")
    assert_bits(sol, ref_sol, "Sol across streams")


def test_plan_follows_the_current_stream(mg, O):
    """solver.Plan built on one stream, used under another (Plan.set/cycles/get call
    mgx_plan_set_stream with the stream that is current)."""
    import torch
    from multigrid_amd import rhs
    m = 513
    dx = 1.0 / (m - 1)
    f = rhs.random_np(m)
    sol, rn, _ = O.multigrid(f, np.zeros((m, m), order="F"), dx, 4, 2, 2, 0.6, 2)
    plan = mg.Plan(m, dx, 2)
    fd = mg.to_device(f)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        plan.set(fd, mg.colmajor(m, m))
        a = plan.cycles(2, 2, 2, 0.6)
    b = plan.cycles(2, 2, 2, 0.6)   # default stream again: ordered after the side stream's work
    got = plan.get()
    torch.cuda.synchronize()
    assert_bits(np.concatenate([a.ResidualNorms, b.ResidualNorms[1:]]), rn, "ResNorms")
    assert_bits(mg.to_numpy(got), sol, "Sol")
    plan.close()
