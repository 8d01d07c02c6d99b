"""GPU tests of the EXTENSION mgx_plan_fmg (full multigrid).  The reference has no FMG
driver, so there is no reference output to pin; the oracle composes its restatements of
the reference's operators (coarsen, refine, direct_solve, MultiGrid) the same way, and
because every operator is bit-exact the whole driver must be too.  Independent of that,
FMG must land on the fixed point of the V-cycle iteration."""
import numpy as np
import pytest

from conftest import assert_bits

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sched", [1, 2])
@pytest.mark.parametrize("M,ncyc,nd,nu,coarsest", [(3, 1, 2, 2, 2), (5, 2, 2, 2, 2), (9, 1, 1, 1, 2), (33, 1, 2, 2, 2),
                                                   (129, 1, 2, 2, 2), (129, 2, 3, 3, 2), (257, 1, 2, 1, 2),
                                                   (513, 1, 2, 2, 2), (1025, 2, 2, 2, 2), (257, 1, 5, 5, 2),
                                                   (97, 2, 2, 2, 3), (257, 1, 0, 2, 2)])
def test_fmg_matches_the_same_composition_on_the_cpu(mg, O, M, ncyc, nd, nu, coarsest, sched):
    from multigrid_amd import rhs
    f = rhs.random_np(M)
    dx = 1.0 / (M - 1)
    rng = np.random.default_rng(M)
    v0 = np.asfortranarray(rng.standard_normal((M, M)))  # the interior is ignored, the boundary is data
    sol, rn = O.fmg(f, v0, dx, ncyc, nd, nu, 0.6, coarsest)
    prob = mg.gridtype(mg.to_device(f), mg.to_device(v0), dx)
    out = mg.FullMultiGrid(prob, ncyc, nd, nu, 0.6, coarsest, schedule=sched)
    assert_bits(out.ResidualNorms, rn, "ResNorms")
    assert_bits(mg.to_numpy(prob.v), sol, "Sol")


def test_fmg_then_more_cycles_reach_the_vcycle_fixed_point(mg, O, f_smooth_257):
    M = 257
    dx = 1.0 / (M - 1)
    ref, rn, _ = O.multigrid(f_smooth_257, np.zeros((M, M), order="F"), dx, 40, 2, 2, 0.6, 2)
    plan = mg.Plan(M, dx, 2)
    plan.set(mg.to_device(f_smooth_257), mg.colmajor(M, M))
    a = plan.fmg(1, 2, 2, 0.6)
    one = mg.to_numpy(plan.get())
    # one FMG pass is already far closer than one V-cycle from zero ...
    v1, _, _ = O.multigrid(f_smooth_257, np.zeros((M, M), order="F"), dx, 1, 2, 2, 0.6, 2)
    assert np.abs(one - ref).max() < 0.02 * np.abs(v1 - ref).max()
    # ... and plain cycles from there converge to the same discrete solution
    b = plan.cycles(30, 2, 2, 0.6)
    assert b.ResidualNorms[0] == a.ResidualNorms[-1]
    assert np.abs(mg.to_numpy(plan.get()) - ref).max() <= 1e-11 * np.abs(ref).max()
    plan.close()


def test_fmg_with_exact_coarsest_solve(mg, O):
    from multigrid_amd import rhs
    M = 353  # levels 353, 177, 89, 45, 23, 12
    dx = 1.0 / (M - 1)
    f = rhs.random_np(M)
    v0 = np.zeros((M, M), order="F")
    sol, rn = O.fmg(f, v0, dx, 2, 2, 2, 0.6, 11, coarse_exact=True)
    prob = mg.gridtype(mg.to_device(f), mg.to_device(v0), dx)
    out = mg.FullMultiGrid(prob, 2, 2, 2, 0.6, 11, coarse_solve=mg.COARSE_EXACT)
    assert np.allclose(out.ResidualNorms, rn, rtol=1e-6, atol=1e-9 * rn[0])
    assert np.abs(mg.to_numpy(prob.v) - sol).max() <= 1e-9 * np.abs(sol).max()


def test_fmg_plus_vcycles_at_32769_schedules_agree(mg):
    """FMG + V-cycles on 32769^2 (the largest benchmark configuration): far too big for the CPU oracle in
    seconds, so the two independent GPU schedules must agree bit for bit, FMG must start the
    finest level well below the zero-guess residual, and the V-cycles after it must contract."""
    import torch
    from multigrid_amd import rhs
    M = 32769
    dx = 1.0 / (M - 1)
    f = rhs.random_torch(M)
    res = []
    for sched in (mg.SCHEDULE_STEPWISE, mg.SCHEDULE_FUSED):
        plan = mg.Plan(M, dx, 2, schedule=sched)
        plan.set(f, mg.colmajor(M, M))
        zero_guess_norm = plan.cycles(0, 2, 2, 0.6).ResidualNorms[0]
        a = plan.fmg(1, 2, 2, 0.6)
        b = plan.cycles(3, 2, 2, 0.6)
        res.append((a.ResidualNorms, b.ResidualNorms, plan.get().clone()))
        plan.close()
        assert a.ResidualNorms[1] < 0.2 * zero_guess_norm
        assert b.ResidualNorms[0] == a.ResidualNorms[1]
        assert b.ResidualNorms[3] < 0.1 * b.ResidualNorms[0]
    assert_bits(res[0][0], res[1][0], "FMG ResNorms")
    assert_bits(res[0][1], res[1][1], "V-cycle ResNorms")
    assert torch.equal(res[0][2].view(torch.int64), res[1][2].view(torch.int64))
