"""GPU parity tests of the whole V-cycle driver (mgx_solve2d / mgx_plan_*)
against the CPU oracle and the reference's known answers: Sol and the full
ResNorms history must match bit for bit."""
import numpy as np
import pytest

from conftest import assert_bits

pytestmark = pytest.mark.gpu


# 1 = one kernel per reference operator, 2 = fused streaming kernels
SCHEDS = [1, 2]


def _case9():
    m = 9
    f = np.zeros((m, m), order="F")
    for i in range(1, m - 1):
        for j in range(1, m - 1):
            f[i, j] = 1 + i + 10 * j
    return m, 1.0 / 8, f


@pytest.mark.parametrize("sched", SCHEDS)
def test_known_answers_9(mg, known, sched):
    K = known["case9"]["multigrid_5_3_3_0.6_2"]
    m, h, f = _case9()
    v = np.zeros((m, m), order="F")
    out = mg.solve_host(f, v, h, 5, 3, 3, 0.6, 2, schedule=sched)
    assert list(out.ResidualNorms) == [float(x) for x in K["ResNorms"]]
    assert v[4, 4] == float(K["Sol(4,4)"])
    assert out.Times[0] == 0 and np.all(np.diff(out.Times) >= 0)


@pytest.mark.parametrize("sched", SCHEDS)
def test_known_answers_257(mg, O, known, f_smooth_257, sched):
    K = known["case257"]
    M = 257
    v = np.zeros((M, M), order="F")
    out = mg.solve_host(f_smooth_257, v, 1.0 / (M - 1), 30, 2, 2, 0.6, 2, schedule=sched)
    rn = out.ResidualNorms
    assert list(rn[:6]) == [float(x) for x in K["ResNorms[0..5]"]]
    assert rn[30] == float(K["ResNorms[30]"])
    assert int(np.argmax(rn <= 1e-8)) == K["first_cycle_le_1e-8"]
    assert v.max() == float(K["max(Sol)"])
    sol, orn, _ = O.multigrid(f_smooth_257, np.zeros((M, M), order="F"), 1.0 / (M - 1), 30, 2, 2, 0.6, 2)
    assert_bits(v, sol, "Sol 257")
    assert_bits(rn, orn, "ResNorms 257")


@pytest.mark.parametrize("sched", SCHEDS)
@pytest.mark.parametrize("m,Nv,nd,nu,omega", [
    (3, 3, 2, 2, 0.6), (5, 4, 2, 2, 0.6), (9, 5, 1, 2, 0.7), (17, 5, 2, 1, 0.8), (33, 8, 2, 2, 0.6),
    (65, 6, 3, 3, 0.6), (129, 6, 5, 5, 0.9), (257, 5, 7, 7, 0.7), (513, 5, 2, 2, 0.6), (1025, 4, 2, 2, 0.6),
    (129, 4, 0, 2, 0.6), (129, 4, 2, 0, 0.6), (65, 3, 0, 0, 0.6), (257, 4, 3, 2, 0.6), (257, 4, 1, 1, 0.6),
])
def test_solver_matches_oracle_random_rhs(mg, O, m, Nv, nd, nu, omega, sched):
    from multigrid_amd import rhs
    f = rhs.random_np(m)
    dx = 1.0 / (m - 1)
    rng = np.random.default_rng(m)
    v0 = np.zeros((m, m), order="F")
    # non-zero Dirichlet data and a non-zero interior guess: the API takes both
    v0[0, :], v0[-1, :], v0[:, 0], v0[:, -1] = rng.standard_normal((4, m))
    v0[1:-1, 1:-1] = rng.standard_normal((m - 2, m - 2)) * 0.01
    v0[0, 1] = v0[2, 0] = v0[-1, -1] = v0[-1, 1] = v0[1, -1] = -0.0  # main.cpp:341 turns these into +0.0
    sol, rn, _ = O.multigrid(f, v0, dx, Nv, nd, nu, omega, 2)
    v = v0.copy(order="F")
    out = mg.solve_host(f, v, dx, Nv, nd, nu, omega, 2, schedule=sched)
    assert_bits(out.ResidualNorms, rn, "ResNorms m=%d" % m)
    assert_bits(v, sol, "Sol m=%d" % m)


@pytest.mark.parametrize("sched", SCHEDS)
@pytest.mark.parametrize("m,coarsest", [(257, 4), (129, 8), (65, 1), (33, 3), (5, 4), (97, 6), (385, 6), (1537, 6),
                                        (193, 3), (257, 128), (513, 64)])
def test_other_coarsest_values(mg, O, m, coarsest, sched):
    # coarsest != 2 either ends on a grid larger than 3x3, where the reference's
    # direct solver does nothing (main.cpp:167), or changes the depth (main.cpp:286).
    from multigrid_amd import rhs
    f = rhs.smooth_np(m)
    dx = 1.0 / (m - 1)
    v0 = np.zeros((m, m), order="F")
    try:
        sol, rn, _ = O.multigrid(f, v0, dx, 4, 2, 2, 0.6, coarsest)
    except ValueError:
        with pytest.raises(mg.MgxError):
            mg.solve_host(f, v0.copy(order="F"), dx, 4, 2, 2, 0.6, coarsest, schedule=sched)
        return
    v = v0.copy(order="F")
    out = mg.solve_host(f, v, dx, 4, 2, 2, 0.6, coarsest, schedule=sched)
    assert_bits(out.ResidualNorms, rn, "ResNorms")
    assert_bits(v, sol, "Sol")


@pytest.mark.parametrize("sched", SCHEDS)
def test_zero_cycles_returns_the_initial_norm_only(mg, O, sched):
    from multigrid_amd import rhs
    m = 33
    f = rhs.random_np(m)
    v0 = np.asfortranarray(np.random.default_rng(1).standard_normal((m, m)))
    v = v0.copy(order="F")
    out = mg.solve_host(f, v, 1.0 / 32, 0, 2, 2, 0.6, 2, schedule=sched)
    assert out.ResidualNorms.shape == (1,) and out.Times.tolist() == [0.0]
    assert out.ResidualNorms[0] == O.calc_norm(O.residual(v0, f, 1.0 / 32))
    assert_bits(v, v0, "v untouched")


@pytest.mark.parametrize("sched", SCHEDS)
def test_pure_jacobi(mg, O, sched):
    from multigrid_amd import rhs
    m = 65
    f = rhs.random_np(m)
    v0 = np.zeros((m, m), order="F")
    sol, rn, _ = O.multigrid(f, v0, 1.0 / 64, 5, 2, 2, 0.6, 2, pure_jacobi=True)
    v = v0.copy(order="F")
    out = mg.solve_host(f, v, 1.0 / 64, 5, 2, 2, 0.6, 2, pureJacobi=True, schedule=sched)
    assert_bits(out.ResidualNorms, rn, "ResNorms")
    assert_bits(v, sol, "Sol")


@pytest.mark.parametrize("sched", SCHEDS)
def test_device_resident_api_and_repeated_calls(mg, O, sched):
    """MultiGrid() on GPU tensors; 3 + 2 cycles through a Plan equal one 5-cycle call."""
    from multigrid_amd import rhs
    m = 129
    dx = 1.0 / (m - 1)
    f = rhs.random_np(m)
    sol, rn, _ = O.multigrid(f, np.zeros((m, m), order="F"), dx, 5, 2, 2, 0.6, 2)
    prob = mg.gridtype(mg.to_device(f), mg.colmajor(m, m), dx)
    out = mg.MultiGrid(prob, 5, 2, 2, 0.6, 2, schedule=sched)
    assert_bits(out.ResidualNorms, rn, "ResNorms")
    assert_bits(mg.to_numpy(prob.v), sol, "Sol")
    plan = mg.Plan(m, dx, 2, schedule=sched)
    assert plan.depth == O.depth(m, 2) and plan.level_m(1) == 65
    plan.set(mg.to_device(f), mg.colmajor(m, m))
    a = plan.cycles(3, 2, 2, 0.6)
    b = plan.cycles(2, 2, 2, 0.6)
    assert_bits(np.concatenate([a.ResidualNorms, b.ResidualNorms[1:]]), rn, "split ResNorms")
    assert a.ResidualNorms[-1] == b.ResidualNorms[0]
    assert_bits(mg.to_numpy(plan.get()), sol, "split Sol")
    plan.close()


@pytest.mark.parametrize("sched", SCHEDS)
def test_large_4097_against_oracle(mg, O, sched):
    """4097^2, V(2,2) (the single-GPU validation size), 2 cycles: the oracle needs a few seconds."""
    from multigrid_amd import rhs
    m = 4097
    dx = 1.0 / (m - 1)
    f = rhs.random_np(m)
    sol, rn, _ = O.multigrid(f, np.zeros((m, m), order="F"), dx, 2, 2, 2, 0.6, 2)
    prob = mg.gridtype(rhs.random_torch(m), mg.colmajor(m, m), dx)
    out = mg.MultiGrid(prob, 2, 2, 2, 0.6, 2, schedule=sched)
    assert_bits(out.ResidualNorms, rn, "ResNorms 4097")
    assert_bits(mg.to_numpy(prob.v), sol, "Sol 4097")


def test_schedules_agree_at_16385(mg):
    """The benchmark size 16385^2: too big for the oracle in seconds, so the two
    independent GPU schedules (one kernel per operator vs fused) must agree
    bit for bit, and the residual history must contract."""
    import torch
    from multigrid_amd import rhs
    m = 16385
    dx = 1.0 / (m - 1)
    f = rhs.random_torch(m)
    outs = []
    sols = []
    for sched in (mg.SCHEDULE_STEPWISE, mg.SCHEDULE_FUSED):
        prob = mg.gridtype(f, mg.colmajor(m, m), dx)
        outs.append(mg.MultiGrid(prob, 3, 2, 2, 0.6, 2, schedule=sched))
        sols.append(prob.v)
    assert_bits(outs[0].ResidualNorms, outs[1].ResidualNorms, "ResNorms 16385")
    assert torch.equal(sols[0].t().contiguous().view(torch.int64), sols[1].t().contiguous().view(torch.int64))
    rn = outs[0].ResidualNorms
    # (a random right-hand side lets the infinity norm rise in the first cycle)
    assert rn[1] > rn[2] > rn[3] and rn[3] < 0.5 * rn[0]


# MGTester.m:22-35,144 sweeps omega, the sweep counts and the size on a random right-hand side
@pytest.mark.parametrize("omega", [0.7, 0.8, 0.9])
@pytest.mark.parametrize("nd,nu", [(1, 2), (2, 1), (3, 3), (5, 5), (7, 7)])
def test_mgtester_parameter_sweep(mg, O, omega, nd, nu):
    from multigrid_amd import rhs
    for m in (129, 257, 513):
        f = rhs.random_np(m)
        dx = 1.0 / (m - 1)
        v0 = np.zeros((m, m), order="F")
        sol, rn, _ = O.multigrid(f, v0, dx, 4, nd, nu, omega, 2)
        v = v0.copy(order="F")
        out = mg.solve_host(f, v, dx, 4, nd, nu, omega, 2)
        assert_bits(out.ResidualNorms, rn, "ResNorms m=%d" % m)
        assert_bits(v, sol, "Sol m=%d" % m)


@pytest.mark.parametrize("sched", SCHEDS)
def test_plan_on_a_stream_of_the_callers(mg, O, sched):
    """A plan created under a non-default stream launches everything there."""
    import torch
    from multigrid_amd import rhs
    m = 513
    dx = 1.0 / (m - 1)
    f = rhs.random_np(m)
    sol, rn, _ = O.multigrid(f, np.zeros((m, m), order="F"), dx, 4, 2, 2, 0.6, 2)
    side = torch.cuda.Stream()
    fd = mg.to_device(f)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        plan = mg.Plan(m, dx, 2, schedule=sched)
        plan.set(fd, mg.colmajor(m, m))
        out = plan.cycles(4, 2, 2, 0.6)
        got = plan.get()
    side.synchronize()
    assert_bits(out.ResidualNorms, rn, "ResNorms")
    assert_bits(mg.to_numpy(got), sol, "Sol")
    plan.close()


@pytest.mark.parametrize("sched", SCHEDS)
@pytest.mark.parametrize("bad", [np.nan, np.inf, -np.inf])
def test_non_finite_right_hand_side(mg, O, sched, bad):
    """calcNorm ignores NaNs (DTDoubleArray.cpp:883-899,923-939 compare with `<` and `>`) and
    the residual array's zero boundary always takes part, so a run that drowns in NaNs
    reports norm 0, not NaN; infinities win the comparison.  The GPU must follow suit."""
    from multigrid_amd import rhs
    for m, Nv in ((65, 3), (257, 1), (513, 2)):
        dx = 1.0 / (m - 1)
        f = rhs.random_np(m)
        f[20, 31] = bad
        v0 = np.zeros((m, m), order="F")
        sol, rn, _ = O.multigrid(f, v0, dx, Nv, 2, 2, 0.6, 2)
        v = v0.copy(order="F")
        out = mg.solve_host(f, v, dx, Nv, 2, 2, 0.6, 2, schedule=sched)
        assert_bits(out.ResidualNorms, rn, "ResNorms m=%d" % m)
        # NaN payloads are not part of the contract: same NaN positions, same bits elsewhere
        assert np.array_equal(np.isnan(v), np.isnan(sol))
        ok = ~np.isnan(sol)
        assert np.array_equal(v[ok].view(np.uint64), sol[ok].view(np.uint64))


@pytest.mark.parametrize("sched", SCHEDS)
def test_negative_zero_boundary_values(mg, O, sched):
    """`cur += refined` (main.cpp:341) runs over the whole array, boundary included, and the
    boundary of `refined` is +0.0: a Dirichlet value of -0.0 becomes +0.0 in the first cycle --
    but not with zero cycles and not in pureJacobi mode, which never get there."""
    m = 33
    rng = np.random.default_rng(4)
    f = np.asfortranarray(rng.standard_normal((m, m)))
    v0 = np.asfortranarray(rng.standard_normal((m, m)) * 0.0)  # a mix of +0.0 and -0.0 everywhere
    assert np.signbit(v0[0, :]).any() and np.signbit(v0[:, -1]).any()
    for Nv, nd, nu, pj in ((2, 2, 2, False), (1, 5, 3, False), (0, 2, 2, False), (2, 2, 2, True), (1, 0, 0, False)):
        sol, rn, _ = O.multigrid(f, v0, 1.0 / (m - 1), Nv, nd, nu, 0.7, 2, pure_jacobi=pj)
        v = v0.copy(order="F")
        out = mg.solve_host(f, v, 1.0 / (m - 1), Nv, nd, nu, 0.7, 2, pureJacobi=pj, schedule=sched)
        assert_bits(out.ResidualNorms, rn, "ResNorms %r" % ((Nv, nd, nu, pj),))
        assert_bits(v, sol, "Sol %r" % ((Nv, nd, nu, pj),))


@pytest.mark.parametrize("k,d", [(5, 10), (3, 11), (7, 10), (9, 10), (11, 10), (3, 12), (7, 11), (13, 10)])
def test_schedules_agree_at_large_sizes_that_are_not_powers_of_two(mg, k, d):
    """Sizes k*2^d+1 between 5000 and 15000 (coarsest = k: the hierarchy ends on a (k+1)^2 grid):
    the big level-0 launches run in their capped, tuned shapes with ragged last strips and
    chunks; the one-kernel-per-operator schedule is the independent implementation to match."""
    import torch
    from multigrid_amd import rhs
    m = k * 2 ** d + 1
    dx = 1.0 / (m - 1)
    f = rhs.random_torch(m)
    res = []
    for sched in (mg.SCHEDULE_STEPWISE, mg.SCHEDULE_FUSED):
        prob = mg.gridtype(f, mg.colmajor(m, m), dx)
        out = mg.MultiGrid(prob, 15, 2, 2, 0.6, k, schedule=sched)  # enough launches for every candidate shape
        res.append((out.ResidualNorms, prob.v))
    assert_bits(res[0][0], res[1][0], "ResNorms m=%d" % m)
    assert torch.equal(res[0][1].view(torch.int64), res[1][1].view(torch.int64))


def test_host_arrays_through_the_staged_transfers(mg, O):
    """mgx_solve2d on ordinary host arrays big enough (134 MB each) for the multi-threaded
    staged upload/download, with a non-zero guess and Dirichlet data: same bits as the oracle;
    with the zero-guess flag only the boundary of v is read."""
    from multigrid_amd import rhs
    m = 4097
    dx = 1.0 / (m - 1)
    f = rhs.random_np(m)
    rng = np.random.default_rng(5)
    v0 = np.asfortranarray(rng.standard_normal((m, m)) * 0.01)
    sol, rn, _ = O.multigrid(f, v0, dx, 2, 2, 2, 0.6, 2, omp=True)
    v = v0.copy(order="F")
    out = mg.solve_host(f, v, dx, 2, 2, 2, 0.6, 2)
    assert_bits(out.ResidualNorms, rn, "ResNorms")
    assert_bits(v, sol, "Sol")
    # zero guess stated: the interior of v is not read (garbage there must not matter)
    vz = np.zeros((m, m), order="F")
    vz[0, :], vz[-1, :], vz[:, 0], vz[:, -1] = rng.standard_normal((4, m))
    sol, rn, _ = O.multigrid(f, vz, dx, 2, 2, 2, 0.6, 2, omp=True)
    v = vz.copy(order="F")
    v[1:-1, 1:-1] = 123.0
    out = mg.solve_host(f, v, dx, 2, 2, 2, 0.6, 2, zero_guess=True)
    assert_bits(out.ResidualNorms, rn, "ResNorms, zero guess")
    assert_bits(v, sol, "Sol, zero guess")
    # zero guess found by looking (no flag): same result; a single -0.0 or a value in the
    # last interior entry makes it a guess with data, which is uploaded whole
    v = vz.copy(order="F")
    out = mg.solve_host(f, v, dx, 2, 2, 2, 0.6, 2)
    assert_bits(out.ResidualNorms, rn, "ResNorms, zero guess detected")
    assert_bits(v, sol, "Sol, zero guess detected")
    for val in (-0.0, 0.25):
        vd = vz.copy(order="F")
        vd[m - 2, m - 2] = val
        sol_d, rn_d, _ = O.multigrid(f, vd, dx, 1, 0, 1, 0.6, 2, omp=True)
        v = vd.copy(order="F")
        out = mg.solve_host(f, v, dx, 1, 0, 1, 0.6, 2)
        assert_bits(out.ResidualNorms, rn_d, "ResNorms, guess with one entry %r" % val)
        assert_bits(v, sol_d, "Sol, guess with one entry %r" % val)


def test_single_cycle_calls_carry_the_next_down_leg(mg, O):
    """Calls of one cycle each (a caller's own tolerance loop): from the third such call on, a
    call's closing launch also runs the next call's opening sweeps and restriction, and the next
    call starts from there.  Whatever the pattern of calls -- parameters changed in between, the
    iterate read, replaced or committed, zero-cycle calls, full multigrid in between -- histories
    and solutions are the oracle's, bit for bit."""
    M = 513
    dx = 1.0 / (M - 1)
    rng = np.random.default_rng(5)
    f = np.asfortranarray(rng.standard_normal((M, M)))
    v0 = np.zeros((M, M), order="F")
    v0[0, :], v0[-1, :], v0[:, 0], v0[:, -1] = rng.standard_normal((4, M))
    v0[0, 3] = v0[7, 0] = -0.0
    plan = mg.Plan(M, dx, 2)
    plan.set(mg.to_device(f), mg.to_device(v0))
    ref_v = v0
    hist = []

    def run(n, nd, nu, om):
        nonlocal ref_v
        out = plan.cycles(n, nd, nu, om)
        sol, rn, _ = O.multigrid(f, ref_v, dx, n, nd, nu, om, 2)
        assert_bits(out.ResidualNorms, rn, "ResNorms of call %d: %d x V(%d,%d)" % (len(hist), n, nd, nu))
        ref_v = sol
        hist.append((n, nd, nu))
        assert_bits(mg.to_numpy(plan.get()), ref_v, "Sol after call %d" % (len(hist) - 1))

    for _ in range(6):
        run(1, 2, 2, 0.6)            # calls 2.. carry
    run(0, 2, 2, 0.6)                # a zero-cycle call in between keeps what was carried
    run(1, 2, 2, 0.6)
    run(3, 2, 2, 0.6)                # a longer continuation
    run(1, 1, 2, 0.6)                # other sweep counts: the carried leg is dropped
    for _ in range(4):
        run(2, 1, 2, 0.6)
    run(1, 1, 2, 0.7)                # other omega
    for _ in range(4):
        run(1, 3, 1, 0.7)
    plan.set(mg.to_device(f), mg.to_device(ref_v))   # data replaced (same values): carried leg dropped
    for _ in range(4):
        run(1, 3, 1, 0.7)
    for _ in range(4):
        run(1, 1, 0, 0.6)            # no post-smoothing: the end-of-cycle iterate is stage 0 of the closing launch
    for _ in range(3):
        run(1, 0, 2, 0.6)            # no pre-smoothing: nothing to carry
    for _ in range(3):
        run(1, 2, 3, 0.6)            # more than four sweeps: not the lean schedule
    plan.close()
