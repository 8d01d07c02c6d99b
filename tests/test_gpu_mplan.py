"""GPU tests of the partitioned plan (mgx_mplan_*, the ngpus option of mgx_solve2d): the
V-cycle over N j-slabs driven from one process.  The test box has one GPU, so the slabs
share it (devices may repeat an ordinal): the schedule, ownership, halo pulls, tail gather
and the thread/event hand-shakes are the ones a multi-GPU node runs, the copies are
device-to-device instead of peer-to-peer.  Sol and ResNorms must equal the single-process
CPU oracle bit for bit for every number of slabs."""
import numpy as np
import pytest

from conftest import assert_bits

pytestmark = pytest.mark.gpu


def _problem(M, zero_guess=False):
    from multigrid_amd import rhs
    f = rhs.random_np(M)
    rng = np.random.default_rng(M)
    v0 = np.zeros((M, M), order="F")
    v0[0, :], v0[-1, :], v0[:, 0], v0[:, -1] = rng.standard_normal((4, M))   # non-zero Dirichlet data
    if not zero_guess:
        v0[1:-1, 1:-1] = rng.standard_normal((M - 2, M - 2)) * 0.01          # and a non-zero guess
    v0[0, 3] = v0[5, 0] = v0[-1, -1] = v0[-1, 7] = v0[M // 2, -1] = -0.0     # main.cpp:341 turns these into +0.0
    return f, v0


@pytest.mark.parametrize("R,M,tail_m,Nv,nd,nu", [
    (1, 257, 33, 3, 2, 2), (2, 513, 65, 3, 2, 2), (3, 1025, 129, 3, 2, 2), (4, 2049, 257, 3, 2, 2),
    (8, 4097, 257, 3, 2, 2), (2, 1025, 65, 4, 1, 2), (3, 1025, 129, 3, 2, 1), (2, 513, 65, 3, 0, 2),
    (2, 513, 65, 3, 2, 0), (2, 513, 65, 2, 1, 0), (2, 513, 65, 3, 3, 1), (4, 2049, 129, 2, 4, 0),
    (2, 513, 65, 3, 5, 6), (3, 1025, 129, 2, 3, 3), (2, 513, 65, 2, 0, 0), (5, 4097, 513, 2, 2, 2),
])
def test_slabs_in_one_process_match_the_oracle(mg, O, R, M, tail_m, Nv, nd, nu):
    f, v0 = _problem(M)
    dx = 1.0 / (M - 1)
    P = mg.MultiPlan(M, dx, 2, nslabs=R, devices=[0] * R, tail_m=tail_m)
    assert P.nslabs == R and P.tail_m <= tail_m and P.ndist >= 1
    P.upload(f, v0)
    out = P.cycles(Nv, nd, nu, 0.6)
    sol = P.download()
    P.close()
    ref_sol, ref_rn, _ = O.multigrid(f, v0, dx, Nv, nd, nu, 0.6, 2, omp=M > 1025)
    assert_bits(out.ResidualNorms, ref_rn, "ResNorms R=%d" % R)
    assert_bits(sol, ref_sol, "Sol R=%d" % R)
    assert out.Times[0] == 0 and np.all(np.diff(out.Times) > 0)


@pytest.mark.parametrize("R", [2, 3])
def test_pure_jacobi_and_split_calls(mg, O, R):
    M = 513
    f, v0 = _problem(M)
    dx = 1.0 / (M - 1)
    P = mg.MultiPlan(M, dx, 2, nslabs=R, devices=[0] * R, tail_m=65)
    P.upload(f, v0)
    out = P.cycles(4, 2, 2, 0.6, pureJacobi=True)
    ref_sol, ref_rn, _ = O.multigrid(f, v0, dx, 4, 2, 2, 0.6, 2, pure_jacobi=True)
    assert_bits(out.ResidualNorms, ref_rn, "pureJacobi ResNorms")
    assert_bits(P.download(), ref_sol, "pureJacobi Sol")
    # 3 + 2 cycles equal one call of 5, and zero cycles return the initial norm only
    P.upload(f, v0)
    a = P.cycles(3, 2, 2, 0.6)
    z = P.cycles(0, 2, 2, 0.6)
    b = P.cycles(2, 2, 2, 0.6)
    ref_sol, ref_rn, _ = O.multigrid(f, v0, dx, 5, 2, 2, 0.6, 2)
    assert_bits(np.concatenate([a.ResidualNorms, b.ResidualNorms[1:]]), ref_rn, "split ResNorms")
    assert z.ResidualNorms.shape == (1,) and z.ResidualNorms[0] == a.ResidualNorms[-1]
    assert_bits(P.download(), ref_sol, "split Sol")
    P.close()


def test_solve2d_with_the_ngpus_option(mg, O):
    """The call a reference-side shim makes (main.cpp:445 -> mgx_solve2d) with
    options.ngpus = 4: four slabs (on the one GPU of the box), same bits; a grid too small to
    split takes the single-device path."""
    M = 1025
    f, v0 = _problem(M)
    dx = 1.0 / (M - 1)
    ref_sol, ref_rn, _ = O.multigrid(f, v0, dx, 4, 2, 2, 0.6, 2)
    v = v0.copy(order="F")
    out = mg.solve_host(f, v, dx, 4, 2, 2, 0.6, 2, ngpus=4, tail_m=129)
    assert_bits(out.ResidualNorms, ref_rn, "ResNorms ngpus=4")
    assert_bits(v, ref_sol, "Sol ngpus=4")
    M = 65
    f, v0 = _problem(M)
    ref_sol, ref_rn, _ = O.multigrid(f, v0, 1.0 / 64, 3, 2, 2, 0.6, 2)
    v = v0.copy(order="F")
    out = mg.solve_host(f, v, 1.0 / 64, 3, 2, 2, 0.6, 2, ngpus=8)
    assert_bits(out.ResidualNorms, ref_rn, "ResNorms small grid")
    assert_bits(v, ref_sol, "Sol small grid")
    with pytest.raises(mg.MgxError):
        mg.MultiPlan(65, 1.0 / 64, 2, nslabs=8, devices=[0] * 8)


def test_exact_coarsest_solve_and_even_coarsest_level(mg, O):
    """705 = 11*2^6+1, coarsest 11: levels ... 23, 12; every slab's tail solves the 12x12 grid exactly."""
    M = 705
    f, v0 = _problem(M)
    dx = 1.0 / (M - 1)
    P = mg.MultiPlan(M, dx, 11, nslabs=2, devices=[0, 0], tail_m=89, coarse_solve=mg.COARSE_EXACT)
    P.upload(f, v0)
    out = P.cycles(4, 2, 2, 0.6)
    sol = P.download()
    P.close()
    ref_sol, ref_rn, _ = O.multigrid(f, v0, dx, 4, 2, 2, 0.6, 11, coarse_exact=True)
    assert np.allclose(out.ResidualNorms, ref_rn, rtol=1e-6, atol=1e-9 * ref_rn[0])
    assert np.abs(sol - ref_sol).max() <= 1e-9 * np.abs(ref_sol).max()
    # the reference's behaviour (12x12 left alone): bit for bit
    P = mg.MultiPlan(M, dx, 11, nslabs=3, devices=[0] * 3, tail_m=89)
    P.upload(f, v0)
    out = P.cycles(3, 2, 2, 0.6)
    ref_sol, ref_rn, _ = O.multigrid(f, v0, dx, 3, 2, 2, 0.6, 11)
    assert_bits(out.ResidualNorms, ref_rn, "ResNorms")
    assert_bits(P.download(), ref_sol, "Sol")
    P.close()


@pytest.mark.parametrize("R", [2, 8])
def test_16385_partitioned_equals_the_single_plan(mg, R):
    """The benchmark size over 2 and 8 slabs (tail from 1025^2), device-generated right-hand
    side, tuned launch shapes: residual history and every column of the solution equal the
    single-device plan's, bit for bit."""
    import torch
    from multigrid_amd import rhs
    from multigrid_amd.solver import _DevPtr
    M, Nv = 16385, 6
    dx = 1.0 / (M - 1)
    plan = mg.Plan(M, dx, 2)
    plan.set(rhs.random_torch(M), mg.colmajor(M, M))
    ref = plan.cycles(Nv, 2, 2, 0.6)
    want = plan.get()
    P = mg.MultiPlan(M, dx, 2, nslabs=R, devices=[0] * R)
    assert P.ndist == 4 and P.tail_m == 1025
    P.set_random_rhs()
    P.tune(2, 2, 0.6)
    out = P.cycles(Nv, 2, 2, 0.6)
    assert_bits(out.ResidualNorms, ref.ResidualNorms, "ResNorms 16385 R=%d" % R)
    for k in range(R):
        s = P.slab(k)
        v = torch.as_tensor(_DevPtr(s.v_dev, (s.ncols, s.m)), device="cuda")
        a = s.own_first - s.col_first
        own = v[a:a + s.own_count]
        assert torch.equal(own.view(torch.int64), want.t()[s.own_first:s.own_first + s.own_count].view(torch.int64))
    P.close()
    plan.close()


def test_exchange_and_gather_counts(mg):
    """Lean schedule: one neighbour exchange and one gather per cycle (plus the exchange before
    the initial norm and the one after the closing up leg)."""
    M = 2049
    P = mg.MultiPlan(M, 1.0 / (M - 1), 2, nslabs=4, devices=[0] * 4, tail_m=257)
    P.set_random_rhs()
    P.profile(True)
    P.cycles(5, 2, 2, 0.6)
    x, g, host, enq = P.stats()
    assert g == 5 and x == 1 + 5 + 1 and 0 < enq <= host
    P.close()


def test_slabs_of_different_width_agree_on_the_number_of_exchanges(mg, O):
    """193 columns over 6 slabs, V(0,1), one distributed level: some slabs are wide enough to
    split their level-0 launch around the exchange and some are not, and the opening down leg
    has nothing to exchange.  Every slab must still take part in the same number of
    hand-shakes (a slab that counted one more waited for ever: found by tools/fuzz_parity.py)."""
    M = 193
    f, v0 = _problem(M)
    dx = 0.37
    for (nd, nu, Nv) in ((0, 1, 1), (0, 2, 2), (1, 0, 2), (0, 4, 1)):
        P = mg.MultiPlan(M, dx, 6, nslabs=6, devices=[0] * 6, tail_m=129)
        assert P.ndist == 1
        P.upload(f, v0)
        a = P.cycles(0, nd, nu, 0.7)
        b = P.cycles(Nv, nd, nu, 0.7)
        sol = P.download()
        P.close()
        ref_sol, ref_rn, _ = O.multigrid(f, v0, dx, Nv, nd, nu, 0.7, 6)
        assert a.ResidualNorms[0] == ref_rn[0]
        assert_bits(b.ResidualNorms, ref_rn, "ResNorms V(%d,%d)" % (nd, nu))
        assert_bits(sol, ref_sol, "Sol V(%d,%d)" % (nd, nu))


@pytest.mark.parametrize("R,M,tail_m,ncyc,nd,nu", [(1, 513, 65, 1, 2, 2), (2, 513, 65, 1, 2, 2), (3, 1025, 129, 2, 2, 2),
                                                  (2, 2049, 257, 1, 2, 2), (4, 2049, 129, 1, 1, 2), (2, 513, 65, 2, 3, 3),
                                                  (2, 257, 129, 1, 2, 1)])
def test_full_multigrid_over_slabs_in_one_process(mg, O, R, M, tail_m, ncyc, nd, nu):
    """mgx_mplan_fmg against the oracle's composition of the reference's operators (mgo_fmg),
    bit for bit; a second call (staging arrays and counters carried over) gives the same; plain
    cycles afterwards continue from the result."""
    f, v0 = _problem(M)
    dx = 1.0 / (M - 1)
    ref_sol, ref_rn = O.fmg(f, v0, dx, ncyc, nd, nu, 0.6, 2)
    P = mg.MultiPlan(M, dx, 2, nslabs=R, devices=[0] * R, tail_m=tail_m)
    for rep in range(2):
        P.upload(f, v0)
        out = P.fmg(ncyc, nd, nu, 0.6)
        assert_bits(out.ResidualNorms, ref_rn, "FMG ResNorms R=%d call %d" % (R, rep))
        assert_bits(P.download(), ref_sol, "FMG Sol R=%d call %d" % (R, rep))
    more = P.cycles(2, nd, nu, 0.6)
    cont_sol, cont_rn, _ = O.multigrid(f, ref_sol, dx, 2, nd, nu, 0.6, 2)
    assert_bits(more.ResidualNorms, cont_rn, "cycles after FMG")
    assert_bits(P.download(), cont_sol, "Sol after FMG + cycles")
    P.close()


def test_full_multigrid_16385_over_four_slabs_equals_the_single_plan(mg):
    """Full multigrid + V-cycles at 16385^2 over four slabs (tail from 1025^2): history and
    every column equal mgx_plan_fmg on one device."""
    import torch
    from multigrid_amd import rhs
    from multigrid_amd.solver import _DevPtr
    M = 16385
    dx = 1.0 / (M - 1)
    plan = mg.Plan(M, dx, 2)
    plan.set(rhs.random_torch(M), mg.colmajor(M, M))
    ref = plan.fmg(2, 2, 2, 0.6)
    want = plan.get()
    P = mg.MultiPlan(M, dx, 2, nslabs=4, devices=[0] * 4)
    P.set_random_rhs()
    out = P.fmg(2, 2, 2, 0.6)
    assert_bits(out.ResidualNorms, ref.ResidualNorms, "FMG ResNorms 16385")
    for k in range(4):
        s = P.slab(k)
        v = torch.as_tensor(_DevPtr(s.v_dev, (s.ncols, s.m)), device="cuda")
        a = s.own_first - s.col_first
        assert torch.equal(v[a:a + s.own_count].view(torch.int64),
                           want.t()[s.own_first:s.own_first + s.own_count].view(torch.int64))
    P.close()
    plan.close()


def test_a_failed_call_poisons_the_plan(mg, monkeypatch):
    """One slab gives up in the middle of a call (injected, the way a HIP error would): the other
    slabs stop waiting for it, the call reports the failure, and from then on every entry point
    but destroy refuses with MGX_ERR_STATE instead of running on counters and ping-pong indices
    that no longer agree between the slabs."""
    M = 513
    f, v0 = _problem(M)
    P = mg.MultiPlan(M, 1.0 / (M - 1), 2, nslabs=3, devices=[0] * 3, tail_m=65)
    P.upload(f, v0)
    ok = P.cycles(2, 2, 2, 0.6)
    assert ok.ResidualNorms[2] < ok.ResidualNorms[0]
    monkeypatch.setenv("MGX_TEST_FAIL_SLAB", "1")
    with pytest.raises(mg.MgxError) as e:
        P.cycles(3, 2, 2, 0.6)
    assert e.value.code == -5 and "injected" in str(e.value)
    monkeypatch.delenv("MGX_TEST_FAIL_SLAB")
    for call in (lambda: P.cycles(1, 2, 2, 0.6), lambda: P.upload(f, v0), lambda: P.download(),
                 lambda: P.fmg(1, 2, 2, 0.6), lambda: P.tune(2, 2, 0.6)):
        with pytest.raises(mg.MgxError) as e:
            call()
        assert e.value.code == -5 and "earlier call" in str(e.value)
    P.close()
    # a fresh plan of the same shape works (nothing of the failed one lingers in the library)
    P = mg.MultiPlan(M, 1.0 / (M - 1), 2, nslabs=3, devices=[0] * 3, tail_m=65)
    P.upload(f, v0)
    again = P.cycles(2, 2, 2, 0.6)
    assert_bits(again.ResidualNorms, ok.ResidualNorms, "after a poisoned plan")
    P.close()


def test_random_partitioned_cases_small_budget():
    """tools/fuzz_parity.py restricted to the partitioned plan (random slab counts 1-8, sweep
    counts, split calls, full multigrid, odd hierarchies): about 150 cases or 100 seconds, every
    one bit-identical to the oracle.  Keeps the threads' hand-shake schedule covered (a sweep of
    this kind found a deadlock between slabs of different width)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, FUZZ_MODES="3,4", FUZZ_CASES="150", MGX_MPLAN_TIMEOUT="60")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "100", "20261112", "700"],
                       env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=400)
    tail = r.stdout.decode()[-1500:]
    assert r.returncode == 0, tail
    assert "bit-identical to the oracle" in tail


def _free_hbm_gb():
    import torch
    return torch.cuda.mem_get_info()[0] / 1e9


def _plan_and_slabs_agree(mg, M, coarsest, R, Nv, exact, fmg=0):
    """A single plan and a partitioned plan of R slabs on the same device-generated right-hand
    side: (history of the plan, history of the slabs, number of solution words that differ)."""
    import torch
    from multigrid_amd import _lib
    from multigrid_amd.slab import _device_view
    from multigrid_amd.solver import _DevPtr
    from test_gpu_configs import _fill_plan_rhs
    cs = mg.COARSE_EXACT if exact else mg.COARSE_REFERENCE
    dx = 1.0 / (M - 1)
    plan = mg.Plan(M, dx, coarsest, coarse_solve=cs)
    _fill_plan_rhs(mg, plan, M)
    ref = plan.fmg(fmg, 2, 2, 0.6) if fmg else plan.cycles(Nv, 2, 2, 0.6)
    want = _device_view(_lib.load().mgx_plan_v_dev(plan._h), M, torch.device("cuda"))  # row k = column k
    P = mg.MultiPlan(M, dx, coarsest, nslabs=R, devices=[0] * R, coarse_solve=cs)
    P.set_random_rhs()
    out = P.fmg(fmg, 2, 2, 0.6) if fmg else P.cycles(Nv, 2, 2, 0.6)
    differ, worst = 0, 0.0
    for k in range(R):
        s = P.slab(k)
        v = torch.as_tensor(_DevPtr(s.v_dev, (s.ncols, s.m)), device="cuda")
        a = s.own_first - s.col_first
        own, w = v[a:a + s.own_count], want[s.own_first:s.own_first + s.own_count]
        differ += int((own.view(torch.int64) != w.view(torch.int64)).sum().item())
        worst = max(worst, float((own - w).abs().max().item()))
    scale = float(want.abs().max().item())
    ndist, tail_m = P.ndist, P.tail_m
    P.close()
    plan.close()
    del want
    torch.cuda.empty_cache()
    return ref.ResidualNorms, out.ResidualNorms, differ, worst / scale, ndist, tail_m


def test_22529_over_two_slabs_like_the_weak_scaling_run(mg):
    """The 2-GPU weak-scaling size at its partitioned shape: 22529 = 11*2^11+1 over two slabs,
    coarsest 11.  With the reference's coarsest behaviour (the 12x12 grid left alone) the slabs
    equal the single plan bit for bit; with the exact coarsest solve every slab's tail runs the
    same sine-transform solve as the single plan's, so the bits are equal there too."""
    if _free_hbm_gb() < 60:
        pytest.skip("needs 60 GB of free HBM")
    a, b, differ, _, ndist, tail_m = _plan_and_slabs_agree(mg, 22529, 11, 2, 3, exact=False)
    assert ndist >= 4 and tail_m <= 1409
    assert_bits(b, a, "ResNorms 22529 over 2 slabs")
    assert differ == 0
    a, b, differ, rel, _, _ = _plan_and_slabs_agree(mg, 22529, 11, 2, 12, exact=True)
    from test_gpu_configs import _stop
    assert _stop(a) == _stop(b)
    assert np.allclose(a, b, rtol=1e-6, atol=1e-9 * a[0]) and rel <= 1e-9
    print("22529^2 over 2 slabs, exact coarsest solve: stop", _stop(b), "words that differ", differ, "rel", rel)


def test_45057_over_eight_slabs_like_the_weak_scaling_run(mg):
    """The 8-GPU weak-scaling size at its partitioned shape: 45057 = 11*2^12+1 over eight slabs
    (all on this one GPU: 81 GB for the plan, 95 GB for the slabs, one after the other would not
    do -- both are alive for the comparison, so it needs most of the card)."""
    if _free_hbm_gb() < 200:
        pytest.skip("needs 200 GB of free HBM")
    a, b, differ, _, ndist, tail_m = _plan_and_slabs_agree(mg, 45057, 11, 8, 2, exact=False)
    assert ndist >= 5
    assert_bits(b, a, "ResNorms 45057 over 8 slabs")
    assert differ == 0
    a, b, differ, rel, _, _ = _plan_and_slabs_agree(mg, 45057, 11, 8, 6, exact=True)
    assert np.allclose(a, b, rtol=1e-6, atol=1e-9 * a[0]) and rel <= 1e-9
    assert b[6] < 5e-3 * b[0]      # six cycles at the usual 0.35 per cycle


def test_full_multigrid_32769_over_four_slabs(mg):
    """Full multigrid + V-cycles at 32769^2 (the configuration's size) over four slabs: the
    history and every column equal mgx_plan_fmg on one device."""
    if _free_hbm_gb() < 120:
        pytest.skip("needs 120 GB of free HBM")
    a, b, differ, _, ndist, tail_m = _plan_and_slabs_agree(mg, 32769, 2, 4, 0, exact=False, fmg=2)
    assert ndist == 5 and tail_m == 1025
    assert_bits(b, a, "FMG ResNorms 32769 over 4 slabs")
    assert differ == 0
