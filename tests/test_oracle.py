"""CPU tests: the oracle against the reference's known answers, against the
reference's own DTSource code (oracle/_ref, when built) and against its frozen
vectors.  No GPU, no product code."""
import numpy as np
import pytest

from conftest import assert_bits


def _case9(O):
    m = 9
    h = 1.0 / 8
    f = np.zeros((m, m), order="F")
    for i in range(1, m - 1):
        for j in range(1, m - 1):
            f[i, j] = 1 + i + 10 * j
    return m, h, f


def _idx(key):
    a, b = key.strip("()").split(",")
    return int(a), int(b)


def test_known_answers_operators_9(O, known):
    K = known["case9"]
    m, h, f = _case9(O)
    u = O.relax(np.zeros((m, m), order="F"), f, h, 1, 0.6)
    for key, val in K["relax_1_0.6"].items():
        assert u[_idx(key)] == float(val), key
    r = O.residual(u, f, h)
    for key, val in K["residual_after_relax"].items():
        if key == "norm":
            assert O.calc_norm(r) == float(val)
        else:
            assert r[_idx(key)] == float(val), key
    c = O.coarsen(r)
    for key, val in K["coarsen_of_residual"].items():
        assert c[_idx(key)] == float(val), key


def test_known_answers_restriction_quirk(O, known):
    # main.cpp:218: fine(2i+1,2j-1) enters twice, fine(2i-1,2j+1) never.
    K = known["case9"]["coarsen_indicator"]
    e = np.zeros((9, 9), order="F")
    e[1, 3] = 1.0
    assert O.coarsen(e)[1, 1] == float(K["fine(1,3)->c(1,1)"])
    e = np.zeros((9, 9), order="F")
    e[3, 1] = 1.0
    assert O.coarsen(e)[1, 1] == float(K["fine(3,1)->c(1,1)"])
    # the full 3x3 weight pattern seen by coarse (1,1)
    w = np.zeros((3, 3))
    for di in (-1, 0, 1):
        for dj in (-1, 0, 1):
            e = np.zeros((9, 9), order="F")
            e[2 + di, 2 + dj] = 1.0
            w[di + 1, dj + 1] = O.coarsen(e)[1, 1]
    expect = np.array([[1 / 16, 1 / 8, 0.0], [1 / 8, 1 / 4, 1 / 8], [2 / 16, 1 / 8, 1 / 16]])
    assert np.array_equal(w, expect)


def test_known_answers_refine(O, known):
    K = known["case9"]["refine"]
    c = np.zeros((5, 5), order="F")
    for key, val in K["coarse"].items():
        c[_idx(key)] = val
    fine = O.refine(c)
    for key, val in K["fine"].items():
        assert fine[_idx(key)] == float(val), key
    assert np.all(fine[0, :] == 0) and np.all(fine[-1, :] == 0) and np.all(fine[:, 0] == 0) and np.all(fine[:, -1] == 0)


def test_known_answers_multigrid_9(O, known):
    K = known["case9"]["multigrid_5_3_3_0.6_2"]
    m, h, f = _case9(O)
    sol, rn, tm = O.multigrid(f, np.zeros((m, m), order="F"), h, 5, 3, 3, 0.6, 2)
    assert [float(x) for x in K["ResNorms"]] == list(rn)
    assert sol[4, 4] == float(K["Sol(4,4)"])
    assert tm[0] == 0 and np.all(np.diff(tm) >= 0)


def test_known_answers_multigrid_257(O, known, f_smooth_257):
    K = known["case257"]
    M = 257
    sol, rn, _ = O.multigrid(f_smooth_257, np.zeros((M, M), order="F"), 1.0 / (M - 1), 30, 2, 2, 0.6, 2)
    assert list(rn[:6]) == [float(x) for x in K["ResNorms[0..5]"]]
    assert rn[30] == float(K["ResNorms[30]"])
    assert int(np.argmax(rn <= 1e-8)) == K["first_cycle_le_1e-8"]
    assert sol.max() == float(K["max(Sol)"])


def test_depth_rule(O):
    # main.cpp:286 in float arithmetic
    assert [O.depth(m, 2) for m in (3, 5, 9, 257, 4097, 16385, 32769)] == [0, 1, 2, 7, 11, 13, 14]
    assert O.depth(257, 4) == 6
    assert O.depth(3, 8) == -1  # int() truncates -1.5 toward zero


def test_frozen_vectors(O, vectors33):
    V = vectors33
    u, f, c, dx = V["u"], V["f"], V["c"], float(V["dx"])
    assert_bits(O.relax(u, f, dx, 3, 0.6), V["relax3_w06"], "relax3")
    assert_bits(O.relax(u, f, dx, 1, 0.9), V["relax1_w09"], "relax1")
    assert_bits(O.residual(u, f, dx), V["residual"], "residual")
    assert_bits(O.coarsen(u), V["coarsen_u"], "coarsen")
    assert_bits(O.refine(c), V["refine_c"], "refine")
    assert O.calc_norm(u) == float(V["norm_u"])
    sol, rn, _ = O.multigrid(f, np.zeros_like(u), dx, 8, 2, 2, 0.6, 2)
    assert_bits(sol, V["mg_sol"], "mg_sol")
    assert_bits(rn, V["mg_resnorms"], "mg_resnorms")


def test_norm_semantics(O):
    # DTDoubleArray.cpp:883-899,923-939: NaNs never win; main.cpp:149 combines max and -min.
    assert O.calc_norm(np.array([1.0, -3.0, 2.0])) == 3.0
    assert O.calc_norm(np.array([1.0, np.nan, -0.5])) == 1.0
    assert O.calc_norm(np.array([-1.0, -2.0])) == 2.0
    assert O.calc_norm(np.array([np.nan, np.nan])) == -np.inf
    assert O.calc_norm(np.array([np.inf, 1.0])) == np.inf
    assert O.calc_norm(np.array([-np.inf, 1.0])) == np.inf


def test_direct_solve_only_3x3(O):
    u = np.zeros((3, 3), order="F")
    u[0, 1], u[2, 1], u[1, 0], u[1, 2] = 1.0, 2.0, 3.0, 4.0
    f = np.zeros((3, 3), order="F")
    f[1, 1] = 8.0
    out, rc = O.direct_solve(u, f, 0.5)
    assert rc == 0 and out[1, 1] == (1.0 + 2.0 + 3.0 + 4.0 - 8.0 * 0.25) * 0.25
    u5 = np.ones((5, 5), order="F")
    out5, rc = O.direct_solve(u5, np.ones((5, 5), order="F"), 0.25)
    assert rc == 1 and np.array_equal(out5, u5)  # main.cpp:167: message, nothing changes


def test_hierarchy_refusals(O):
    with pytest.raises(ValueError):
        O.multigrid(np.zeros((7, 7), order="F"), np.zeros((7, 7), order="F"), 1.0, 1, 1, 1, 0.6, 2)  # 7 -> 4 (even)
    with pytest.raises(ValueError):
        O.multigrid(np.zeros((3, 3), order="F"), np.zeros((3, 3), order="F"), 1.0, 1, 1, 1, 0.6, 8)  # depth -1


def test_pure_jacobi_branch(O):
    # main.cpp:347-349: one fine-grid sweep per "cycle"
    rng = np.random.default_rng(3)
    m = 17
    f = np.asfortranarray(rng.standard_normal((m, m)))
    v0 = np.zeros((m, m), order="F")
    sol, rn, tm = O.multigrid(f, v0, 1.0 / 16, 4, 2, 2, 0.6, 2, pure_jacobi=True)
    assert_bits(sol, O.relax(v0, f, 1.0 / 16, 4, 0.6), "pure jacobi")
    assert np.all(tm == 0)


# ---- against the reference's own DTSource code (oracle/_ref) -----------------

def _ref(O):
    R = O.ref_lib()
    if R is None:
        pytest.skip("oracle/_ref not built (reference checkout absent)")
    return R


def test_ref_maximum_minimum(O):
    R = _ref(O)
    rng = np.random.default_rng(11)
    cases = [rng.standard_normal(1000), -np.abs(rng.standard_normal(257)), np.abs(rng.standard_normal(64)),
             np.array([np.nan, 1.0, -2.0, np.nan]), np.array([np.nan]), np.array([0.0, -0.0]), np.array([-0.0, 0.0]),
             np.array([np.inf, -np.inf, 3.0])]
    for a in cases:
        a = np.ascontiguousarray(a, dtype=np.float64)
        p = a.ctypes.data_as(O._dp)
        mx, mn = R.ref_maximum(p, a.size), R.ref_minimum(p, a.size)
        assert_bits(np.array([O.lib().mgo_maximum(p, a.size)]), np.array([mx]), "Maximum")
        assert_bits(np.array([O.lib().mgo_minimum(p, a.size)]), np.array([mn]), "Minimum")
        expect = -mn if (mx < -mn) else mx  # std::max<double>(a,b) = (a<b)?b:a
        assert_bits(np.array([O.calc_norm(a)]), np.array([expect]), "calcNorm")


def test_ref_plus_equals(O):
    R = _ref(O)
    rng = np.random.default_rng(12)
    a = rng.standard_normal(513)
    b = rng.standard_normal(513)
    a[0], b[0] = -0.0, 0.0
    x = a.copy()
    y = a.copy()
    R.ref_plus_equals(x.ctypes.data_as(O._dp), b.ctypes.data_as(O._dp), a.size)
    O.lib().mgo_plus_equals(y.ctypes.data_as(O._dp), b.ctypes.data_as(O._dp), a.size)
    assert_bits(x, y, "operator+=")


def test_ref_column_major(O):
    R = _ref(O)
    # DTDoubleArray.h:125-128: (i,j) -> i + j*m
    for (m, n, i, j) in [(5, 5, 1, 0), (5, 5, 0, 1), (9, 7, 3, 4), (3, 3, 2, 2)]:
        assert R.ref_offset_of(m, n, i, j) == i + j * m


def test_threaded_oracle_build_returns_the_same_bits(O):
    """libmg_oracle_omp.so (the all-cores CPU baseline) is the same source with OpenMP over
    columns: Jacobi makes the result independent of the thread that computes a point."""
    from multigrid_amd import rhs
    M = 257
    f = rhs.random_np(M)
    rng = np.random.default_rng(3)
    v0 = np.asfortranarray(rng.standard_normal((M, M)))
    for (nd, nu, pj) in [(2, 2, False), (1, 3, False), (0, 0, True)]:
        a = O.multigrid(f, v0, 1.0 / (M - 1), 3, nd, nu, 0.7, 2, pure_jacobi=pj)
        b = O.multigrid(f, v0, 1.0 / (M - 1), 3, nd, nu, 0.7, 2, pure_jacobi=pj, omp=True)
        assert_bits(a[0], b[0], "Sol")
        assert_bits(a[1], b[1], "ResNorms")
    assert O.lib_omp().mgo_threads() >= 1


@pytest.mark.parametrize("m", [4, 5, 6, 9, 12, 17, 33, 65])
def test_exact_coarse_solve_matches_the_references_eigen_cholesky(O, m):
    """EXTENSION oracle: the banded Cholesky of mgo_coarse_exact against
    Eigen::SimplicialCholesky on the same 5-point system (main.cpp:79-120), built from the
    Eigen the reference vendors (oracle/_ref/libeigenref.so)."""
    if O.ref_eigen() is None:
        pytest.skip("oracle/_ref/libeigenref.so not built (reference checkout absent)")
    rng = np.random.default_rng(m)
    f = np.asfortranarray(rng.standard_normal((m, m)))
    u = np.asfortranarray(rng.standard_normal((m, m)))
    dx = 1.0 / (m - 1)
    a = O.coarse_exact(u, f, dx)
    b = O.coarse_exact(u, f, dx, eigen=True)
    assert np.abs(a - b).max() <= 1e-13 * np.abs(b).max()
    # boundary untouched, interior residual at rounding level (amplified by 1/h^2)
    assert_bits(a[0, :], u[0, :], "boundary")
    assert_bits(a[:, -1], u[:, -1], "boundary")
    r = O.residual(a, f, dx)
    assert np.abs(r[1:-1, 1:-1]).max() <= 1e-11 * (np.abs(f).max() + np.abs(u).max() / dx ** 2)


def test_exact_coarse_solve_3x3_equals_the_closed_form(O):
    rng = np.random.default_rng(0)
    f = np.asfortranarray(rng.standard_normal((3, 3)))
    u = np.asfortranarray(rng.standard_normal((3, 3)))
    a = O.coarse_exact(u, f, 0.5)
    b, rc = O.direct_solve(u, f, 0.5)
    assert rc == 0 and abs(a[1, 1] - b[1, 1]) <= 4e-16 * abs(b[1, 1])


def test_even_coarsest_level_is_a_legal_hierarchy(O):
    """177 = 11*2^4 + 1 with coarsest 11: levels 177, 89, 45, 23, 12.  coarsen() and refine()
    stay in range between 23 and 12 points (main.cpp:212-219,231-247), so the reference
    accepts it; an even level ABOVE the coarsest does not work."""
    from multigrid_amd import rhs
    M = 177
    assert O.depth(M, 11) == 4
    f = rhs.random_np(M)
    v0 = np.zeros((M, M), order="F")
    sol, rn, _ = O.multigrid(f, v0, 1.0 / (M - 1), 3, 2, 2, 0.6, 11)
    assert np.all(np.isfinite(rn)) and rn[3] < rn[0]
    sol2, rn2, _ = O.multigrid(f, v0, 1.0 / (M - 1), 3, 2, 2, 0.6, 11, coarse_exact=True)
    assert rn2[3] < 0.2 * rn[3]  # an exact coarsest solve converges much faster
    with pytest.raises(ValueError):
        O.multigrid(rhs.random_np(91), np.zeros((91, 91), order="F"), 1.0 / 90, 1, 2, 2, 0.6, 2)  # 91, 46, 23, ...


# ---- against the reference's own solver functions -----------------------------
# main.cpp:147-357 compiled from the reference checkout (oracle/_ref/librefops.so, recipe in
# oracle/Makefile).  Two layers: the committed outputs of that library
# (tests/golden/reference_vectors.npz, made by tests/golden/make_golden.py) are checked
# everywhere; the library itself is called side by side with the oracle where it is built.

@pytest.fixture(scope="module")
def refvec():
    import os
    from conftest import GOLDEN
    return np.load(os.path.join(GOLDEN, "reference_vectors.npz"))


def _golden_cases():
    import importlib.util
    import os
    from conftest import GOLDEN
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


_MG = _golden_cases()


@pytest.mark.parametrize("m", _MG.OP_SIZES)
def test_reference_vectors_operators(O, refvec, m):
    V, k, dx = refvec, "op%d_" % m, 1.0 / (m - 1)
    u, f = V[k + "u"], V[k + "f"]
    assert_bits(O.relax(u, f, dx, 1, 0.6), V[k + "relax1_w06"], "relax")
    assert_bits(O.relax(u, f, dx, 3, 0.9), V[k + "relax3_w09"], "relax x3")
    assert_bits(O.residual(u, f, dx), V[k + "residual"], "residual")
    assert_bits(np.array([O.calc_norm(u)]), np.array([float(V[k + "norm"])]), "calcNorm")
    assert_bits(O.direct_solve(u, f, dx)[0], V[k + "direct"], "direct_solve")
    if m >= 5:
        assert_bits(O.coarsen(u), V[k + "coarsen"], "coarsen")
        assert_bits(O.refine(V[k + "c"], m), V[k + "refine"], "refine")


@pytest.mark.parametrize("mf,mc", [(23, 12), (7, 4)])
def test_reference_vectors_even_level_transfers(O, refvec, mf, mc):
    V, k = refvec, "even%d_" % mf
    assert_bits(O.coarsen(V[k + "fine"], mc), V[k + "coarsen"], "coarsen to an even level")
    assert_bits(O.refine(V[k + "c"], mf), V[k + "refine"], "refine from an even level")


@pytest.mark.parametrize("case", _MG.MG_CASES, ids=[c[0] for c in _MG.MG_CASES])
def test_reference_vectors_multigrid(O, refvec, case):
    name, m, Nv, nd, nu, om, co, pj, data = case
    V = refvec
    f, v = _MG.mg_inputs(name, m, data)
    assert_bits(f, V[name + "_f"], "seeded f")   # the generator is part of the fixture
    assert_bits(v, V[name + "_v"], "seeded v")
    sol, rn, _ = O.multigrid(V[name + "_f"], V[name + "_v"], 1.0 / (m - 1), Nv, nd, nu, om, co, pure_jacobi=pj)
    assert_bits(sol, V[name + "_sol"], "Sol")
    assert_bits(rn, V[name + "_resnorms"], "ResNorms")
    assert O.depth(m, co) == int(V[name + "_depth"])


def test_reference_vectors_depth_table(O, refvec):
    for m, c, d in refvec["depth_table"]:
        assert O.depth(int(m), int(c)) == int(d), (m, c)


def _R(O):
    if O.ref_ops() is None:
        pytest.skip("oracle/_ref/librefops.so not built (reference checkout absent)")
    return O.Ref()


def _rand(rng, m, scale=1.0):
    return np.asfortranarray(rng.standard_normal((m, m)) * scale)


@pytest.mark.parametrize("m", [3, 5, 9, 17, 33, 65, 129, 257, 513])
def test_live_reference_operators(O, m):
    R = _R(O)
    rng = np.random.default_rng(1000 + m)
    dx = 1.0 / (m - 1)
    u, f = _rand(rng, m), _rand(rng, m, 100.0)
    u[0, :] = -0.0
    for (n, w) in [(1, 0.6), (2, 0.8), (4, 1.0)]:
        assert_bits(O.relax(u, f, dx, n, w), R.relax(u, f, dx, n, w), "relax %d %.1f" % (n, w))
    assert_bits(O.residual(u, f, dx), R.residual(u, f, dx), "residual")
    assert_bits(np.array([O.calc_norm(u)]), np.array([R.calc_norm(u)]), "calcNorm")
    assert_bits(O.direct_solve(u, f, dx)[0], R.direct_solve(u, f, dx), "direct_solve")
    if m >= 5:
        assert_bits(O.coarsen(u), R.coarsen(u), "coarsen")
        c = _rand(rng, (m - 1) // 2 + 1)
        assert_bits(O.refine(c, m), R.refine(c, m), "refine")
        # cur += refined over the whole array (main.cpp:341): -0.0 boundary becomes +0.0
        x = u.copy(order="F")
        O.lib().mgo_plus_equals(x.ctypes.data_as(O._dp), O.refine(c, m).ctypes.data_as(O._dp), x.size)
        assert not np.signbit(x[0, :]).any()


@pytest.mark.parametrize("m,Nv,nd,nu,om,co,pj", [
    (129, 4, 2, 2, 0.6, 2, False), (257, 3, 3, 3, 0.6, 2, False), (513, 2, 2, 2, 0.6, 2, False),
    (257, 3, 1, 2, 0.9, 2, False), (129, 3, 5, 5, 0.7, 2, False), (129, 2, 0, 0, 0.6, 2, False),
    (129, 3, 2, 2, 0.6, 4, False), (65, 3, 2, 2, 0.6, 8, False), (177, 3, 2, 2, 0.6, 11, False),
    (97, 3, 2, 2, 0.6, 3, False), (193, 3, 2, 2, 0.6, 6, False), (257, 6, 2, 2, 0.6, 2, True),
    (5, 3, 2, 2, 0.6, 2, False), (3, 3, 2, 2, 0.6, 2, False)])
def test_live_reference_multigrid(O, m, Nv, nd, nu, om, co, pj):
    R = _R(O)
    rng = np.random.default_rng(7 * m + Nv + co)
    f, v = _rand(rng, m, 50.0), _rand(rng, m)
    if m % 2:
        v[:, 0] = -0.0
        v[-1, :] = -0.0
    dx = 1.0 / (m - 1)
    assert O.depth(m, co) == R.depth(m, co)
    a = O.multigrid(f, v, dx, Nv, nd, nu, om, co, pure_jacobi=pj)
    b = R.multigrid(f, v, dx, Nv, nd, nu, om, co, pj)
    assert_bits(a[0], b[0], "Sol")
    assert_bits(a[1], b[1], "ResNorms")
    assert b[2][0] == 0 and np.all(np.diff(b[2]) >= 0)   # Times: cumulative (main.cpp:308,350)
