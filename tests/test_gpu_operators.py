"""GPU parity tests, one per reference operator: the HIP kernel behind each
C-ABI entry point against the CPU oracle on the same seeded inputs, bit for bit
(fp64 evaluated in the reference's order with no FMA, so the bar is equality,
not a tolerance)."""
import numpy as np
import pytest

from conftest import assert_bits

pytestmark = pytest.mark.gpu

SIZES = [3, 5, 9, 17, 33, 65, 67, 129, 131, 257, 513, 1025]


def _rand(m, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    return np.asfortranarray(rng.standard_normal((m, m)) * scale)


@pytest.mark.parametrize("m", SIZES)
@pytest.mark.parametrize("niter,omega", [(1, 0.6), (2, 0.6), (3, 0.9), (4, 1.8)])
def test_relax(mg, O, m, niter, omega):
    u, f = _rand(m, m), _rand(m, m + 1, 50.0)
    dx = 1.0 / (m - 1)
    p = mg.gridtype(mg.to_device(f), mg.to_device(u), dx)
    mg.relax(p, niter, omega)
    assert_bits(mg.to_numpy(p.v), O.relax(u, f, dx, niter, omega), "relax m=%d" % m)


def test_relax_zero_iterations_is_identity(mg):
    u, f = _rand(33, 1), _rand(33, 2)
    p = mg.gridtype(mg.to_device(f), mg.to_device(u), 0.1)
    mg.relax(p, 0, 0.6)
    assert_bits(mg.to_numpy(p.v), u)


@pytest.mark.parametrize("m", SIZES)
def test_residual(mg, O, m):
    u, f = _rand(m, 3 * m), _rand(m, 3 * m + 1, 10.0)
    dx = 1.0 / (m - 1)
    p = mg.gridtype(mg.to_device(f), mg.to_device(u), dx)
    r = mg.to_numpy(mg.residual(p))
    assert_bits(r, O.residual(u, f, dx), "residual m=%d" % m)
    assert np.all(r[0, :] == 0) and np.all(r[-1, :] == 0) and np.all(r[:, 0] == 0) and np.all(r[:, -1] == 0)


@pytest.mark.parametrize("m", SIZES)
def test_coarsen(mg, O, m):
    fine = _rand(m, 5 * m)
    mc = (m - 1) // 2 + 1
    c = mg.colmajor(mc, mc)
    c += 7.0  # must be overwritten, boundary included (main.cpp:211)
    mg.coarsen(mg.to_device(fine), c)
    assert_bits(mg.to_numpy(c), O.coarsen(fine), "coarsen m=%d" % m)


def test_coarsen_corner_quirk(mg):
    # main.cpp:218: weights seen by coarse (1,1) from the 3x3 fine patch around (2,2)
    w = np.zeros((3, 3))
    for di in (-1, 0, 1):
        for dj in (-1, 0, 1):
            e = np.zeros((9, 9), order="F")
            e[2 + di, 2 + dj] = 1.0
            c = mg.colmajor(5, 5)
            mg.coarsen(mg.to_device(e), c)
            w[di + 1, dj + 1] = mg.to_numpy(c)[1, 1]
    assert np.array_equal(w, np.array([[1 / 16, 1 / 8, 0.0], [1 / 8, 1 / 4, 1 / 8], [2 / 16, 1 / 8, 1 / 16]]))


@pytest.mark.parametrize("m", SIZES)
def test_refine_and_prolong_add(mg, O, m):
    mc = (m - 1) // 2 + 1
    c = _rand(mc, 7 * m)
    fine = mg.colmajor(m, m)
    fine += 3.0
    mg.refine(mg.to_device(c), fine)
    ref = O.refine(c, m)
    assert_bits(mg.to_numpy(fine), ref, "refine m=%d" % m)
    v = _rand(m, 7 * m + 1)
    vd = mg.to_device(v)
    mg.prolong_add(mg.to_device(c), vd)
    expect = v.copy(order="F")
    O.lib().mgo_plus_equals(expect.ctypes.data_as(O._dp), ref.ctypes.data_as(O._dp), expect.size)
    assert_bits(mg.to_numpy(vd), expect, "prolong_add m=%d" % m)


def test_direct_solve(mg, O):
    u = np.zeros((3, 3), order="F")
    u[0, 1], u[2, 1], u[1, 0], u[1, 2] = 1.5, -2.0, 3.25, 4.0
    f = np.zeros((3, 3), order="F")
    f[1, 1] = 7.0
    p = mg.gridtype(mg.to_device(f), mg.to_device(u), 0.3)
    assert mg.direct_solve(p) is True
    expect, _ = O.direct_solve(u, f, 0.3)
    assert_bits(mg.to_numpy(p.v), expect)
    u5 = _rand(5, 9)
    p5 = mg.gridtype(mg.to_device(_rand(5, 10)), mg.to_device(u5), 0.25)
    assert mg.direct_solve(p5) is False  # main.cpp:167
    assert_bits(mg.to_numpy(p5.v), u5)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 4097, 257 * 257, 1 << 21])
def test_calc_norm(mg, O, n):
    import torch
    rng = np.random.default_rng(n)
    a = rng.standard_normal(n)
    assert mg.calcNorm(torch.from_numpy(a).cuda()) == O.calc_norm(a)
    b = -np.abs(a)
    assert mg.calcNorm(torch.from_numpy(b).cuda()) == O.calc_norm(b)


def test_calc_norm_special_values(mg, O):
    import torch
    for a in ([1.0, np.nan, -0.5], [np.nan, np.nan], [np.inf, 1.0], [-np.inf, 1.0], [0.0, 0.0], [-3.0]):
        a = np.array(a)
        got, want = mg.calcNorm(torch.from_numpy(a).cuda()), O.calc_norm(a)
        assert got == want or (np.isnan(got) and np.isnan(want)), (a, got, want)


@pytest.mark.parametrize("m", SIZES)
def test_residual_norm(mg, O, m):
    u, f = _rand(m, 11 * m), _rand(m, 11 * m + 1, 10.0)
    dx = 1.0 / (m - 1)
    p = mg.gridtype(mg.to_device(f), mg.to_device(u), dx)
    assert mg.residual_norm(p) == O.calc_norm(O.residual(u, f, dx))


def test_residual_norm_of_exact_zero(mg):
    z = np.zeros((17, 17), order="F")
    p = mg.gridtype(mg.to_device(z), mg.to_device(z), 0.5)
    n = mg.residual_norm(p)
    assert n == 0.0 and not np.signbit(n)


def test_shape_errors(mg):
    import torch
    with pytest.raises(ValueError, match="not square"):
        mg.relax(mg.gridtype(mg.colmajor(5, 7), mg.colmajor(5, 7), 1.0), 1, 0.6)
    with pytest.raises(ValueError, match="even rows"):
        mg.relax(mg.gridtype(mg.colmajor(6, 6), mg.colmajor(6, 6), 1.0), 1, 0.6)
    with pytest.raises(ValueError, match="column-major"):
        t = torch.zeros((5, 5), dtype=torch.float64, device="cuda")
        mg.relax(mg.gridtype(t, t, 1.0), 1, 0.6)
    with pytest.raises(mg.MgxError):
        mg.coarsen(mg.colmajor(9, 9), mg.colmajor(7, 7))


def test_rhs_generators_match_on_gpu(mg):
    from multigrid_amd import rhs
    for M in (17, 257):
        assert_bits(mg.to_numpy(rhs.smooth_torch(M)), rhs.smooth_np(M), "smooth")
        assert_bits(mg.to_numpy(rhs.random_torch(M)), rhs.random_np(M), "random")
