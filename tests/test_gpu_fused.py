"""GPU parity tests of the fused streaming kernels (mgx_fused_down / mgx_fused_up)
against compositions of the oracle's operators, bit for bit, over grid sizes
that exercise one strip, several strips, partial strips and chunk boundaries."""
import numpy as np
import pytest

from conftest import assert_bits

pytestmark = pytest.mark.gpu

SIZES = [3, 5, 9, 17, 33, 65, 67, 119, 121, 123, 125, 127, 129, 131, 241, 243, 245, 247, 257, 363, 513, 1025, 2049]


def _rand(m, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    return np.asfortranarray(rng.standard_normal((m, m)) * scale)


@pytest.mark.parametrize("m", SIZES)
@pytest.mark.parametrize("ns", [0, 1, 2, 3, 4, 5, 9])
def test_fused_down(mg, O, m, ns):
    u, f = _rand(m, m + ns), _rand(m, m + ns + 1, 30.0)
    dx = 1.0 / (m - 1)
    omega = 0.6 if ns % 2 == 0 else 0.85
    p = mg.gridtype(mg.to_device(f), mg.to_device(u), dx)
    c = mg.fused_down(p, ns, omega)
    u_ref = O.relax(u, f, dx, ns, omega)
    c_ref = O.coarsen(O.residual(u_ref, f, dx))
    assert_bits(mg.to_numpy(p.v), u_ref, "v after fused down m=%d ns=%d" % (m, ns))
    assert_bits(mg.to_numpy(c), c_ref, "coarse f after fused down m=%d ns=%d" % (m, ns))


@pytest.mark.parametrize("m", SIZES)
@pytest.mark.parametrize("ns", [0, 1, 2, 3, 4, 6])
def test_fused_up(mg, O, m, ns):
    mc = (m - 1) // 2 + 1
    u, f, c = _rand(m, 2 * m + ns), _rand(m, 2 * m + ns + 1, 30.0), _rand(mc, 2 * m + ns + 2)
    # the coarse iterate has a zero boundary in the solver (main.cpp:295-299)
    c[0, :] = c[-1, :] = 0.0
    c[:, 0] = c[:, -1] = 0.0
    dx = 1.0 / (m - 1)
    omega = 0.6
    p = mg.gridtype(mg.to_device(f), mg.to_device(u), dx)
    nrm = mg.fused_up(mg.to_device(c), p, ns, omega, norm=True)
    v_ref = u.copy(order="F")
    fine = O.refine(c, m)
    O.lib().mgo_plus_equals(v_ref.ctypes.data_as(O._dp), fine.ctypes.data_as(O._dp), v_ref.size)
    v_ref = O.relax(v_ref, f, dx, ns, omega)
    assert_bits(mg.to_numpy(p.v), v_ref, "v after fused up m=%d ns=%d" % (m, ns))
    assert nrm == O.calc_norm(O.residual(v_ref, f, dx))
    # and without the norm
    p2 = mg.gridtype(mg.to_device(f), mg.to_device(u), dx)
    assert mg.fused_up(mg.to_device(c), p2, ns, omega) is None
    assert_bits(mg.to_numpy(p2.v), v_ref, "v after fused up (no norm) m=%d ns=%d" % (m, ns))


def test_fused_up_nonzero_coarse_boundary(mg, O):
    """refine() zeroes the fine boundary whatever the coarse boundary holds
    (main.cpp:229-232); the fused kernel must not leak it into the fine boundary."""
    m, mc = 65, 33
    u, f, c = _rand(m, 1), _rand(m, 2), _rand(mc, 3)
    dx = 1.0 / 64
    p = mg.gridtype(mg.to_device(f), mg.to_device(u), dx)
    mg.fused_up(mg.to_device(c), p, 2, 0.6)
    v_ref = u.copy(order="F")
    fine = O.refine(c, m)
    O.lib().mgo_plus_equals(v_ref.ctypes.data_as(O._dp), fine.ctypes.data_as(O._dp), v_ref.size)
    v_ref = O.relax(v_ref, f, dx, 2, 0.6)
    assert_bits(mg.to_numpy(p.v), v_ref, "nonzero coarse boundary")
