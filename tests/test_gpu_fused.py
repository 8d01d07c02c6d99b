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


@pytest.mark.gpu
def test_two_column_ranges_in_one_launch(O):
    """mgx_slab_op.jlo2/jhi2: one launch that produces two bands of columns (the columns next to
    both neighbours of a slab, launched ahead of the interior) equals two launches of one band each,
    and together with the interior launch the whole-range launch, bit for bit -- for every kind of
    leg that a partitioned level-0 launch runs, with the norm merged by both."""
    import ctypes
    import torch
    from multigrid_amd import _lib
    L = _lib.load()
    m, ncols = 2049, 300
    rng = np.random.default_rng(11)
    dev = torch.device("cuda")
    f = torch.from_numpy(rng.standard_normal((ncols, m))).to(dev)
    vin = torch.from_numpy(rng.standard_normal((ncols, m))).to(dev)
    mc, nccols = (m - 1) // 2 + 1, 160
    cv = torch.from_numpy(rng.standard_normal((nccols, mc))).to(dev)
    gf0, gc0 = 4000, 1996           # local column 0 is global fine column 4000; coarse column 1996 <-> fine 3992
    jlo, jhi, B = 12, 288, 40

    def launch(ns, pa, nm, res, store, ranges):
        vout = torch.zeros_like(vin)
        cf = torch.zeros((nccols, mc), dtype=torch.float64, device=dev)
        slot = torch.zeros(1, dtype=torch.int64, device=dev)
        for rg in ranges:
            o = _lib.mgx_slab_op()
            o.struct_size = ctypes.sizeof(_lib.mgx_slab_op)
            o.nsweeps, o.pro_stage, o.norm_stage, o.restrict_out, o.zero_in, o.store = ns, pa, nm, res, 0, store
            o.v_in, o.v_out, o.f = vin.data_ptr(), vout.data_ptr(), f.data_ptr()
            o.coarse_v, o.coarse_f, o.norm_slot = cv.data_ptr(), cf.data_ptr(), slot.data_ptr()
            o.m, o.ld, o.ncols, o.mc, o.ldc, o.nccols = m, m, ncols, mc, mc, nccols
            o.jlo, o.jhi = rg[0]
            if len(rg) > 1:
                o.jlo2, o.jhi2 = rg[1]
            o.jfix_lo = o.jfix_hi = -1
            o.gf0, o.gc0, o.dx, o.omega = gf0, gc0, 1.0 / 4096, 0.6
            _lib.check(L.mgx_slab_launch(ctypes.byref(o), None))
        torch.cuda.synchronize()
        return vout.cpu().numpy(), cf.cpu().numpy(), int(slot.item())

    lo, hi, mid = (jlo, jlo + B), (jhi - B, jhi), (jlo + B, jhi - B)
    for (ns, pa, nm, res, store) in ((4, 0, 2, 1, 1), (2, -1, -1, 1, 1), (2, 0, 2, 0, 1), (0, -1, 0, 0, 0), (3, 0, 1, 1, 1)):
        whole = launch(ns, pa, nm, res, store, [((jlo, jhi),)])
        split = launch(ns, pa, nm, res, store, [(lo, hi), (mid,)])        # two bands in one launch + the interior
        three = launch(ns, pa, nm, res, store, [(lo,), (hi,), (mid,)])    # three launches
        only2 = launch(ns, pa, nm, res, store, [((jlo, jlo), hi), (lo,), (mid,)])   # an empty first range
        for other, what in ((split, "two ranges + interior"), (three, "three launches"), (only2, "empty first range")):
            for a, b, name in zip(whole, other, ("v_out", "coarse_f", "norm")):
                assert np.array_equal(np.asarray(a).view(np.uint64) if name != "norm" else a,
                                      np.asarray(b).view(np.uint64) if name != "norm" else b), (ns, pa, nm, res, what, name)
    # a second range that overlaps or precedes the first is refused
    o = _lib.mgx_slab_op()
    o.struct_size = ctypes.sizeof(_lib.mgx_slab_op)
    o.nsweeps, o.pro_stage, o.norm_stage, o.restrict_out, o.store = 2, -1, -1, 0, 1
    o.v_in, o.v_out, o.f = vin.data_ptr(), torch.zeros_like(vin).data_ptr(), f.data_ptr()
    o.m, o.ld, o.ncols, o.jlo, o.jhi, o.jlo2, o.jhi2 = m, m, ncols, 50, 100, 90, 120
    o.jfix_lo = o.jfix_hi = -1
    o.dx, o.omega = 1.0, 0.6
    assert L.mgx_slab_launch(ctypes.byref(o), None) == -1 and b"second range" in L.mgx_last_error()
