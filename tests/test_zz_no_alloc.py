"""mgx_residual_norm_noalloc (the entry point that allocates nothing: its partial results
live in a fixed device array inside the library) against mgx_residual_norm, through the C
ABI, bit for bit, on two calls in a row, the second on a smaller grid."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(mg, L, v, f, m, dx, out_dev):
    from multigrid_amd import _lib
    import torch
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    out_dev.fill_(-7.0)
    _lib.check(L.mgx_residual_norm_noalloc(v.data_ptr(), f.data_ptr(), m, dx, out_dev.data_ptr(), stream))
    ref = ctypes.c_double(0.0)
    _lib.check(L.mgx_residual_norm(v.data_ptr(), f.data_ptr(), m, dx, ctypes.byref(ref), stream))
    got = float(out_dev.cpu().item())
    return got, ref.value


def test_no_alloc_residual_norm_matches_the_allocating_entry_point(mg, O):
    import torch
    from multigrid_amd import _lib, rhs
    L = _lib.load()
    out_dev = torch.zeros(1, dtype=torch.float64, device="cuda")
    # a large grid with a large residual first, then smaller grids with smaller residuals:
    # nothing of an earlier call may leak into a later one
    seq = [(4097, 1e3), (1025, 1.0), (129, 1e-3), (3, 1.0), (2049, 1e-6), (5, 1e2)]
    for m, scale in seq:
        dx = 1.0 / (m - 1)
        g = torch.Generator(device="cuda").manual_seed(m)
        f = rhs.random_torch(m) * scale
        v = mg.colmajor(m, m)
        v.copy_(torch.rand((m, m), generator=g, device="cuda", dtype=torch.float64) * scale * dx * dx)
        got, ref = _both(mg, L, v, f, m, dx, out_dev)
        assert np.float64(got).view(np.uint64) == np.float64(ref).view(np.uint64), (m, got, ref)
        assert got > 0
        if m <= 1025:  # and against the CPU oracle
            want = O.calc_norm(O.residual(mg.to_numpy(v), mg.to_numpy(f), dx))
            assert got == want, (m, got, want)
    # NaNs are ignored (DTDoubleArray.cpp:883-899,923-939): both entry points agree
    m = 257
    f = rhs.random_torch(m)
    f[17, 40] = float("nan")
    v = mg.colmajor(m, m)
    got, ref = _both(mg, L, v, f, m, 1.0 / (m - 1), out_dev)
    assert got == ref and np.isfinite(got)


def test_no_alloc_rejects_what_it_cannot_hold(mg):
    import torch
    from multigrid_amd import _lib
    L = _lib.load()
    t = torch.zeros(16, dtype=torch.float64, device="cuda")
    assert L.mgx_residual_norm_noalloc(t.data_ptr(), t.data_ptr(), 65539, 1.0, t.data_ptr(), None) == -1
    assert b"exceeds" in L.mgx_last_error()
    assert L.mgx_residual_norm_noalloc(t.data_ptr(), t.data_ptr(), 4, 1.0, t.data_ptr(), None) == -1
    assert L.mgx_residual_norm_noalloc(None, t.data_ptr(), 5, 1.0, t.data_ptr(), None) == -1


def test_no_alloc_under_graph_capture(mg):
    """The call is capturable: no allocation, no synchronisation.  The replayed graph gives
    the norm of whatever v holds at replay time."""
    import torch
    from multigrid_amd import _lib, rhs
    L = _lib.load()
    m = 513
    dx = 1.0 / (m - 1)
    f = rhs.random_torch(m)
    v = mg.colmajor(m, m)
    out_dev = torch.zeros(1, dtype=torch.float64, device="cuda")
    # first use outside capture (looks up the scratch address of this device)
    _lib.check(L.mgx_residual_norm_noalloc(v.data_ptr(), f.data_ptr(), m, dx, out_dev.data_ptr(),
                                           ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        _lib.check(L.mgx_residual_norm_noalloc(v.data_ptr(), f.data_ptr(), m, dx, out_dev.data_ptr(),
                                               ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    for scale in (1.0, 0.25):
        v.fill_(0.0)
        v[1:-1, 1:-1] = scale
        g.replay()
        torch.cuda.synchronize()
        ref = ctypes.c_double(0.0)
        _lib.check(L.mgx_residual_norm(v.data_ptr(), f.data_ptr(), m, dx, ctypes.byref(ref), None))
        assert float(out_dev.item()) == ref.value
