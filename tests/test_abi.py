"""CPU tests of the C-ABI library: it loads, exports every symbol that
include/mgx.h declares, and its argument checks answer before any GPU call."""
import ctypes
import os
import re

import numpy as np

from conftest import ROOT


def _lib():
    from multigrid_amd import _lib
    return _lib


def test_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "mgx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mgx_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    L = _lib().load()
    for name in sorted(declared):
        assert hasattr(L, name), "libmgx.so does not export %s" % name
    # and the Python binding table covers the header exactly
    assert declared == set(_lib().SIGNATURES), declared ^ set(_lib().SIGNATURES)


def test_version_and_options_layout():
    L = _lib().load()
    assert L.mgx_version() == 210
    assert ctypes.sizeof(_lib().mgx_options) == 4 + 4 + 8 + 4 + 4 + 32


def test_depth_matches_oracle(O):
    L = _lib().load()
    for m in (3, 5, 9, 17, 33, 65, 101, 129, 257, 1025, 4097, 16385, 32769):
        for c in (1, 2, 3, 4, 8, 11):
            assert L.mgx_depth(m, c) == O.depth(m, c), (m, c)


def test_argument_errors_need_no_gpu():
    lib = _lib()
    L = lib.load()
    a = np.zeros((4, 5), order="F")
    rc = L.mgx_solve2d(a.ctypes.data, a.ctypes.data, 4, 5, 1.0, 1, 1, 1, 0.6, 2, 0, None, None, None)
    assert rc == -1 and b"not square" in L.mgx_last_error()  # main.cpp:410
    b = np.zeros((4, 4), order="F")
    rc = L.mgx_solve2d(b.ctypes.data, b.ctypes.data, 4, 4, 1.0, 1, 1, 1, 0.6, 2, 0, None, None, None)
    assert rc == -1 and b"even rows and columns" in L.mgx_last_error()  # main.cpp:415
    c = np.zeros((7, 7), order="F")
    rc = L.mgx_solve2d(c.ctypes.data, c.ctypes.data, 7, 7, 1.0, 1, 1, 1, 0.6, 2, 0, None, None, None)
    assert rc == -2  # 7 -> 4: even coarse level
    rc = L.mgx_solve2d(None, c.ctypes.data, 7, 7, 1.0, 1, 1, 1, 0.6, 2, 0, None, None, None)
    assert rc == -1
    assert L.mgx_relax(None, None, 9, 1.0, 1, 0.6, None) == -1
    assert L.mgx_restrict(c.ctypes.data, 7, c.ctypes.data, 5, None) == -1  # 5 does not pair with 7
    assert L.mgx_direct3x3(c.ctypes.data, c.ctypes.data, 5, 1.0, None) == 1  # main.cpp:167, not an error


def test_rhs_generators_numpy():
    from multigrid_amd import rhs
    f = rhs.smooth_np(17)
    assert f.flags.f_contiguous and f[0, 0] == 0.0 and f[16, 16] == -(2.0 * 2.5)
    r = rhs.random_np(17)
    assert r.min() == 0.0 and 0.0 < r.max() < 1.0
    assert np.all(r[0, :] == 0) and np.all(r[:, -1] == 0)
    assert np.array_equal(r, rhs.random_np(17))
    t = rhs.random_torch(17, device="cpu").numpy()
    assert np.array_equal(r, t)
    assert np.array_equal(f, rhs.smooth_torch(17, device="cpu").numpy())


def test_slab_op_accepts_the_struct_without_the_second_range():
    """mgx_slab_op grew a second column range (jlo2, jhi2) at its end; a caller built against the
    older struct passes the smaller struct_size and must get past the size check (the next check,
    a null input, answers instead); any other size is refused.  No GPU call is made."""
    lib = _lib()
    L = lib.load()
    op = lib.mgx_slab_op()
    full = ctypes.sizeof(lib.mgx_slab_op)
    old = lib.mgx_slab_op.jlo2.offset
    assert full - old == 8
    for size, expect in ((full, b"null input"), (old, b"null input"), (old - 4, b"struct_size"), (full + 8, b"struct_size")):
        op.struct_size = size
        op.m, op.ld, op.ncols = 9, 9, 9
        op.nsweeps, op.pro_stage, op.norm_stage, op.store = 1, -1, -1, 1
        assert L.mgx_slab_launch(ctypes.byref(op), None) == -1
        assert expect in L.mgx_last_error(), (size, L.mgx_last_error())


def test_tune_report_is_empty_before_anything_ran():
    lib = _lib()
    assert lib.tune_report() == [] or all("k_stream<" in l for l in lib.tune_report())
    assert lib.load().mgx_tune_report(None, 0) >= 1
