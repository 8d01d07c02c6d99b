"""CPU tests of the host logic of the partitioned plan (csrc/mgx_multi.hip): which levels are
split, which columns a slab owns and how wide the recomputed halos are.  The library is loaded
but nothing runs on a GPU.  The C++ results are compared with the Python driver's
(multigrid_amd/slab.py), which the gloo tests exercise end to end, and checked against the
invariants the schedule relies on."""
import ctypes

import pytest

from multigrid_amd import _lib, slab


def _partition(L, M, coarsest, R, tail_m):
    nd = ctypes.c_int(0)
    rc = L.mgx_mplan_partition(M, coarsest, R, tail_m, ctypes.byref(nd), None, 0)
    if rc != 0:
        return rc, None, None
    n = (nd.value + 1) * (R + 1)
    buf = (ctypes.c_int * n)()
    assert L.mgx_mplan_partition(M, coarsest, R, tail_m, ctypes.byref(nd), buf, n) == 0
    b = [[buf[l * (R + 1) + g] for g in range(R + 1)] for l in range(nd.value + 1)]
    return 0, nd.value, b


@pytest.mark.parametrize("nd", [1, 2, 3, 4, 5, 6])
def test_halo_plan_equals_the_python_driver(nd):
    L = _lib.load()
    for a in range(5):
        for b in range(5 - a):
            if a + b == 0:
                continue
            Ev = (ctypes.c_int * (nd + 1))()
            Ef = (ctypes.c_int * (nd + 1))()
            assert L.mgx_halo_plan(nd, a, b, Ev, Ef) == 0
            ev, ef = slab.halo_plan(nd, a, b)
            assert list(Ev) == ev and list(Ef) == ef, (nd, a, b)
    assert L.mgx_halo_plan(nd, 3, 2, Ev, Ef) < 0      # more than 4 sweeps: not a lean schedule
    assert L.mgx_halo_plan(nd, 0, 0, Ev, Ef) < 0


@pytest.mark.parametrize("M,coarsest,R,tail_m", [
    (16385, 2, 1, 0), (16385, 2, 2, 0), (16385, 2, 4, 0), (16385, 2, 8, 0), (16385, 2, 8, 513), (32769, 2, 4, 0),
    (22529, 11, 2, 0), (45057, 11, 8, 0), (4097, 2, 3, 257), (2049, 2, 5, 129), (705, 11, 2, 89), (193, 6, 6, 129),
    (513, 2, 7, 65), (257, 2, 64, 33), (65537, 2, 8, 0),
])
def test_partition_invariants(M, coarsest, R, tail_m):
    L = _lib.load()
    rc, nd, b = _partition(L, M, coarsest, R, tail_m)
    if rc != 0:
        assert b"is not split" in L.mgx_last_error()
        return
    depth = L.mgx_depth(M, coarsest)
    dims = [M]
    for _ in range(depth):
        dims.append((dims[-1] - 1) // 2 + 1)
    assert 1 <= nd <= depth
    used = tail_m or (2049 if R == 1 else 1025)
    assert dims[nd - 1] > used                       # every distributed level is above the threshold ...
    if R == 1:
        assert dims[nd] <= used or nd == depth - (dims[-1] % 2 == 0)
    Ev, Ef = slab.view_capacity(nd)
    for l in range(nd + 1):
        assert b[l][0] == 0 and b[l][R] == dims[l]
        assert all(b[l][g] < b[l][g + 1] for g in range(R))       # every slab owns columns, in order
    for l in range(nd):
        for g in range(1, R):
            assert b[l][g] == 2 * b[l + 1][g] - 1                  # owner of fine 2J owns coarse J
        need = 8 if l == 0 else max(8, Ef[l], Ev[l])
        if R > 1:
            assert min(b[l][g + 1] - b[l][g] for g in range(R)) >= need   # a halo's worth of own columns


def test_partition_errors():
    L = _lib.load()
    assert L.mgx_mplan_partition(16384, 2, 2, 0, None, None, 0) < 0 and b"even" in L.mgx_last_error()
    assert L.mgx_mplan_partition(65, 2, 8, 0, None, None, 0) < 0 and b"is not split" in L.mgx_last_error()
    assert L.mgx_mplan_partition(16385, 2, 0, 0, None, None, 0) < 0
    assert L.mgx_mplan_partition(91, 2, 2, 0, None, None, 0) < 0    # 91, 46, 23: an even level above the coarsest
    nd = ctypes.c_int(0)
    small = (ctypes.c_int * 3)()
    assert L.mgx_mplan_partition(16385, 2, 8, 0, ctypes.byref(nd), small, 3) < 0 and b"bounds needs" in L.mgx_last_error()
    assert nd.value == 4
