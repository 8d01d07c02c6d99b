"""CPU tests of the host logic of the grouped small-level launches (csrc/mgx_mid.hip): which
points of which level a workgroup computes for its tile of the output.  The library is loaded
but nothing runs on a GPU.  The kernels trust these index ranges blindly -- a range one cell
too narrow gives wrong numbers at tile edges, not a fault -- so the ranges are checked here by
replaying, along one axis, what the kernels do phase by phase with a flag per cell: "this value
is the one the whole-grid operator would give".  The operators are tensor products of their
one-axis footprints (the 5-point sweep moves along one axis at a time; restriction and
interpolation are products outright), so flags that stay true along each axis stay true in the
plane."""
import ctypes

import pytest

from multigrid_amd import _lib

LEAN = [(a, b) for a in range(5) for b in range(5 - a)]


def _sizes(m0, nl):
    m = [m0]
    for _ in range(nl):
        m.append((m[-1] + 1) // 2)
    return m


def _shape(L, down, nl, m0, nd, nu):
    info = _lib.mgx_mid_info()
    assert L.mgx_mid_shape(int(down), nl, m0, nd, nu, ctypes.byref(info)) == 0
    return info


def _spans(L, down, nl, m0, nd, nu, tile, t):
    n = 2 * (nl + 1)
    lo, hi = (ctypes.c_int * n)(), (ctypes.c_int * n)()
    assert L.mgx_mid_spans(int(down), nl, m0, nd, nu, tile, t, lo, hi) == 0, L.mgx_last_error()
    return list(lo), list(hi)


class Cells:
    """Flags of one level along one axis inside a region [lo, hi]; outside the region a cell
    reads as the zero border: right where that is the grid's boundary (index 0 or m-1 of an
    array whose boundary is zero), wrong anywhere else."""

    def __init__(self, m, lo, hi, flags=None):
        self.m, self.lo, self.hi = m, lo, hi
        self.ok = dict((x, True) for x in range(lo, hi + 1)) if flags is None else flags

    def __call__(self, x):
        if self.lo <= x <= self.hi:
            return self.ok[x]
        assert self.lo - 1 <= x <= self.hi + 1, "read outside the array"
        return x == 0 or x == self.m - 1

    def stencil(self, extra=None):
        """one sweep or residual: a cell needs its neighbours, itself (and f, `extra`)"""
        new = {}
        for x in range(self.lo, self.hi + 1):
            new[x] = self(x - 1) and self(x) and self(x + 1) and (extra is None or extra(x))
        return Cells(self.m, self.lo, self.hi, new)


def _replay_down(m, nd, need, own_next):
    """f of level k on need[k] -> f of level k+1 on need[k+1], all of it right."""
    nl = len(m) - 1
    f = Cells(m[0], *need[0])
    for k in range(nl):
        lo, hi = need[k]
        assert 1 <= lo <= hi <= m[k] - 2
        if nd == 0:
            u = Cells(m[k], lo, hi, dict((x, True) for x in range(lo, hi + 1)))   # zero: right everywhere ...
            u_everywhere = True
        else:
            u = Cells(m[k], lo, hi, dict((x, f(x)) for x in range(lo, hi + 1)))   # first sweep: f at the point only
            u_everywhere = False
            for _ in range(nd - 1):
                u = u.stencil(f)
        if u_everywhere:
            r = Cells(m[k], lo, hi, dict((x, f(x)) for x in range(lo, hi + 1)))   # ... so r = f - 0 needs f only
        else:
            r = u.stencil(f)
        clo, chi = need[k + 1]
        nxt = {}
        for I in range(clo, chi + 1):
            assert lo - 1 <= 2 * I - 1 and 2 * I + 1 <= hi + 1
            nxt[I] = r(2 * I - 1) and r(2 * I) and r(2 * I + 1)
            assert nxt[I], "level %d: restricted value at %d is wrong" % (k + 1, I)
        olo, ohi = own_next[k + 1]
        assert clo <= olo and ohi <= chi                     # what is written was computed
        f = Cells(m[k + 1], clo, chi, nxt)


@pytest.mark.parametrize("m0,nl", [(513, 2), (257, 2), (1025, 1), (1025, 3), (129, 2), (705, 3), (1153, 4), (37, 2), (9, 2), (5, 1)])
def test_down_tiles_cover_what_they_need(m0, nl):
    L = _lib.load()
    m = _sizes(m0, nl)
    for nd in range(5):
        info = _shape(L, True, nl, m0, nd, 0)
        tiles = [1, 2, 3, 7] + ([info.tile_i, info.tile_j] if info.ok else [])
        for tile in sorted(set(tiles)):
            nt = (m[nl] - 2 + tile - 1) // tile
            owned = [[] for _ in range(nl + 1)]
            for t in range(nt):
                lo, hi = _spans(L, True, nl, m0, nd, 0, tile, t)
                need = [(lo[k], hi[k]) for k in range(nl + 1)]
                own = [(lo[nl + 1 + k], hi[nl + 1 + k]) for k in range(nl + 1)]
                _replay_down(m, nd, need, own)
                for k in range(1, nl + 1):
                    owned[k].append(own[k])
            for k in range(1, nl + 1):                      # the tiles' written ranges partition every level
                assert owned[k][0][0] == 1 and owned[k][-1][1] == m[k] - 2
                for a, b in zip(owned[k], owned[k][1:]):
                    assert b[0] == a[1] + 1 and a[0] <= a[1]


def _replay_up(m, nd, nu, G, cload, own):
    nl = len(m) - 1
    clo, chi = cload
    assert 0 <= clo <= chi <= m[nl] - 1
    # level nl as loaded: every cell of cload, boundary included; nothing else is read
    coarse = dict((x, True) for x in range(clo, chi + 1))
    coarse_m = m[nl]
    loaded = True
    for k in range(nl - 1, -1, -1):
        lo, hi = G[k]
        assert 1 <= lo <= hi <= m[k] - 2

        def c_ok(y):
            if y in coarse:
                return coarse[y]
            assert not loaded, "level %d reads coarse cell %d that was not loaded" % (k, y)
            return y == 0 or y == coarse_m - 1           # an inner level's zero boundary sits in the border

        def corr_ok(x):
            return c_ok(x >> 1) and (x % 2 == 0 or c_ok((x >> 1) + 1))

        u = Cells(m[k], lo, hi)                           # zero, or the pointwise first sweep: right everywhere
        for _ in range(max(nd - 1, 0)):
            u = u.stencil()
        u = Cells(m[k], lo, hi, dict((x, u(x) and corr_ok(x)) for x in range(lo, hi + 1)))
        for _ in range(nu):
            u = u.stencil()
        if k > 0:
            nlo, nhi = G[k - 1]
            want = range(max(1, nlo >> 1), min(m[k] - 2, (nhi + 1) >> 1) + 1)
        else:
            want = range(own[0], own[1] + 1)
        for x in want:
            assert lo <= x <= hi and u(x), "level %d: value at %d is wrong" % (k, x)
        coarse, coarse_m, loaded = u.ok, m[k], False


@pytest.mark.parametrize("m0,nl", [(513, 4), (513, 3), (257, 2), (1025, 1), (1025, 5), (705, 3), (1153, 4), (37, 2), (9, 2), (5, 1)])
def test_up_tiles_cover_what_they_need(m0, nl):
    L = _lib.load()
    m = _sizes(m0, nl)
    for nd, nu in LEAN:
        info = _shape(L, False, nl, m0, nd, nu)
        tiles = [1, 2, 5, 16] + ([info.tile_i, info.tile_j] if info.ok else [])
        for tile in sorted(set(tiles)):
            nt = (m0 - 2 + tile - 1) // tile
            owned = []
            for t in range(nt):
                lo, hi = _spans(L, False, nl, m0, nd, nu, tile, t)
                G = [(lo[k], hi[k]) for k in range(nl)]
                own = (lo[nl + 1], hi[nl + 1])
                _replay_up(m, nd, nu, G, (lo[nl], hi[nl]), own)
                owned.append(own)
            assert owned[0][0] == 1 and owned[-1][1] == m0 - 2
            for a, b in zip(owned, owned[1:]):
                assert b[0] == a[1] + 1 and a[0] <= a[1]


def _lanes(w):
    return 16 if w <= 16 else (32 if w <= 32 else 64)


@pytest.mark.parametrize("down", [True, False])
@pytest.mark.parametrize("m0,nl", [(513, 2), (513, 4), (129, 2), (1025, 1), (257, 3), (705, 2), (2049, 1)])
def test_chosen_shapes_fit_the_kernels(down, m0, nl):
    """What the launcher picks: every region at most 64 cells along the lanes, at most
    points_per_thread passes along the other axis, and the arrays within the LDS it asks for."""
    L = _lib.load()
    m = _sizes(m0, nl)
    for nd, nu in LEAN:
        info = _shape(L, down, nl, m0, nd, nu)
        if not info.ok:
            continue
        assert info.threads % 64 == 0 and 64 <= info.threads <= 1024 and info.points_per_thread in (1, 2, 4)
        n_out = (m[nl] if down else m0) - 2
        assert info.ntiles_i == (n_out + info.tile_i - 1) // info.tile_i
        assert info.ntiles_j == (n_out + info.tile_j - 1) // info.tile_j
        nreg = nl + 1                                            # regions a workgroup walks: down need[0..nl]; up G[0..nl-1], cload
        wi = [[] for _ in range(nreg)]
        wj = [[] for _ in range(nreg)]
        for tile, nt, w in ((info.tile_i, info.ntiles_i, wi), (info.tile_j, info.ntiles_j, wj)):
            for t in range(nt):
                lo, hi = _spans(L, down, nl, m0, nd, nu, tile, t)
                for k in range(nreg):
                    w[k].append(hi[k] - lo[k] + 1)
        cells = 0
        for k in range(nreg):
            assert max(wi[k]) <= 64
            for a in set(wi[k]):
                dj = info.threads // _lanes(a)
                assert (max(wj[k]) + dj - 1) // dj <= info.points_per_thread
            if not (down and k == nl):                          # the down kernel keeps level nl in registers only
                cells = max(cells, (max(wi[k]) + 2) * (max(wj[k]) + 2))
        assert info.lds_bytes >= (2 if down else 3) * cells * 8
        assert info.lds_bytes <= 160 * 1024


def test_levels_that_do_not_nest_are_refused():
    L = _lib.load()
    info = _lib.mgx_mid_info()
    assert L.mgx_mid_shape(1, 2, 512, 2, 2, ctypes.byref(info)) == 0 and info.ok == 0     # even
    assert L.mgx_mid_shape(1, 3, 23, 2, 2, ctypes.byref(info)) == 0 and info.ok == 0      # 23 -> 12 -> 6.5
    assert L.mgx_mid_shape(1, 1, 23, 2, 2, ctypes.byref(info)) == 0 and info.ok == 1      # 23 -> 12 is fine
    assert L.mgx_mid_shape(1, 9, 513, 2, 2, ctypes.byref(info)) < 0
    lo, hi = (ctypes.c_int * 8)(), (ctypes.c_int * 8)()
    assert L.mgx_mid_spans(1, 2, 513, 2, 2, 4, 10 ** 6, lo, hi) < 0
