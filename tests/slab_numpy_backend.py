"""TEST INFRASTRUCTURE: a CPU backend for multigrid_amd.slab.SlabSolver.

It lets the multi-rank driver (ownership, halo exchange, tail all-gather, norm
reduction) run under gloo without a GPU.  One fused slab launch is emulated
with whole-array numpy operations in the reference's evaluation order
(elementwise IEEE operations, no FMA), the tail runs through the CPU oracle.
Arrays have the slab layout of the product backend: shape (ncols, m), row k of
the array is column k of the slab, so a[j, i] is grid point (i, j).

Never imported by the product package.
"""
import numpy as np
import torch


def _encode(v):
    """Order-preserving integer image of a norm (0 <-> nothing merged), as libmgx encodes it."""
    if v == -np.inf:
        return 0
    return int(np.array([v], dtype=np.float64).view(np.uint64)[0]) + 1


def _decode(k):
    if k == 0:
        return -np.inf
    return float(np.array([k - 1], dtype=np.uint64).view(np.float64)[0])


def _jacobi(u, f, h2, omega, nomega):
    """One sweep on every column that has both neighbours; rows 1..m-2 (main.cpp:194-195)."""
    s = u[1:-1, :-2] + u[1:-1, 2:]        # W + E  (rows i-1, i+1)
    s = s + u[:-2, 1:-1]                  # + S    (column j-1)
    s = s + u[2:, 1:-1]                   # + N    (column j+1)
    s = s - f[1:-1, 1:-1] * h2
    s = s * 0.25
    s = s * omega
    return u[1:-1, 1:-1] * nomega + s


def _residual(u, f, invh2):
    s = u[1:-1, :-2] + u[1:-1, 2:]        # main.cpp:277
    s = s + u[:-2, 1:-1]
    s = s + u[2:, 1:-1]
    s = s - u[1:-1, 1:-1] * 4.0
    s = s * invh2
    return f[1:-1, 1:-1] - s


class NumpyBackend:
    name = "numpy"
    device = None

    def __init__(self, oracle, coarse_exact=False):
        self.O = oracle
        self.coarse_exact = coarse_exact

    def zeros(self, ncols, m, phase=0):
        return np.zeros((ncols, m))

    def slots(self, n):
        return torch.zeros(n, dtype=torch.int64)

    def from_numpy(self, a):
        return np.array(a, dtype=np.float64, order="C", copy=True)

    def to_numpy(self, t):
        return np.asarray(t)

    def comm_tensor(self, t):
        return torch.from_numpy(t) if isinstance(t, np.ndarray) else t

    def index_array(self, rows):
        return np.asarray(rows, dtype=np.int64)

    def take_rows(self, src, rows, out):
        out[:] = np.asarray(src)[rows]

    def slot_ptr(self, slots, k):
        return (slots, k)

    def decode_slots(self, slots):
        return np.array([_decode(int(k)) for k in slots.numpy()])

    def synchronize(self):
        pass

    def make_tail(self, m, dx, coarsest):
        return _OracleTail(self.O, m, dx, coarsest, self.coarse_exact)

    def _correction(self, op, ncols, m):
        c = op.coarse_v
        j = np.arange(ncols)
        jg = j + op.gf0
        J = np.clip((jg >> 1) - op.gc0, 0, op.nccols - 1)
        J1 = np.clip(J + 1, 0, op.nccols - 1)
        odd_j = (jg & 1) == 1
        i = np.arange(m)
        I = i >> 1
        I1 = np.clip(I + 1, 0, op.mc - 1)
        odd_i = (i & 1) == 1
        c00 = c[J][:, I]
        c10 = c[J][:, I1]      # row I+1, column J
        c01 = c[J1][:, I]      # row I,   column J+1
        c11 = c[J1][:, I1]
        p_ee = c00                                   # main.cpp:236
        p_eo = 0.5 * (c00 + c01)                     # i even, j odd  (:239)
        p_oe = 0.5 * (c00 + c10)                     # i odd,  j even (:242)
        s4 = c00 + c10
        s4 = s4 + c01
        s4 = s4 + c11
        p_oo = 0.25 * s4                             # :245
        return np.where(odd_i[None, :], np.where(odd_j[:, None], p_oo, p_oe), np.where(odd_j[:, None], p_eo, p_ee))

    def launch(self, op):
        m, ncols = op.m, op.ncols
        h2 = op.dx * op.dx
        invh2 = 1.0 / h2
        omega, nomega = op.omega, 1 - op.omega
        fixed = [j for j in (op.jfix_lo, op.jfix_hi) if j is not None and j >= 0]
        # the produced columns: one range, or two (the columns next to both neighbours in one launch)
        ranges = [(op.jlo, op.jhi)]
        if getattr(op, "jhi2", None) and op.jhi2 > op.jlo2:
            ranges.append((op.jlo2, op.jhi2))
        u = np.zeros((ncols, m)) if op.zero_in else np.array(op.v_in, copy=True)
        for s in range(op.nsweeps + 1):
            if s > 0:
                new = u.copy()
                new[1:-1, 1:-1] = _jacobi(u, op.f, h2, omega, nomega)
                for jf in fixed:
                    new[jf, :] = u[jf, :]
                u = new
            if op.pro_stage == s:
                add = u + self._correction(op, ncols, m)     # main.cpp:341
                mask = np.ones((ncols, m), dtype=bool)
                mask[:, 0] = mask[:, -1] = False
                for jf in fixed:
                    mask[jf, :] = False
                u = np.where(mask, add, u)
            if op.norm_stage == s:
                r = np.zeros((ncols, m))
                r[1:-1, 1:-1] = _residual(u, op.f, invh2)
                slots, k = op.norm_slot
                for a, b in ranges:
                    val = float(np.max(np.abs(r[a:b, 1:m - 1]))) if b > a else 0.0
                    slots[k] = max(int(slots[k]), _encode(max(val, 0.0)))
        if op.store:
            for a, b in ranges:
                op.v_out[a:b, 1:m - 1] = u[a:b, 1:m - 1]
        if op.restrict_out:
            r = np.zeros((ncols, m))
            r[1:-1, 1:-1] = _residual(u, op.f, invh2)
            cf = op.coarse_f
            for j in [j for a, b in ranges for j in range(a, b)]:
                jg = j + op.gf0
                if jg & 1:
                    continue
                Jc = (jg >> 1) - op.gc0
                I = np.arange(1, op.mc - 1)
                a = r[j, 2 * I] * 0.25                                  # main.cpp:216-218
                nb = r[j, 2 * I - 1] + r[j, 2 * I + 1]
                nb = nb + r[j - 1, 2 * I]
                nb = nb + r[j + 1, 2 * I]
                nb = nb * 0.125
                cn = r[j - 1, 2 * I - 1] + r[j - 1, 2 * I + 1]
                cn = cn + r[j - 1, 2 * I + 1]
                cn = cn + r[j + 1, 2 * I + 1]
                cn = cn * 0.0625
                cf[Jc, 1:op.mc - 1] = (a + nb) + cn


class _OracleTail:
    def __init__(self, O, m, dx, coarsest, coarse_exact=False):
        self.O, self.m, self.dx, self.coarsest, self.coarse_exact = O, m, dx, coarsest, coarse_exact
        self.f = np.zeros((m, m))  # slab layout: row k = column k
        self.v = np.zeros((m, m))

    def run(self, Ndown, Nup, omega):
        f = np.asfortranarray(self.f.T)
        sol, _, _ = self.O.multigrid(f, np.zeros((self.m, self.m), order="F"), self.dx, 1, Ndown, Nup, omega,
                                     self.coarsest, coarse_exact=self.coarse_exact)
        self.v[:] = sol.T
        return self.v

    def fmg(self, ncyc, Ndown, Nup, omega):
        f = np.asfortranarray(self.f.T)
        sol, _ = self.O.fmg(f, np.asfortranarray(self.v.T), self.dx, ncyc, Ndown, Nup, omega, self.coarsest,
                            coarse_exact=self.coarse_exact)
        self.v[:] = sol.T
        return self.v

    def reset(self):
        self.v[:] = 0.0

    def close(self):
        pass
