"""Synthetic right-hand sides for the benchmark and the parity tests.

Both generators are libm-free (only IEEE +, * and integer arithmetic), so the
numpy version on the host and the torch version on the GPU produce the same
bits.  Domain [0,1]^2, x = i*dx, y = j*dx, dx = 1/(M-1), the recipe of the
reference's MATLAB tester (MGTester.m:130-147):

  smooth : f = -(x^2+y^2) * (1 + xy + (xy)^2/2)      (polynomial stand-in for the
           tester's -(x^2+y^2)*exp(xy), MGTester.m:140; evaluated at every i,j)
  random : interior f = u01(splitmix64(seed ^ (i + j*M))), 0 on the boundary
           (the tester's rand(N-2,N-2), MGTester.m:144)
"""
import numpy as np

SEED = 0x9E3779B97F4A7C15
_M64 = (1 << 64) - 1


def smooth_np(M):
    dx = 1.0 / (M - 1)
    x = np.arange(M, dtype=np.float64) * dx
    X = x[:, None]
    Y = x[None, :]
    xy = X * Y
    r2 = X * X + Y * Y
    p = 1.0 + xy
    q = xy * xy
    q = q * 0.5
    p = p + q
    return np.asfortranarray(-(r2 * p))


def _splitmix64_np(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15))
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def random_np(M, seed=SEED):
    i = np.arange(M, dtype=np.uint64)[:, None]
    j = np.arange(M, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        k = i + j * np.uint64(M)
        z = _splitmix64_np(np.uint64(seed) ^ k)
    u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)  # 53-bit mantissa -> [0,1)
    u[0, :] = 0.0
    u[-1, :] = 0.0
    u[:, 0] = 0.0
    u[:, -1] = 0.0
    return np.asfortranarray(u)


def smooth_torch(M, device="cuda"):
    """Column-major M x M GPU tensor, bit-identical to smooth_np(M)."""
    import torch
    dx = 1.0 / (M - 1)
    x = torch.arange(M, dtype=torch.float64, device=device) * dx
    # build as (j, i) row-major == column-major (i, j)
    Yj = x[:, None]
    Xi = x[None, :]
    xy = Xi * Yj
    r2 = Xi * Xi
    r2 = r2 + Yj * Yj
    p = xy + 1.0
    q = xy * xy
    q = q * 0.5
    p = p + q
    out = r2 * p
    out = -out
    return out.contiguous().t()


def _lsr(z, k):
    """Logical shift right of int64 tensors."""
    return (z >> k) & ((1 << (64 - k)) - 1)


def _wrap(c):
    """Python int -> the int64 with the same low 64 bits."""
    c &= _M64
    return c - (1 << 64) if c >= (1 << 63) else c


def random_torch(M, seed=SEED, device="cuda"):
    """Column-major M x M GPU tensor, bit-identical to random_np(M, seed)."""
    import torch
    i = torch.arange(M, dtype=torch.int64, device=device)[None, :]
    j = torch.arange(M, dtype=torch.int64, device=device)[:, None]
    k = i + j * M
    z = k ^ _wrap(seed)
    z = z + _wrap(0x9E3779B97F4A7C15)
    z = (z ^ _lsr(z, 30)) * _wrap(0xBF58476D1CE4E5B9)
    z = (z ^ _lsr(z, 27)) * _wrap(0x94D049BB133111EB)
    z = z ^ _lsr(z, 31)
    u = _lsr(z, 11).to(torch.float64) * (1.0 / 9007199254740992.0)
    u[0, :] = 0.0
    u[-1, :] = 0.0
    u[:, 0] = 0.0
    u[:, -1] = 0.0
    return u.contiguous().t()


def random_torch_cols(M, j0, j1, seed=SEED, device="cuda"):
    """Columns j0..j1-1 of random_np(M, seed) as a (j1-j0, M) GPU tensor whose row k
    is column j0+k of the grid (the slab layout of multigrid_amd.slab)."""
    import torch
    i = torch.arange(M, dtype=torch.int64, device=device)[None, :]
    j = torch.arange(j0, j1, dtype=torch.int64, device=device)[:, None]
    k = i + j * M
    z = k ^ _wrap(seed)
    z = z + _wrap(0x9E3779B97F4A7C15)
    z = (z ^ _lsr(z, 30)) * _wrap(0xBF58476D1CE4E5B9)
    z = (z ^ _lsr(z, 27)) * _wrap(0x94D049BB133111EB)
    z = z ^ _lsr(z, 31)
    u = _lsr(z, 11).to(torch.float64) * (1.0 / 9007199254740992.0)
    u[:, 0] = 0.0
    u[:, -1] = 0.0
    if j0 == 0:
        u[0, :] = 0.0
    if j1 == M:
        u[-1, :] = 0.0
    return u.contiguous()
