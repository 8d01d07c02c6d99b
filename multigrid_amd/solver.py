"""Python mirror of the reference's solver interface over libmgx (HIP).

Names, argument meaning and error behaviour follow the reference's main.cpp so
that tests read like tests of the reference:

    gridtype{f, v}                      main.cpp:29-33
    MGOutputs{ResidualNorms, Times}     main.cpp:35-40
    relax / residual / coarsen / refine / direct_solve / calcNorm / MultiGrid
                                        main.cpp:147-357

Arrays are torch float64 tensors on a GPU, indexed a[i, j] like the reference's
DTDoubleArray(i, j) and stored COLUMN-MAJOR (stride 1 along i), which is what
`colmajor()` / `to_device()` produce.  Everything here runs on the GPU through
the C ABI; nothing falls back to the CPU.
"""
import ctypes
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib


def _require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("multigrid_amd needs a HIP GPU (torch.cuda.is_available() is False); "
                           "there is no CPU path")


def colmajor(m, n, device="cuda", zero=True):
    """m x n float64 tensor with column-major storage (a[i, j] at i + j*m)."""
    base = (torch.zeros if zero else torch.empty)((n, m), dtype=torch.float64, device=device)
    return base.t()


def to_device(a, device="cuda"):
    """numpy (any order) -> column-major GPU tensor with the same [i, j] indexing."""
    a = np.asarray(a, dtype=np.float64)
    if a.ndim != 2:
        raise ValueError("expected a 2-D array")
    return torch.from_numpy(np.ascontiguousarray(a.T)).to(device).t()


def to_numpy(t):
    """column-major GPU tensor -> Fortran-ordered numpy array."""
    return np.asfortranarray(t.t().contiguous().cpu().numpy().T)


def _check_colmajor(t, name):
    if t.dtype != torch.float64 or t.dim() != 2 or not t.is_cuda:
        raise TypeError("%s must be a 2-D float64 GPU tensor" % name)
    m, n = t.shape
    if t.stride(0) != 1 or (n > 1 and t.stride(1) != m):
        raise ValueError("%s must be column-major (use multigrid_amd.colmajor/to_device)" % name)
    return m, n


def _stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


@dataclass
class gridtype:
    """main.cpp:29-33; dx replaces DTMutableMesh2D::Grid().dx()."""
    f: torch.Tensor
    v: torch.Tensor
    dx: float


@dataclass
class MGOutputs:
    """main.cpp:35-40."""
    ResidualNorms: np.ndarray
    Times: np.ndarray


def _square_odd(t, name):
    m, n = _check_colmajor(t, name)
    if m != n:
        raise ValueError("Error: Input is not square matrix!")       # main.cpp:410
    if m % 2 == 0:
        raise ValueError("Error: Input has even rows and columns!")  # main.cpp:415
    return m


def relax(p, Niter, omega):
    """main.cpp:171-199: Niter damped-Jacobi sweeps on p.v, in place."""
    _require_gpu()
    m = _square_odd(p.v, "p.v")
    _square_odd(p.f, "p.f")
    L = _lib.load()
    _lib.check(L.mgx_relax(p.v.data_ptr(), p.f.data_ptr(), m, p.dx, Niter, omega, _stream_ptr()))


def residual(p):
    """main.cpp:255-281: returns a new m x m array."""
    _require_gpu()
    m = _square_odd(p.v, "p.v")
    r = colmajor(m, m, p.v.device, zero=False)
    L = _lib.load()
    _lib.check(L.mgx_residual(p.v.data_ptr(), p.f.data_ptr(), r.data_ptr(), m, p.dx, _stream_ptr()))
    return r


def coarsen(fine, coarse):
    """main.cpp:201-221: restriction of `fine` into `coarse` (overwritten)."""
    _require_gpu()
    mf = _square_odd(fine, "fine")
    mc, _ = _check_colmajor(coarse, "coarse")
    L = _lib.load()
    _lib.check(L.mgx_restrict(fine.data_ptr(), mf, coarse.data_ptr(), mc, _stream_ptr()))


def refine(coarse, fine):
    """main.cpp:223-253: bilinear interpolation of `coarse` into `fine` (overwritten)."""
    _require_gpu()
    mf = _square_odd(fine, "fine")
    mc, _ = _check_colmajor(coarse, "coarse")
    L = _lib.load()
    _lib.check(L.mgx_refine(coarse.data_ptr(), mc, fine.data_ptr(), mf, _stream_ptr()))


def prolong_add(coarse, v):
    """refine() + `cur += refined` (main.cpp:339-341) without the temporary."""
    _require_gpu()
    mf = _square_odd(v, "v")
    mc, _ = _check_colmajor(coarse, "coarse")
    L = _lib.load()
    _lib.check(L.mgx_prolong_add(coarse.data_ptr(), mc, v.data_ptr(), mf, _stream_ptr()))


def coarse_exact(p):
    """EXTENSION (no counterpart in the reference's MultiGrid): exact solve of the grid p
    (4..65 points per side) for the boundary values p.v holds: the 5-point system of
    getSparseSol(), main.cpp:79-110.  p.v is overwritten in place."""
    _require_gpu()
    m, _ = _check_colmajor(p.v, "p.v")
    L = _lib.load()
    _lib.check(L.mgx_coarse_exact(p.v.data_ptr(), p.f.data_ptr(), m, p.dx, _stream_ptr()))


def direct_solve(p):
    """main.cpp:153-169.  Returns False (and changes nothing) unless the grid is 3x3."""
    _require_gpu()
    m, _ = _check_colmajor(p.v, "p.v")
    L = _lib.load()
    rc = _lib.check(L.mgx_direct3x3(p.v.data_ptr(), p.f.data_ptr(), m, p.dx, _stream_ptr()))
    return rc == 0


def calcNorm(a):
    """main.cpp:147-150: max(Maximum(a), -Minimum(a))."""
    _require_gpu()
    if a.dtype != torch.float64 or not a.is_cuda:
        raise TypeError("calcNorm needs a float64 GPU tensor")
    if a.dim() == 2:
        _check_colmajor(a, "a")
    elif not a.is_contiguous():
        raise ValueError("calcNorm needs dense storage")
    out = ctypes.c_double(0.0)
    L = _lib.load()
    _lib.check(L.mgx_norminf(a.data_ptr(), a.numel(), ctypes.byref(out), _stream_ptr()))
    return out.value


def residual_norm(p):
    """calcNorm(residual(p)) in one pass (main.cpp:307,351)."""
    _require_gpu()
    m = _square_odd(p.v, "p.v")
    out = ctypes.c_double(0.0)
    L = _lib.load()
    _lib.check(L.mgx_residual_norm(p.v.data_ptr(), p.f.data_ptr(), m, p.dx, ctypes.byref(out), _stream_ptr()))
    return out.value


def fused_down(p, Niter, omega):
    """relax(p, Niter, omega); coarsen(residual(p), coarse) in one pass (main.cpp:321-327).

    p.v is updated in place; returns the coarse right-hand side (mc x mc)."""
    _require_gpu()
    m = _square_odd(p.v, "p.v")
    _square_odd(p.f, "p.f")
    mc = (m - 1) // 2 + 1
    coarse = colmajor(mc, mc, p.v.device, zero=False)
    L = _lib.load()
    _lib.check(L.mgx_fused_down(p.v.data_ptr(), p.f.data_ptr(), coarse.data_ptr(), m, p.dx, Niter, omega,
                                _stream_ptr()))
    return coarse


def fused_up(coarse, p, Niter, omega, norm=False):
    """refine + `cur += refined` + relax(p, Niter, omega) in one pass (main.cpp:337-344);
    with norm=True also returns calcNorm(residual(p)) of the result (main.cpp:351)."""
    _require_gpu()
    m = _square_odd(p.v, "p.v")
    _square_odd(p.f, "p.f")
    mc, _ = _check_colmajor(coarse, "coarse")
    if mc != (m - 1) // 2 + 1:
        raise ValueError("coarse size %d does not pair with fine size %d" % (mc, m))
    out = ctypes.c_double(0.0)
    L = _lib.load()
    _lib.check(L.mgx_fused_up(coarse.data_ptr(), p.v.data_ptr(), p.f.data_ptr(), m, p.dx, Niter, omega,
                              ctypes.byref(out) if norm else None, _stream_ptr()))
    return out.value if norm else None


class Plan:
    """Device-resident level hierarchy (mgx_plan_*), reusable across solves."""

    def __init__(self, m, dx, coarsest=2, schedule=_lib.SCHEDULE_DEFAULT, verbose=0,
                 coarse_solve=_lib.COARSE_REFERENCE):
        _require_gpu()
        self._L = _lib.load()
        self._h = ctypes.c_void_p()
        self.m = m
        self.dx = dx
        opt = _lib.make_options(device=torch.cuda.current_device(), stream=torch.cuda.current_stream().cuda_stream,
                                schedule=schedule, verbose=verbose, coarse_solve=coarse_solve)
        _lib.check(self._L.mgx_plan_create(ctypes.byref(self._h), m, dx, coarsest, ctypes.byref(opt)))

    def close(self):
        if self._h:
            self._L.mgx_plan_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _on_current_stream(self):
        """The plan runs on whatever torch stream is current at each call (work it enqueued
        on the stream of an earlier call is ordered first)."""
        _lib.check(self._L.mgx_plan_set_stream(self._h, _stream_ptr()))

    @property
    def depth(self):
        return self._L.mgx_plan_depth(self._h)

    def level_m(self, d):
        return self._L.mgx_plan_level_m(self._h, d)

    @property
    def nbytes(self):
        return self._L.mgx_plan_bytes(self._h)

    def set(self, f, v):
        """Copy column-major GPU tensors f, v into the plan's fine level."""
        m = self.m
        for t, name in ((f, "f"), (v, "v")):
            if _square_odd(t, name) != m:
                raise ValueError("%s does not match the plan size %d" % (name, m))
        self._on_current_stream()
        _lib.check(self._L.mgx_plan_set_dev(self._h, f.data_ptr(), v.data_ptr()))

    def get(self, out=None):
        """The fine-level v as a column-major GPU tensor."""
        m = self.m
        if out is None:
            out = colmajor(m, m, "cuda", zero=False)
        elif _square_odd(out, "out") != m:
            raise ValueError("out does not match the plan size %d" % m)
        self._on_current_stream()
        _lib.check(self._L.mgx_plan_get_dev(self._h, out.data_ptr()))
        return out

    def tune(self, Ndown, Nup, omega):
        """Measure the launch shapes of the level-0 launches of V(Ndown,Nup) cycles now
        (mgx_plan_tune); the data in the plan is left as it was."""
        self._on_current_stream()
        _lib.check(self._L.mgx_plan_tune(self._h, Ndown, Nup, omega))

    def profile(self, enable=True):
        """Start (and reset) or stop per-launch timing of the fine-level kernels; enable=2
        brackets the launches of every level (see profile_levels)."""
        _lib.check(self._L.mgx_plan_profile(self._h, 2 if enable == 2 else int(bool(enable))))

    def profile_levels(self):
        """After profile(2): [(level, rows, launches, total ms)] for the levels that start a
        launch; a launch over a group of small levels is booked on its finest level."""
        out = []
        for d in range(self.depth + 1):
            n = ctypes.c_long(0)
            ms = ctypes.c_double(0.0)
            _lib.check(self._L.mgx_plan_profile_level(self._h, d, ctypes.byref(n), ctypes.byref(ms)))
            if n.value:
                out.append((d, self.level_m(d), n.value, ms.value))
        return out

    def profile_totals(self):
        """{kernel name: (launch count, total milliseconds)} since profile(True)."""
        out = {}
        for k in range(_lib.KERNEL_COUNT):
            n = ctypes.c_long(0)
            ms = ctypes.c_double(0.0)
            _lib.check(self._L.mgx_plan_profile_get(self._h, k, ctypes.byref(n), ctypes.byref(ms)))
            if n.value:
                out[self._L.mgx_kernel_name(k).decode()] = (n.value, ms.value)
        return out

    def cycles(self, Nv, Ndown, Nup, omega, pureJacobi=False):
        res = np.zeros(Nv + 1)
        tim = np.zeros(Nv + 1)
        self._on_current_stream()
        _lib.check(self._L.mgx_plan_cycles(self._h, Nv, Ndown, Nup, omega, int(bool(pureJacobi)),
                                           res.ctypes.data, tim.ctypes.data))
        return MGOutputs(res, tim)

    def cycles_until(self, max_cycles, Ndown, Nup, omega, rel_tol, batch=4):
        """EXTENSION: cycles in batches of `batch` until ||r|| <= rel_tol ||r0||, until the norm
        stops halving (the fp64 floor of the residual) or until max_cycles (mgx_plan_cycles_until).
        Returns (MGOutputs of the cycles done, reason) with reason one of _lib.STOP_*."""
        res = np.zeros(max_cycles + 1)
        tim = np.zeros(max_cycles + 1)
        done, why = ctypes.c_int(0), ctypes.c_int(0)
        self._on_current_stream()
        _lib.check(self._L.mgx_plan_cycles_until(self._h, max_cycles, Ndown, Nup, omega, rel_tol, batch, res.ctypes.data,
                                                 tim.ctypes.data, ctypes.byref(done), ctypes.byref(why)))
        return MGOutputs(res[:done.value + 1], tim[:done.value + 1]), why.value

    def fmg(self, ncyc, Ndown, Nup, omega):
        """EXTENSION (not in the reference): full multigrid -- the right-hand side restricted to
        every level, the coarsest grid solved, then ncyc V(Ndown,Nup) cycles per level on the way
        up, each level starting from the interpolated solution of the one below.  Only the
        boundary values of v are used.  Returns the finest level's MGOutputs (ncyc+1 entries)."""
        res = np.zeros(ncyc + 1)
        tim = np.zeros(ncyc + 1)
        self._on_current_stream()
        _lib.check(self._L.mgx_plan_fmg(self._h, ncyc, Ndown, Nup, omega, res.ctypes.data, tim.ctypes.data))
        return MGOutputs(res, tim)


class MultiPlan:
    """The grid split into j-slabs over several devices, driven from this one process
    (mgx_mplan_*): one host thread and two streams per slab inside the library, halo columns
    pulled with peer copies.  `devices` lists the HIP ordinal of each slab's device and may
    repeat an ordinal (several slabs on one GPU); None puts slab k on device k modulo the
    number of visible devices.  Results are bit-identical to Plan for any number of slabs."""

    def __init__(self, m, dx, coarsest=2, nslabs=2, devices=None, tail_m=0, verbose=0,
                 coarse_solve=_lib.COARSE_REFERENCE):
        _require_gpu()
        self._L = _lib.load()
        self._h = ctypes.c_void_p()
        self.m, self.dx = m, dx
        opt = _lib.make_options(verbose=verbose, coarse_solve=coarse_solve, tail_m=tail_m)
        devs = None
        if devices is not None:
            if len(devices) != nslabs:
                raise ValueError("devices must name one device per slab")
            devs = (ctypes.c_int * nslabs)(*devices)
        _lib.check(self._L.mgx_mplan_create(ctypes.byref(self._h), m, dx, coarsest, nslabs, devs, ctypes.byref(opt)))

    def close(self):
        if self._h:
            self._L.mgx_mplan_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    nslabs = property(lambda self: self._L.mgx_mplan_nslabs(self._h))
    depth = property(lambda self: self._L.mgx_mplan_depth(self._h))
    ndist = property(lambda self: self._L.mgx_mplan_distributed_levels(self._h))
    tail_m = property(lambda self: self._L.mgx_mplan_tail_m(self._h))

    def slab(self, k):
        """mgx_slab_info of slab k (device, stored and owned columns, device pointers)."""
        info = _lib.mgx_slab_info()
        info.struct_size = ctypes.sizeof(_lib.mgx_slab_info)
        _lib.check(self._L.mgx_mplan_slab(self._h, k, ctypes.byref(info)))
        return info

    def upload(self, f, v):
        """f, v: Fortran-ordered float64 m x m host arrays; every slab takes its columns."""
        f = np.asfortranarray(f, dtype=np.float64)
        v = np.asfortranarray(v, dtype=np.float64)
        if f.shape != (self.m, self.m) or v.shape != (self.m, self.m):
            raise ValueError("f and v must be %d x %d" % (self.m, self.m))
        _lib.check(self._L.mgx_mplan_upload(self._h, f.ctypes.data, v.ctypes.data))

    def download(self, out=None):
        if out is None:
            out = np.empty((self.m, self.m), order="F")
        elif not (out.dtype == np.float64 and out.flags.f_contiguous and out.shape == (self.m, self.m)):
            raise TypeError("out must be a Fortran-ordered float64 m x m array")
        _lib.check(self._L.mgx_mplan_download(self._h, out.ctypes.data))
        return out

    def set_random_rhs(self):
        """Benchmark input: the splitmix64 random right-hand side generated on each slab's
        device for the columns it stores, zero guess."""
        from . import rhs
        for k in range(self.nslabs):
            s = self.slab(k)
            dev = torch.device("cuda", s.device)
            with torch.cuda.device(dev):
                f = torch.as_tensor(_DevPtr(s.f_dev, (s.ncols, s.m)), device=dev)
                v = torch.as_tensor(_DevPtr(s.v_dev, (s.ncols, s.m)), device=dev)
                f.copy_(rhs.random_torch_cols(self.m, s.col_first, s.col_first + s.ncols, device=dev))
                v.zero_()
                torch.cuda.synchronize(dev)
        _lib.check(self._L.mgx_mplan_commit(self._h))

    def tune(self, Ndown, Nup, omega):
        _lib.check(self._L.mgx_mplan_tune(self._h, Ndown, Nup, omega))

    def cycles(self, Nv, Ndown, Nup, omega, pureJacobi=False):
        res = np.zeros(Nv + 1)
        tim = np.zeros(Nv + 1)
        _lib.check(self._L.mgx_mplan_cycles(self._h, Nv, Ndown, Nup, omega, int(bool(pureJacobi)),
                                            res.ctypes.data, tim.ctypes.data))
        return MGOutputs(res, tim)

    def fmg(self, ncyc, Ndown, Nup, omega):
        """EXTENSION: full multigrid over the slabs (mgx_mplan_fmg), same bits as Plan.fmg."""
        res = np.zeros(ncyc + 1)
        tim = np.zeros(ncyc + 1)
        _lib.check(self._L.mgx_mplan_fmg(self._h, ncyc, Ndown, Nup, omega, res.ctypes.data, tim.ctypes.data))
        return MGOutputs(res, tim)

    def profile(self, enable=True):
        _lib.check(self._L.mgx_mplan_profile(self._h, int(bool(enable))))

    def profile_totals(self):
        out = {}
        self._pts = {}
        for k in range(_lib.KERNEL_COUNT):
            n, ms, pts = ctypes.c_long(0), ctypes.c_double(0.0), ctypes.c_double(0.0)
            _lib.check(self._L.mgx_mplan_profile_get(self._h, k, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(pts)))
            if n.value:
                name = self._L.mgx_kernel_name(k).decode()
                out[name] = (n.value, ms.value)
                self._pts[name] = pts.value
        return out

    def profile_points(self):
        return dict(getattr(self, "_pts", {}))

    def stats(self):
        """(neighbour exchanges, tail gathers, wall seconds of the last cycles() call, seconds
        slab 0's host thread needed to enqueue that call's work)."""
        x, g, t, q = ctypes.c_long(0), ctypes.c_long(0), ctypes.c_double(0.0), ctypes.c_double(0.0)
        _lib.check(self._L.mgx_mplan_stats(self._h, ctypes.byref(x), ctypes.byref(g), ctypes.byref(t), ctypes.byref(q)))
        return x.value, g.value, t.value, q.value


class _DevPtr:
    """Lets torch wrap device memory owned by libmgx as a tensor without copying."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def FullMultiGrid(prob, ncyc, Ndown, Nup, omega, coarsest, schedule=_lib.SCHEDULE_DEFAULT,
                  coarse_solve=_lib.COARSE_REFERENCE):
    """EXTENSION: Plan.fmg() on a gridtype; prob.v (boundary values in, interior ignored) is
    overwritten with the result."""
    _require_gpu()
    m = _square_odd(prob.f, "prob.f")
    if _square_odd(prob.v, "prob.v") != m:
        raise ValueError("f and v differ in size")
    plan = Plan(m, prob.dx, coarsest, schedule, coarse_solve=coarse_solve)
    try:
        plan.set(prob.f, prob.v)
        out = plan.fmg(ncyc, Ndown, Nup, omega)
        plan.get(out=prob.v)
        torch.cuda.current_stream().synchronize()
    finally:
        plan.close()
    return out


def MultiGrid(prob, Nv, Ndown, Nup, omega, coarsest, pureJacobi=False, schedule=_lib.SCHEDULE_DEFAULT,
              coarse_solve=_lib.COARSE_REFERENCE):
    """main.cpp:283-357: Nv V(Ndown,Nup) cycles; prob.v is overwritten in place.

    Returns MGOutputs(ResidualNorms[Nv+1], Times[Nv+1])."""
    _require_gpu()
    m = _square_odd(prob.f, "prob.f")
    if _square_odd(prob.v, "prob.v") != m:
        raise ValueError("f and v differ in size")
    plan = Plan(m, prob.dx, coarsest, schedule, coarse_solve=coarse_solve)
    try:
        plan.set(prob.f, prob.v)
        out = plan.cycles(Nv, Ndown, Nup, omega, pureJacobi)
        plan.get(out=prob.v)
        torch.cuda.current_stream().synchronize()
    finally:
        plan.close()
    return out


def solve_host(f, v, dx, Nv, Ndown, Nup, omega, coarsest=2, pureJacobi=False, schedule=_lib.SCHEDULE_DEFAULT,
               coarse_solve=_lib.COARSE_REFERENCE, ngpus=0, tail_m=0, zero_guess=False):
    """mgx_solve2d on host numpy arrays (the drop-in for main.cpp:445).

    f, v: Fortran-ordered float64 m x n; v is overwritten.  Returns MGOutputs.  ngpus > 1 splits
    the grid into that many j-slabs (mgx_options.ngpus), tail_m sets the largest level every
    slab holds whole; zero_guess=True states that the interior of v is zero (only its boundary
    is then uploaded)."""
    f = np.asfortranarray(f, dtype=np.float64)
    if not (isinstance(v, np.ndarray) and v.dtype == np.float64 and v.flags.f_contiguous):
        raise TypeError("v must be a Fortran-ordered float64 numpy array (it is written in place)")
    if f.ndim != 2 or v.shape != f.shape:
        raise ValueError("f %s and v %s must be 2-D arrays of one shape" % (f.shape, v.shape))
    m, n = f.shape
    res = np.zeros(Nv + 1)
    tim = np.zeros(Nv + 1)
    L = _lib.load()
    opt = _lib.make_options(schedule=schedule, coarse_solve=coarse_solve, ngpus=ngpus, tail_m=tail_m,
                            flags=_lib.FLAG_ZERO_GUESS if zero_guess else 0)
    _lib.check(L.mgx_solve2d(f.ctypes.data, v.ctypes.data, m, n, dx, Nv, Ndown, Nup, omega, coarsest,
                             int(bool(pureJacobi)), res.ctypes.data, tim.ctypes.data, ctypes.byref(opt)))
    return MGOutputs(res, tim)
