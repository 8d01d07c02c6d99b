"""ctypes loader for the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (multigrid_amd/) never does.

All arrays are numpy float64, Fortran (column-major) order, matching
DTDoubleArray (reference DTSource/DTDoubleArray.h:38-49).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_OMP = None
_REF = None

_dp = ctypes.POINTER(ctypes.c_double)


def _ptr(a):
    return a.ctypes.data_as(_dp)


def _f64(a):
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 2 and not a.flags.f_contiguous:
        a = np.asfortranarray(a)
    return a


def lib_omp():
    """The OpenMP build of the same restatement (same bits, all host cores)."""
    global _OMP
    if _OMP is None:
        _OMP = _load("libmg_oracle_omp.so")
    return _OMP


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _load("libmg_oracle.so")
    return _LIB


def _load(name):
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        raise RuntimeError("oracle not built: run `make -C oracle`")
    L = ctypes.CDLL(path)
    L.mgo_threads.restype = ctypes.c_int
    L.mgo_threads.argtypes = []
    L.mgo_maximum.restype = ctypes.c_double
    L.mgo_maximum.argtypes = [_dp, ctypes.c_size_t]
    L.mgo_minimum.restype = ctypes.c_double
    L.mgo_minimum.argtypes = [_dp, ctypes.c_size_t]
    L.mgo_calc_norm.restype = ctypes.c_double
    L.mgo_calc_norm.argtypes = [_dp, ctypes.c_size_t]
    L.mgo_relax.restype = None
    L.mgo_relax.argtypes = [_dp, _dp, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_double]
    L.mgo_residual.restype = None
    L.mgo_residual.argtypes = [_dp, _dp, _dp, ctypes.c_int, ctypes.c_double]
    L.mgo_coarsen.restype = None
    L.mgo_coarsen.argtypes = [_dp, ctypes.c_int, _dp, ctypes.c_int]
    L.mgo_refine.restype = None
    L.mgo_refine.argtypes = [_dp, ctypes.c_int, _dp, ctypes.c_int]
    L.mgo_plus_equals.restype = None
    L.mgo_plus_equals.argtypes = [_dp, _dp, ctypes.c_size_t]
    L.mgo_direct_solve.restype = ctypes.c_int
    L.mgo_direct_solve.argtypes = [_dp, _dp, ctypes.c_int, ctypes.c_double, ctypes.c_int]
    L.mgo_depth.restype = ctypes.c_int
    L.mgo_depth.argtypes = [ctypes.c_int, ctypes.c_int]
    L.mgo_coarse_exact.restype = ctypes.c_int
    L.mgo_coarse_exact.argtypes = [_dp, _dp, ctypes.c_int, ctypes.c_double]
    L.mgo_multigrid_opt.restype = ctypes.c_int
    L.mgo_multigrid_opt.argtypes = [_dp, _dp, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, _dp]
    L.mgo_fmg.restype = ctypes.c_int
    L.mgo_fmg.argtypes = [_dp, _dp, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                          ctypes.c_double, ctypes.c_int, ctypes.c_int, _dp]
    L.mgo_multigrid.restype = ctypes.c_int
    L.mgo_multigrid.argtypes = [_dp, _dp, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int,
                                ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, _dp, _dp]
    return L


def ref_lib():
    """The reference's own DTSource code (oracle/_ref), or None if not built."""
    global _REF
    if _REF is None:
        path = os.path.join(_HERE, "_ref", "libdtref.so")
        if not os.path.exists(path):
            return None
        L = ctypes.CDLL(path)
        L.ref_maximum.restype = ctypes.c_double
        L.ref_maximum.argtypes = [_dp, ctypes.c_long]
        L.ref_minimum.restype = ctypes.c_double
        L.ref_minimum.argtypes = [_dp, ctypes.c_long]
        L.ref_plus_equals.restype = None
        L.ref_plus_equals.argtypes = [_dp, _dp, ctypes.c_long]
        L.ref_offset_of.restype = ctypes.c_long
        L.ref_offset_of.argtypes = [ctypes.c_int] * 4
        _REF = L
    return _REF


_OPS = None


def _bind_ref_ops(path):
    L = ctypes.CDLL(path)
    i, d = ctypes.c_int, ctypes.c_double
    L.ref_calc_norm.restype = d
    L.ref_calc_norm.argtypes = [_dp, i]
    L.ref_relax.restype = None
    L.ref_relax.argtypes = [_dp, _dp, i, d, i, d]
    L.ref_residual.restype = None
    L.ref_residual.argtypes = [_dp, _dp, _dp, i, d]
    L.ref_coarsen.restype = None
    L.ref_coarsen.argtypes = [_dp, i, _dp, i]
    L.ref_refine.restype = None
    L.ref_refine.argtypes = [_dp, i, _dp, i]
    L.ref_direct_solve.restype = None
    L.ref_direct_solve.argtypes = [_dp, _dp, i, d]
    L.ref_depth.restype = i
    L.ref_depth.argtypes = [i, i]
    L.ref_multigrid.restype = None
    L.ref_multigrid.argtypes = [_dp, _dp, i, d, i, i, i, d, i, i, _dp, _dp]
    return L


def ref_ops():
    """The reference's own solver functions (main.cpp:147-357) compiled from the reference
    checkout into oracle/_ref/librefops.so (oracle/Makefile, oracle/ref_main_harness.cpp), or
    None if not built.  Used only to pin this oracle."""
    global _OPS
    if _OPS is None:
        path = os.path.join(_HERE, "_ref", "librefops.so")
        if not os.path.exists(path):
            return None
        _OPS = _bind_ref_ops(path)
    return _OPS


_OPS_REL = None


def ref_ops_release():
    """TIMING ONLY: the same functions built with the reference's Release flags (-O3 -Ofast
    -DNDEBUG, CMakeLists.txt:17) into oracle/_ref/librefops_release.so, or None.  Fast-math bits
    are not the parity definition; nothing compares against this build."""
    global _OPS_REL
    if _OPS_REL is None:
        path = os.path.join(_HERE, "_ref", "librefops_release.so")
        if not os.path.exists(path):
            return None
        _OPS_REL = _bind_ref_ops(path)
    return _OPS_REL


class Ref:
    """numpy front end of ref_ops() with the same call shapes as this module's functions, so a
    test can run `O.relax(...)` and `O.Ref().relax(...)` side by side."""

    def __init__(self, release=False):
        """release=True: the -Ofast build, for timing only."""
        self.L = ref_ops_release() if release else ref_ops()
        if self.L is None:
            raise RuntimeError("oracle/_ref/librefops%s.so not built (reference checkout absent)"
                               % ("_release" if release else ""))

    def calc_norm(self, r):
        r = _f64(r)
        assert r.ndim == 2 and r.shape[0] == r.shape[1]
        return self.L.ref_calc_norm(_ptr(r), r.shape[0])

    def relax(self, u, f, dx, niter, omega):
        u = _f64(u).copy(order="F")
        f = _f64(f)
        self.L.ref_relax(_ptr(u), _ptr(f), u.shape[0], dx, niter, omega)
        return u

    def residual(self, u, f, dx):
        u = _f64(u)
        f = _f64(f)
        r = np.full(u.shape, np.nan, order="F")
        self.L.ref_residual(_ptr(u), _ptr(f), _ptr(r), u.shape[0], dx)
        return r

    def coarsen(self, fine, mc=None):
        fine = _f64(fine)
        mf = fine.shape[0]
        if mc is None:
            mc = (mf - 1) // 2 + 1
        c = np.full((mc, mc), np.nan, order="F")
        self.L.ref_coarsen(_ptr(fine), mf, _ptr(c), mc)
        return c

    def refine(self, coarse, mf=None):
        coarse = _f64(coarse)
        mc = coarse.shape[0]
        if mf is None:
            mf = 2 * (mc - 1) + 1
        fine = np.full((mf, mf), np.nan, order="F")
        self.L.ref_refine(_ptr(coarse), mc, _ptr(fine), mf)
        return fine

    def direct_solve(self, u, f, dx):
        u = _f64(u).copy(order="F")
        f = _f64(f)
        self.L.ref_direct_solve(_ptr(u), _ptr(f), u.shape[0], dx)
        return u

    def depth(self, m, coarsest):
        return self.L.ref_depth(m, coarsest)

    def multigrid(self, f, v, dx, Nv, Ndown, Nup, omega, coarsest=2, pure_jacobi=False):
        f = _f64(f)
        v = _f64(v).copy(order="F")
        res = np.zeros(Nv + 1)
        tim = np.zeros(Nv + 1)
        self.L.ref_multigrid(_ptr(f), _ptr(v), f.shape[0], dx, Nv, Ndown, Nup, omega, coarsest,
                             int(bool(pure_jacobi)), _ptr(res), _ptr(tim))
        return v, res, tim


def calc_norm(r):
    r = np.ascontiguousarray(np.asarray(r, dtype=np.float64).ravel(order="K"))
    return lib().mgo_calc_norm(_ptr(r), r.size)


def relax(u, f, dx, niter, omega):
    """In-place on a copy; returns the new u (m x m, Fortran order)."""
    u = _f64(u).copy(order="F")
    f = _f64(f)
    lib().mgo_relax(_ptr(u), _ptr(f), u.shape[0], dx, niter, omega)
    return u


def residual(u, f, dx):
    u = _f64(u)
    f = _f64(f)
    r = np.empty_like(u, order="F")
    lib().mgo_residual(_ptr(u), _ptr(f), _ptr(r), u.shape[0], dx)
    return r


def coarsen(fine, mc=None):
    fine = _f64(fine)
    mf = fine.shape[0]
    if mc is None:
        mc = (mf - 1) // 2 + 1
    c = np.empty((mc, mc), dtype=np.float64, order="F")
    lib().mgo_coarsen(_ptr(fine), mf, _ptr(c), mc)
    return c


def refine(coarse, mf=None):
    coarse = _f64(coarse)
    mc = coarse.shape[0]
    if mf is None:
        mf = 2 * (mc - 1) + 1
    fine = np.empty((mf, mf), dtype=np.float64, order="F")
    lib().mgo_refine(_ptr(coarse), mc, _ptr(fine), mf)
    return fine


def direct_solve(u, f, dx):
    u = _f64(u).copy(order="F")
    f = _f64(f)
    rc = lib().mgo_direct_solve(_ptr(u), _ptr(f), u.shape[0], dx, 1)
    return u, rc


def depth(m, coarsest):
    return lib().mgo_depth(m, coarsest)


_EIG = None


def ref_eigen():
    """Eigen::SimplicialCholesky on the 5-point system (oracle/_ref), or None if not built."""
    global _EIG
    if _EIG is None:
        path = os.path.join(_HERE, "_ref", "libeigenref.so")
        if not os.path.exists(path):
            return None
        L = ctypes.CDLL(path)
        L.ref_poisson_exact.restype = ctypes.c_int
        L.ref_poisson_exact.argtypes = [_dp, _dp, ctypes.c_int, ctypes.c_double]
        _EIG = L
    return _EIG


def coarse_exact(u, f, dx, eigen=False):
    """EXTENSION: exact solve of one grid (boundary of u kept); eigen=True uses the reference's Eigen."""
    u = _f64(u).copy(order="F")
    f = _f64(f)
    fn = ref_eigen().ref_poisson_exact if eigen else lib().mgo_coarse_exact
    rc = fn(_ptr(f), _ptr(u), u.shape[0], dx) if eigen else fn(_ptr(u), _ptr(f), u.shape[0], dx)
    if rc != 0:
        raise ValueError("exact solve failed (rc=%d)" % rc)
    return u


def multigrid(f, v, dx, Nv, Ndown, Nup, omega, coarsest=2, pure_jacobi=False, omp=False, coarse_exact=False):
    """Returns (Sol, ResNorms, Times) like main.cpp:445-455.  omp=True runs the threaded build;
    coarse_exact=True is the extension that solves a coarsest grid other than 3x3 exactly."""
    f = _f64(f)
    v = _f64(v).copy(order="F")
    res = np.zeros(Nv + 1)
    tim = np.zeros(Nv + 1)
    rc = (lib_omp() if omp else lib()).mgo_multigrid_opt(_ptr(f), _ptr(v), f.shape[0], dx, Nv, Ndown, Nup, omega,
                                                         coarsest, int(bool(pure_jacobi)), int(bool(coarse_exact)),
                                                         _ptr(res), _ptr(tim))
    if rc != 0:
        raise ValueError("oracle refused the hierarchy (rc=%d)" % rc)
    return v, res, tim


def fmg(f, v, dx, ncyc, Ndown, Nup, omega, coarsest=2, coarse_exact=False):
    """EXTENSION: full multigrid composed of the oracle's operators.  Returns (Sol, ResNorms)
    with ResNorms the history of the finest level's ncyc cycles."""
    f = _f64(f)
    v = _f64(v).copy(order="F")
    res = np.zeros(ncyc + 1)
    rc = lib().mgo_fmg(_ptr(f), _ptr(v), f.shape[0], dx, ncyc, Ndown, Nup, omega, coarsest, int(bool(coarse_exact)),
                       _ptr(res))
    if rc != 0:
        raise ValueError("oracle refused the hierarchy (rc=%d)" % rc)
    return v, res
