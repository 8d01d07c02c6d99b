// ref_main_harness.cpp -- TEST INFRASTRUCTURE ONLY.
//
// extern "C" wrappers around the reference's own solver functions
//   calcNorm main.cpp:147-150      direct_solve main.cpp:153-169   relax    main.cpp:171-199
//   coarsen  main.cpp:201-221      refine       main.cpp:223-253   residual main.cpp:255-281
//   MultiGrid main.cpp:283-357
// This file is NOT a translation unit of its own.  oracle/Makefile streams lines 1-20 and
// 24-358 of the reference's main.cpp (everything but the boost include/alias at 21-23 and
// main() at 360-459, the only parts that need boost) into the compiler's standard input and
// appends this file, so the wrappers below see gridtype, MGOutputs and the functions exactly
// as the reference defines them.  No reference text is written to disk and no stand-in
// header is involved; the DTSource sources are compiled where they lie.  Flags:
// -O2 -DNDEBUG -ffp-contract=off (the reference's Release build defines NDEBUG too,
// CMakeLists.txt:17; without it coarsen() asserts on an even-sized coarse level,
// main.cpp:206).  Output: oracle/_ref/librefops.so, used by tests/test_oracle.py and
// tests/golden/make_golden.py to pin oracle/mg_oracle.c.
//
// Arrays are column-major m x m doubles owned by the caller; the wrappers copy in and out.
#include <cstring>

namespace refh {

static DTMutableDoubleArray from_raw(const double *p, int m)
{
    DTMutableDoubleArray a(m, m);
    std::memcpy(a.Pointer(), p, sizeof(double) * (size_t)m * (size_t)m);
    return a;
}

static void to_raw(const DTDoubleArray &a, double *p)
{
    std::memcpy(p, a.Pointer(), sizeof(double) * (size_t)a.Length());
}

static gridtype make_grid(const double *f, const double *v, int m, double dx)
{
    DTMesh2DGrid g(DTPoint2D(0, 0), dx, dx, m, m);
    gridtype p;
    p.f = DTMutableMesh2D(g, from_raw(f, m));
    p.v = DTMutableMesh2D(g, from_raw(v, m));
    return p;
}

}  // namespace refh

extern "C" {

double ref_calc_norm(const double *a, int m)
{
    return calcNorm(refh::from_raw(a, m));
}

void ref_relax(double *v, const double *f, int m, double dx, int niter, double omega)
{
    gridtype p = refh::make_grid(f, v, m, dx);
    relax(p, niter, omega);
    refh::to_raw(p.v.DoubleData(), v);
}

void ref_residual(const double *v, const double *f, double *r, int m, double dx)
{
    gridtype p = refh::make_grid(f, v, m, dx);
    refh::to_raw(residual(p), r);
}

// coarse (mc x mc) is pre-filled by the caller so that the test sees that coarsen() overwrites
// every entry, boundary included
void ref_coarsen(const double *fine, int mf, double *coarse, int mc)
{
    DTMutableDoubleArray c = refh::from_raw(coarse, mc);
    coarsen(refh::from_raw(fine, mf), c);
    refh::to_raw(c, coarse);
}

void ref_refine(const double *coarse, int mc, double *fine, int mf)
{
    DTMutableDoubleArray fi = refh::from_raw(fine, mf);
    refine(refh::from_raw(coarse, mc), fi);
    refh::to_raw(fi, fine);
}

void ref_direct_solve(double *v, const double *f, int m, double dx)
{
    gridtype p = refh::make_grid(f, v, m, dx);
    direct_solve(p);
    refh::to_raw(p.v.DoubleData(), v);
}

// depth exactly as MultiGrid() computes it (main.cpp:286), evaluated by the same compiler on
// the same expression: the float arithmetic is part of the contract
int ref_depth(int m, int coarsest)
{
    return int(log2(1.0f * (m - 1) / coarsest) + 0.5f);
}

void ref_multigrid(const double *f, double *v, int m, double dx, int Nv, int Ndown, int Nup, double omega,
                   int coarsest, int pure_jacobi, double *resnorms, double *times)
{
    gridtype p = refh::make_grid(f, v, m, dx);
    MGOutputs out = MultiGrid(p, Nv, Ndown, Nup, omega, coarsest, pure_jacobi != 0);
    refh::to_raw(p.v.DoubleData(), v);
    refh::to_raw(out.ResidualNorms, resnorms);
    refh::to_raw(out.Times, times);
}

}  // extern "C"
