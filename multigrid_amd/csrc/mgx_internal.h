// Internal launcher interface between the C ABI (mgx_api.hip) and the gfx950
// kernels (mgx_kernels.hip, mgx_stream.hip).  Not installed; include/mgx.h is
// the public boundary.
//
// Geometry convention shared by every launcher: a level is an m-row
// column-major array (leading dimension m).  Launchers take a half-open
// column range [jlo, jhi) to update, so that the same kernels serve a whole
// grid (jlo = 1, jhi = m-1) and a j-slab of a partitioned grid.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace mgx {

// Order-preserving encoding of a norm value for unsigned atomic max:
// 0 <-> -infinity ("nothing seen"), otherwise bits(value >= +0.0) + 1.
__host__ __device__ inline unsigned long long norm_encode(double v)
{
    union { double d; unsigned long long u; } x;
    x.d = v;
    return (v == -__builtin_inf()) ? 0ull : x.u + 1ull;
}
__host__ __device__ inline double norm_decode(unsigned long long k)
{
    union { double d; unsigned long long u; } x;
    if (k == 0ull) return -__builtin_inf();
    x.u = k - 1ull;
    return x.d;
}

struct SweepCoef {
    double h2;      // dx*dx                     (main.cpp:180)
    double invh2;   // 1.0/h2                    (main.cpp:268)
    double omega;
    double nomega;  // 1 - omega                 (main.cpp:179)
};

// ---- one kernel per reference operator (mgx_kernels.hip) -------------------
hipError_t launch_jacobi(const double *uin, double *uout, const double *f, int m, int jlo, int jhi,
                         const SweepCoef &c, hipStream_t s);
hipError_t launch_residual(const double *u, const double *f, double *r, int m, int jlo, int jhi,
                           const SweepCoef &c, hipStream_t s);
// Level transfers on slabs: gf0 / gc0 are the GLOBAL column indices of local
// column 0 of the fine / coarse array (both 0 on a whole grid).  Global coarse
// column J pairs with global fine column 2J.
// restrict: writes coarse local columns [Jlo, Jhi), rows 1..mc-2.
hipError_t launch_restrict(const double *fine, int mf, double *coarse, int mc, int Jlo, int Jhi,
                           int gf0, int gc0, hipStream_t s);
// refine: fine local columns [jlo, jhi), rows 1..mf-2; add=1: fine += P(coarse),
// add=0: fine = P(coarse).
hipError_t launch_refine(const double *coarse, int mc, double *fine, int mf, int jlo, int jhi,
                         int gf0, int gc0, int add, hipStream_t s);
hipError_t launch_direct3x3(double *v, const double *f, double h2, hipStream_t s);
hipError_t launch_norminf(const double *a, size_t len, unsigned long long *slot, hipStream_t s);
hipError_t launch_residual_norm(const double *u, const double *f, int m, int jlo, int jhi, const SweepCoef &c,
                                unsigned long long *slot, hipStream_t s);

} // namespace mgx
