// mgx_coarse.h -- exact solve of the coarsest grid by a sine-transform
// (fast-Poisson) solver, shared by the stand-alone kernel (mgx_coarse.hip) and
// the LDS-resident tail (mgx_tail.hip).
//
// EXTENSION, off by default.  The reference solves only a 3x3 coarsest grid
// (direct_solve, main.cpp:153-169) and leaves any other untouched.  With
// mgx_options.coarse_solve = MGX_COARSE_EXACT the coarsest grid satisfies
//     4 u(i,j) - u(i-1,j) - u(i+1,j) - u(i,j-1) - u(i,j+1) = -h^2 f(i,j)
// for given boundary values of u: the system of the reference's unused
// getSparseSol() (main.cpp:79-110).  With n = m-2 unknowns per side the matrix
// is T (x) I + I (x) T, T = tridiag(-1, 2, -1), whose eigenvectors are sines:
//     U = c^2 S ((S B S) ./ (lam_i + lam_j)) S,    S(i,k) = sin(pi (i+1)(k+1) / (n+1)),
//     lam_i = 2 - 2 cos(pi (i+1) / (n+1)),  c = 2 / (n+1),
// four n x n matrix products -- no sequential factorisation, so one workgroup
// solves a 63 x 63 grid in a few microseconds.  S and lam are tabulated on the
// host when the plan is created.  Accuracy is that of any backward-stable
// solver for this matrix (condition number ~ 0.4 (n+1)^2); it is compared with
// the oracle's Cholesky solve to a tolerance, not bit for bit.
#pragma once
#include <hip/hip_runtime.h>

namespace mgx {

static constexpr int COARSE_EXACT_MAX_M = 65;  // largest coarsest grid the exact solve takes

// C = A * B for n x n column-major matrices in LDS (or any memory); the workgroup's
// waves take columns, its lanes rows (n <= 63 < 64).  With W != nullptr the result
// is scaled: C(i,j) *= scale / (W[i] + W[j]).
__device__ __forceinline__ void coarse_matmul(double *__restrict__ C, const double *__restrict__ A,
                                              const double *__restrict__ B, int n, const double *__restrict__ W,
                                              double scale)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    if (lane < n) {
        for (int j = wave; j < n; j += nwaves) {
            double acc0 = 0.0, acc1 = 0.0;  // two chains: the sum order is fixed, the latency halves
            int k = 0;
            for (; k + 1 < n; k += 2) {
                acc0 = acc0 + A[lane + k * n] * B[k + j * n];
                acc1 = acc1 + A[lane + (k + 1) * n] * B[(k + 1) + j * n];
            }
            if (k < n) acc0 = acc0 + A[lane + k * n] * B[k + j * n];
            double r = acc0 + acc1;
            if (W) r = r * scale / (W[lane] + W[j]);
            C[lane + j * n] = r;
        }
    }
    __syncthreads();
}

// Solves the m x m grid u (pitch ldu; boundary values given, interior overwritten)
// for the right-hand side f (pitch ldf).  S (n*n), lam (n), B and T (n*n each) are
// workgroup-visible arrays; B and T are scratch.  Every thread of the workgroup calls.
__device__ __forceinline__ void coarse_exact_solve(double *u, int ldu, const double *f, int ldf, int m, double h2,
                                                   const double *S, const double *lam, double *B, double *T)
{
    const int n = m - 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    if (lane < n) {
        for (int j = wave; j < n; j += nwaves) {
            const int i = lane;
            double b = -h2 * f[(i + 1) + (j + 1) * ldf];
            if (i == 0) b += u[0 + (j + 1) * ldu];
            if (i == n - 1) b += u[(m - 1) + (j + 1) * ldu];
            if (j == 0) b += u[(i + 1) + 0 * ldu];
            if (j == n - 1) b += u[(i + 1) + (m - 1) * ldu];
            B[i + j * n] = b;
        }
    }
    __syncthreads();
    const double c = 2.0 / (double)(n + 1);
    coarse_matmul(T, S, B, n, nullptr, 0.0);  // T = S B
    coarse_matmul(B, T, S, n, lam, c * c);    // B = c^2 (T S) ./ (lam_i + lam_j)
    coarse_matmul(T, S, B, n, nullptr, 0.0);  // T = S B
    coarse_matmul(B, T, S, n, nullptr, 0.0);  // B = T S = U
    if (lane < n)
        for (int j = wave; j < n; j += nwaves) u[(lane + 1) + (j + 1) * ldu] = B[lane + j * n];
    __syncthreads();
}

// Host: the tables for an m x m coarsest grid, n = m-2: S (n*n doubles) then lam (n doubles).
void coarse_tables(int m, double *S_then_lam);
// Stand-alone launch (one workgroup); tables = device pointer to S then lam.
hipError_t launch_coarse_exact(double *u, const double *f, int m, double h2, const double *tables, hipStream_t s);

} // namespace mgx
