// mgx_coarse.hip -- stand-alone launch of the exact coarsest-grid solve (see
// mgx_coarse.h) for the schedules that do not run the LDS-resident tail, and
// the host-side tables.
#include "mgx_coarse.h"

#include <math.h>

#include <atomic>

namespace mgx {

namespace {

__global__ __launch_bounds__(1024) void k_coarse_exact(double *u, const double *f, int m, double h2,
                                                       const double *tables)
{
    extern __shared__ __align__(16) double lds[];
    const int n = m - 2;
    double *S = lds;            // n*n
    double *lam = S + n * n;    // n
    double *B = lam + n;        // n*n
    double *T = B + n * n;      // n*n
    for (int k = threadIdx.x; k < n * n + n; k += blockDim.x) lds[k] = tables[k];
    __syncthreads();
    coarse_exact_solve(u, m, f, m, m, h2, S, lam, B, T);
}

} // namespace

void coarse_tables(int m, double *t)
{
    const int n = m - 2;
    const double pi = 3.14159265358979323846;
    for (int k = 0; k < n; k++)
        for (int i = 0; i < n; i++) t[i + k * n] = sin(pi * (double)(i + 1) * (double)(k + 1) / (double)(n + 1));
    for (int i = 0; i < n; i++) t[n * n + i] = 2.0 - 2.0 * cos(pi * (double)(i + 1) / (double)(n + 1));
}

hipError_t launch_coarse_exact(double *u, const double *f, int m, double h2, const double *tables, hipStream_t s)
{
    if (m < 4 || m > COARSE_EXACT_MAX_M) return hipErrorInvalidValue;
    const int n = m - 2;
    const size_t bytes = sizeof(double) * ((size_t)3 * n * n + n);
    // a launch with more than 64 KB of dynamic LDS must be allowed once per device
    static std::atomic<unsigned long long> configured{0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = (dev >= 0 && dev < 64) ? (1ull << dev) : 0ull;
    if (!(configured.load(std::memory_order_acquire) & bit)) {
        e = hipFuncSetAttribute((const void *)k_coarse_exact, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * (3 * 63 * 63 + 63)));
        if (e != hipSuccess) return e;
        configured.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL(k_coarse_exact, dim3(1), dim3(1024), bytes, s, u, f, m, h2, tables);
    return hipGetLastError();
}

} // namespace mgx
