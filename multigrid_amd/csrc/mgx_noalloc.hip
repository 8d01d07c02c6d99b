// mgx_noalloc.hip -- mgx_residual_norm_noalloc(): calcNorm(residual(p)) (reference
// main.cpp:147-150 over :255-281, as called at main.cpp:307,351) for callers that cannot
// allocate: code under HIP graph capture, tight serving loops, harnesses that forbid
// hipMalloc.  Same arithmetic as mgx_residual_norm(); the difference is where the reduction
// keeps its partial results.
//
// mgx_residual_norm() allocates a norm slot, merges workgroup maxima into it with atomics
// and frees it.  Here the partial maxima go to a file-scope device array, `scratch`: one
// double per workgroup, written by k_resnorm_partials and folded by one workgroup of
// k_resnorm_fold, which stores the norm where the caller wants it (device memory).  No
// atomics, nothing that has to start at zero, no allocation, no host synchronisation.
//
// `scratch` is 64 KiB per GPU, held until the process ends.  Every call on a GPU uses that
// GPU's one array: calls that overlap on one GPU (two streams, two threads) race on it.
#include "../../include/mgx.h"
#include "mgx_internal.h"

#include <mutex>

namespace {

constexpr int SCRATCH_DOUBLES = 8192;  // one partial maximum per workgroup of a call
constexpr int NA_BX = 64, NA_BY = 4;   // lanes along i, columns per workgroup pass
constexpr int NA_MAX_M = 65537;        // grid rows/columns a call accepts
constexpr int NA_MAX_DEVICES = 64;

// In host code this name is a stand-in, not a device address: everything below passes the
// address hipGetSymbolAddress() returned for the current device, never `scratch` itself.
static __device__ __align__(256) double scratch[SCRATCH_DOUBLES];

std::mutex g_mu;
double *g_scratch_ptr[NA_MAX_DEVICES] = {nullptr};

// The address of `scratch` on the current device (each GPU holds its own copy of the array
// at its own address), looked up once per device with that device current.
hipError_t scratch_on_current_device(double **out)
{
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= NA_MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g_scratch_ptr[dev]) {
        double *p = nullptr;
        e = hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return hipSuccess;
}

} // namespace

// Look the array up for the current device now (mgx_plan_create and mgx_mplan_create call this),
// so that a later mgx_residual_norm_noalloc() on this device, possibly under stream capture,
// finds the address cached and makes no runtime call but its two launches.
void mgx::noalloc_prepare_device()
{
    double *p = nullptr;
    if (scratch_on_current_device(&p) != hipSuccess) (void)hipGetLastError();
}

namespace {

// Workgroup (bx, by) walks columns 1 + by*NA_BY + ty, stride gridDim.y*NA_BY, of rows
// 1 + bx*64 + tx, and writes the maximum |residual| it saw to partials[bx + by*gridDim.x].
// Boundary entries of the residual are +0.0 (main.cpp:267), so accumulators start at +0.0.
__global__ __launch_bounds__(NA_BX *NA_BY) void k_resnorm_partials(const double *__restrict__ u,
                                                                  const double *__restrict__ f, int m,
                                                                  double invh2, double *__restrict__ partials)
{
    __shared__ double wmax[NA_BY];
    const int i = 1 + blockIdx.x * NA_BX + threadIdx.x;
    double acc = 0.0;
    if (i < m - 1) {
        for (int j = 1 + blockIdx.y * NA_BY + threadIdx.y; j < m - 1; j += gridDim.y * NA_BY) {
            const size_t k = (size_t)i + (size_t)j * (size_t)m;
            acc = mgx::absmax_step(acc, mgx::residual_at(u, f, k, m, invh2));
        }
    }
    acc = mgx::wave_max(acc);  // one wavefront per threadIdx.y
    if (threadIdx.x == 0) wmax[threadIdx.y] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        double v = wmax[0];
#pragma unroll
        for (int w = 1; w < NA_BY; w++) v = (v < wmax[w]) ? wmax[w] : v;
        partials[blockIdx.x + blockIdx.y * gridDim.x] = v;
    }
}

// One workgroup folds the n partial maxima of THIS call (never more: older calls may have
// left larger values further up the array) and stores the norm.
__global__ __launch_bounds__(256) void k_resnorm_fold(const double *__restrict__ partials, int n,
                                                      double *__restrict__ out)
{
    __shared__ double wmax[4];
    double acc = 0.0;
    for (int k = threadIdx.x; k < n; k += 256) {
        const double v = partials[k];
        acc = (acc < v) ? v : acc;
    }
    acc = mgx::wave_max(acc);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double v = wmax[0];
#pragma unroll
        for (int w = 1; w < 4; w++) v = (v < wmax[w]) ? wmax[w] : v;
        *out = v;
    }
}

}  // namespace

extern "C" int mgx_residual_norm_noalloc(const double *v_dev, const double *f_dev, int m, double dx, double *norm_dev,
                                         void *stream)
{
    using mgx::api_fail;
    if (!v_dev || !f_dev || !norm_dev) return api_fail(MGX_ERR_ARG, "mgx_residual_norm_noalloc: null pointer");
    if (m < 3) return api_fail(MGX_ERR_ARG, "mgx_residual_norm_noalloc: grid size %d is below 3", m);
    if ((m % 2) == 0)
        return api_fail(MGX_ERR_ARG, "mgx_residual_norm_noalloc: Input has even rows and columns! (m=%d)", m);
    if (m > NA_MAX_M)
        return api_fail(MGX_ERR_ARG, "mgx_residual_norm_noalloc: m=%d exceeds the %d this entry point is built for "
                                     "(its fixed scratch array holds %d partial results); use mgx_residual_norm()",
                        m, NA_MAX_M, SCRATCH_DOUBLES);
    if (!(dx > 0.0)) return api_fail(MGX_ERR_ARG, "mgx_residual_norm_noalloc: dx must be positive");
    // grid: one workgroup column per 64 rows; as many workgroup rows as fit the scratch array,
    // at most one per NA_BY columns and about 4096 workgroups in all
    const unsigned gx = (unsigned)((m - 2 + NA_BX - 1) / NA_BX);
    unsigned gy = (unsigned)((m - 2 + NA_BY - 1) / NA_BY);
    const unsigned maxy = (4096 + gx - 1) / gx;
    if (gy > maxy) gy = maxy;
    if ((size_t)gx * gy > (size_t)SCRATCH_DOUBLES)
        return api_fail(MGX_ERR_ARG, "mgx_residual_norm_noalloc: %u x %u workgroups exceed the scratch array (%d)", gx,
                        gy, SCRATCH_DOUBLES);
    double *ptr = nullptr;
    hipError_t e = scratch_on_current_device(&ptr);
    if (e != hipSuccess)
        return api_fail(MGX_ERR_HIP, "mgx_residual_norm_noalloc: scratch address: %s", hipGetErrorString(e));
    hipStream_t s = (hipStream_t)stream;
    const double h2 = dx * dx;      // main.cpp:263
    const double invh2 = 1.0 / h2;  // main.cpp:268
    hipLaunchKernelGGL(k_resnorm_partials, dim3(gx, gy), dim3(NA_BX, NA_BY), 0, s, v_dev, f_dev, m, invh2, ptr);
    e = hipGetLastError();
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_resnorm_fold, dim3(1), dim3(256), 0, s, (const double *)ptr, (int)(gx * gy), norm_dev);
        e = hipGetLastError();
    }
    if (e != hipSuccess) return api_fail(MGX_ERR_HIP, "mgx_residual_norm_noalloc: launch: %s", hipGetErrorString(e));
    return MGX_OK;
}
