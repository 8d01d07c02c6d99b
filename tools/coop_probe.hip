// Probe: cost of a grid-wide barrier in a cooperative launch shaped like a persistent
// multigrid "mid-level" kernel (one 1024-thread workgroup per CU, 137 KB of dynamic LDS).
#include <hip/hip_cooperative_groups.h>
#include <hip/hip_runtime.h>
#include <cstdio>
namespace cg = cooperative_groups;

// hand-rolled barrier: one arrival per workgroup on a monotonically increasing counter
__device__ __forceinline__ void grid_barrier(unsigned *counter, unsigned target)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target)
            __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void k_probe2(int nsync, double *buf, int n, unsigned *counter)
{
    extern __shared__ double lds[];
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (threadIdx.x == 0) lds[0] = 1.0;
    for (int s = 0; s < nsync; s++) {
        const int k = (gid + s * 1024) % n;
        buf[k] = buf[k] + 1.0;
        grid_barrier(counter, (unsigned)(s + 1) * gridDim.x);
    }
}

__global__ __launch_bounds__(1024) void k_probe(int nsync, double *buf, int n)
{
    extern __shared__ double lds[];
    cg::grid_group grid = cg::this_grid();
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (threadIdx.x == 0) lds[0] = 1.0;
    for (int s = 0; s < nsync; s++) {
        // every phase: each thread bumps one element that a thread of ANOTHER workgroup bumped last phase
        const int k = (gid + s * 1024) % n;
        buf[k] = buf[k] + 1.0;
        grid.sync();
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main()
{
    int dev = 0, ncu = 0, coop = 0;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    CK(hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, dev));
    const size_t lds = 137 * 1024;
    CK(hipFuncSetAttribute((const void *)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_probe, 1024, lds));
    printf("CUs %d cooperative %d blocks/CU %d\n", ncu, coop, per_cu);
    if (!coop || per_cu < 1) return 0;
    const int grid = ncu * per_cu;
    const int n = grid * 1024;
    double *buf;
    CK(hipMalloc(&buf, sizeof(double) * n));
    CK(hipMemset(buf, 0, sizeof(double) * n));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int nsync : {0, 1, 10, 100, 100}) {
        int ns = nsync;
        int nn = n;
        void *args[] = {&ns, &buf, &nn};
        CK(hipEventRecord(a, 0));
        CK(hipLaunchCooperativeKernel((const void *)k_probe, dim3(grid), dim3(1024), args, lds, 0));
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        printf("grid %d nsync %3d: %.1f us total, %.2f us per sync\n", grid, nsync, ms * 1e3, nsync ? ms * 1e3 / nsync : 0.0);
    }
    CK(hipFuncSetAttribute((const void *)k_probe2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    unsigned *counter;
    CK(hipMalloc(&counter, 256));
    CK(hipMemset(buf, 0, sizeof(double) * n));
    int total = 0;
    for (int nsync : {0, 1, 10, 100, 100}) {
        int ns = nsync;
        int nn = n;
        total += nsync;
        void *args[] = {&ns, &buf, &nn, &counter};
        CK(hipMemsetAsync(counter, 0, 4, 0));
        CK(hipEventRecord(a, 0));
        CK(hipLaunchCooperativeKernel((const void *)k_probe2, dim3(grid), dim3(1024), args, lds, 0));
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        printf("own barrier: grid %d nsync %3d: %.1f us total, %.2f us per sync\n", grid, nsync, ms * 1e3, nsync ? ms * 1e3 / nsync : 0.0);
    }
    double h[4];
    CK(hipMemcpy(h, buf, sizeof(h), hipMemcpyDeviceToHost));
    printf("buf[0..3] = %g %g %g %g (expect 211 each, twice)\n", h[0], h[1], h[2], h[3]);
    return 0;
}
