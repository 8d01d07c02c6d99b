// hbm_ceiling.hip -- measured HBM streaming ceilings of the GPU this runs on:
// read-only, write-only, copy (1R+1W) and the 2R+1W mix of a Jacobi sweep, as
// plain 16-byte-per-lane grid-stride kernels over arrays far larger than the
// 256 MiB Infinity Cache.  The V-cycle kernels are compared against these
// numbers as well as against the 8 TB/s datasheet peak.
//
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_ceiling.hip -o tools/hbm_ceiling && tools/hbm_ceiling [MiB] [LDS bytes]
// The optional second argument reserves that much (unused) LDS per workgroup, which caps the
// number of resident workgroups per CU: fewer concurrent streams can use the DRAM better.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

__global__ __launch_bounds__(256) void k_read(const double2 *__restrict__ a, size_t n, double *sink)
{
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double2 v = a[i];
        acc += v.x + v.y;
    }
    if (acc == 123456.789) *sink = acc;  // keeps the loads alive, never true in practice
}

__global__ __launch_bounds__(256) void k_write(double2 *__restrict__ a, size_t n, double val)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        a[i] = make_double2(val, val);
}

__global__ __launch_bounds__(256) void k_copy(const double2 *__restrict__ a, double2 *__restrict__ b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}

__global__ __launch_bounds__(256) void k_2r1w(const double2 *__restrict__ a, const double2 *__restrict__ f,
                                              double2 *__restrict__ b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double2 x = a[i], y = f[i];
        b[i] = make_double2(x.x * 0.4 + y.x, x.y * 0.4 + y.y);
    }
}

int main(int argc, char **argv)
{
    const size_t mib = argc > 1 ? (size_t)atol(argv[1]) : 2048;
    const size_t lds = argc > 2 ? (size_t)atol(argv[2]) : 0;
    const size_t bytes = mib << 20;
    const size_t n = bytes / sizeof(double2);
    double2 *a, *b, *c;
    double *sink;
    CHECK(hipMalloc(&a, bytes));
    CHECK(hipMalloc(&b, bytes));
    CHECK(hipMalloc(&c, bytes));
    CHECK(hipMalloc(&sink, 8));
    CHECK(hipMemset(a, 0, bytes));
    CHECK(hipMemset(b, 0, bytes));
    CHECK(hipMemset(c, 0, bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int reps = 10;
    const int grids[] = {512, 1024, 1536, 2048, 4096, 8192, 16384, 65536};
    printf("array %zu MiB, %d reps, %zu bytes of LDS reserved per workgroup, best grid per kernel\n", mib, reps, lds);
    for (int which = 0; which < 4; which++) {
        double best = 0;
        int bestg = 0;
        for (int g : grids) {
            for (int r = -2; r < reps; r++) {
                if (r == 0) CHECK(hipEventRecord(e0, 0));
                switch (which) {
                case 0: hipLaunchKernelGGL(k_read, dim3(g), dim3(256), lds, 0, a, n, sink); break;
                case 1: hipLaunchKernelGGL(k_write, dim3(g), dim3(256), lds, 0, b, n, 1.0); break;
                case 2: hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), lds, 0, a, b, n); break;
                default: hipLaunchKernelGGL(k_2r1w, dim3(g), dim3(256), lds, 0, a, c, b, n); break;
                }
            }
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double moved = (double)bytes * (which == 0 || which == 1 ? 1 : which == 2 ? 2 : 3) * reps;
            const double tbs = moved / (ms * 1e-3) / 1e12;
            if (tbs > best) { best = tbs; bestg = g; }
        }
        const char *names[] = {"read-only  (1R)   ", "write-only (1W)   ", "copy       (1R+1W)", "sweep mix  (2R+1W)"};
        printf("%s %.3f TB/s (grid %d)\n", names[which], best, bestg);
    }
    return 0;
}
