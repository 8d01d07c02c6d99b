// Is a kernel made of many barrier-separated phases slower when every phase has its own
// code (compile-time unrolled, as k_tail and the grouped kernels are) than when the phases
// run the same code in a loop?  Same arithmetic, same LDS traffic, same number of barriers.
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/icache_probe.hip -o <binary>
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void phase(double *__restrict__ a, double *__restrict__ b, int x, double c, double &u)
{
    double s = a[x - 1] + a[x + 1];
    s = s + a[x - 66];
    s = s + a[x + 66];
    s = s - c * 0.37;
    s = s * 0.25;
    u = u * 0.4 + s * 0.6;
    b[x] = u;
    __syncthreads();
}

template <int N>
__global__ __launch_bounds__(1024) void unrolled(double *out, double c)
{
    __shared__ double A[66 * 66], B[66 * 66];
    const int x = 67 + (threadIdx.x & 63) + 66 * (threadIdx.x >> 6);
    for (int k = threadIdx.x; k < 66 * 66; k += blockDim.x) A[k] = B[k] = 0.001 * k;
    __syncthreads();
    double u = c;
#pragma unroll
    for (int p = 0; p < N; p += 2) {
        phase(A, B, x, c + p, u);
        phase(B, A, x, c - p, u);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = u;
}

__global__ __launch_bounds__(1024) void looped(double *out, double c, int n)
{
    __shared__ double A[66 * 66], B[66 * 66];
    const int x = 67 + (threadIdx.x & 63) + 66 * (threadIdx.x >> 6);
    for (int k = threadIdx.x; k < 66 * 66; k += blockDim.x) A[k] = B[k] = 0.001 * k;
    __syncthreads();
    double u = c;
#pragma unroll 1
    for (int p = 0; p < n; p += 2) {
        phase(A, B, x, c + p, u);
        phase(B, A, x, c - p, u);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = u;
}

template <typename F>
float timed(F launch, hipStream_t s, hipEvent_t a, hipEvent_t b)
{
    float best = 1e9f, ms = 0;
    for (int rep = 0; rep < 20; rep++) {
        hipEventRecord(a, s);
        launch();
        hipEventRecord(b, s);
        hipStreamSynchronize(s);
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best * 1e3f;
}

int main()
{
    double *out;
    CK(hipMalloc(&out, sizeof(double) * 1024 * 512));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int wg : {1, 256}) {
        for (int threads : {1024, 512}) {
            const float e = timed([&] { hipLaunchKernelGGL(looped, dim3(wg), dim3(threads), 0, s, out, 1.0, 0); }, s, a, b);
            const float l = timed([&] { hipLaunchKernelGGL(looped, dim3(wg), dim3(threads), 0, s, out, 1.0, 48); }, s, a, b);
            const float u = timed([&] { hipLaunchKernelGGL((unrolled<48>), dim3(wg), dim3(threads), 0, s, out, 1.0); }, s, a, b);
            const float l2 = timed([&] { hipLaunchKernelGGL(looped, dim3(wg), dim3(threads), 0, s, out, 1.0, 192); }, s, a, b);
            const float u2 = timed([&] { hipLaunchKernelGGL((unrolled<192>), dim3(wg), dim3(threads), 0, s, out, 1.0); }, s, a, b);
            printf("%3d workgroups x %4d threads: empty %.1f us | 48 phases: looped %.1f, unrolled %.1f | 192 phases: looped %.1f, unrolled %.1f  (per phase: looped %.3f, unrolled %.3f us)\n",
                   wg, threads, e, l, u, l2, u2, (l2 - l) / 144, (u2 - u) / 144);
        }
    }
    return 0;
}
