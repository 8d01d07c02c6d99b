// What does a chain of small dependent kernels cost per kernel when it is enqueued launch by
// launch, and when the same chain is replayed from a hipGraph?  (Would capturing the small
// levels of a cycle in a graph remove part of the ~4.7 us every launch costs?)
//   hipcc --offload-arch=gfx950 -O2 tools/graph_probe.hip -o <binary>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void tiny(double *p, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 0.5 + 1.0;
}
int main()
{
    const int N = 200, n = 256 * 256;
    double *p;
    CK(hipMalloc(&p, sizeof(double) * n));
    CK(hipMemset(p, 0, sizeof(double) * n));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int wg : {1, 256, 1024}) {
        float ms = 0;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(a, s));
            for (int k = 0; k < N; k++) hipLaunchKernelGGL(tiny, dim3(wg), dim3(256), 0, s, p, n);
            CK(hipEventRecord(b, s));
            CK(hipStreamSynchronize(s));
            CK(hipEventElapsedTime(&ms, a, b));
        }
        printf("stream, %4d workgroups: %.2f us per kernel\n", wg, ms * 1e3 / N);
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int k = 0; k < N; k++) hipLaunchKernelGGL(tiny, dim3(wg), dim3(256), 0, s, p, n);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(a, s));
            CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(b, s));
            CK(hipStreamSynchronize(s));
            CK(hipEventElapsedTime(&ms, a, b));
        }
        printf("graph,  %4d workgroups: %.2f us per kernel\n", wg, ms * 1e3 / N);
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    return 0;
}
