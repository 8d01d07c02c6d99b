// Does gfx950 execute the wavefront-wide DPP shifts (wave_shr:1 / wave_shl:1) the assembler
// accepts?  They would replace the ds_bpermute pair behind a 64-bit __shfl_up/__shfl_down by two
// VALU moves.   hipcc --offload-arch=gfx950 -O2 tools/dpp_probe.hip -o gpurun_out/dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *up, int *down, const int *in)
{
    const int v = in[threadIdx.x];
    up[threadIdx.x] = __builtin_amdgcn_update_dpp(-7, v, 0x138, 0xf, 0xf, false);    // wave_shr:1: lane t <- lane t-1
    down[threadIdx.x] = __builtin_amdgcn_update_dpp(-7, v, 0x130, 0xf, 0xf, false);  // wave_shl:1: lane t <- lane t+1
}
int main()
{
    int h[128], hu[128], hd[128], *in, *up, *down;
    for (int i = 0; i < 128; i++) h[i] = 100 + i;
    hipMalloc(&in, sizeof h);
    hipMalloc(&up, sizeof h);
    hipMalloc(&down, sizeof h);
    hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(128), 0, 0, up, down, in);
    hipMemcpy(hu, up, sizeof h, hipMemcpyDeviceToHost);
    hipMemcpy(hd, down, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 128; i++) {
        const int lane = i & 63;
        const int wu = lane == 0 ? -7 : h[i - 1], wd = lane == 63 ? -7 : h[i + 1];
        if (hu[i] != wu || hd[i] != wd) {
            if (bad < 8) printf("lane %d: up %d (want %d) down %d (want %d)\n", i, hu[i], wu, hd[i], wd);
            bad++;
        }
    }
    printf("wave_shr/wave_shl: %d of 128 lanes wrong\n", bad);
    return bad != 0;
}
