// How fast does one MI355X issue the vector fp64 instructions the sweeps are made of?
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/fp64_rate tools/fp64_rate.hip && tools/fp64_rate
// Every wave runs ITER rounds of 16 independent v_add_f64 (or v_mul_f64, v_fma_f64, or 64-bit
// DPP moves); with 1, 2, 4, 8 waves per SIMD.  Prints wave-instructions per SIMD and microsecond
// and the cycles per instruction that follow from the shader clock measured beside it
// (s_memtime against the 100 MHz s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void k_rate(double *out, int iter, double a, double b, unsigned long long *clk)
{
    double r[16], q[16], p[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        r[k] = a + k + threadIdx.x;
        q[k] = b + 1e-9 * (k + threadIdx.x);  // operands that live in vector registers and never change
        p[k] = a - 1e-9 * (k + threadIdx.x);
    }
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iter; it++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (OP == 0) r[k] = r[k] + b;
            if (OP == 1) r[k] = r[k] * b;
            if (OP == 2) r[k] = __builtin_fma(r[k], b, a);
            if (OP == 3) {  // a 64-bit lane shift: two DPP moves
                int lo = __double2loint(r[k]), hi = __double2hiint(r[k]);
                lo = __builtin_amdgcn_update_dpp(lo, lo, 0x111, 0xf, 0xf, false);  // row_shr:1
                hi = __builtin_amdgcn_update_dpp(hi, hi, 0x111, 0xf, 0xf, false);
                r[k] = __hiloint2double(hi, lo);
            }
            if (OP == 4) r[k] = __builtin_fma(r[k], b, r[(k + 1) & 15]);  // dependent on a neighbour: still 16 chains
            if (OP == 5) r[k] = r[k] + q[k];                              // two vector-register sources
            if (OP == 6) r[k] = r[k] * q[k];
            if (OP == 7) r[k] = __builtin_fma(r[k], q[k], p[k]);          // three
            if (OP == 8) r[k] = __builtin_fma(r[k], b, p[k]);             // two, one scalar
        }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    double s = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) s += r[k] + q[k] + p[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = c1 - c0;
        clk[1] = w1 - w0;
    }
}

template <int OP>
int run(const char *name, int ncu, double *out, unsigned long long *clk)
{
    const int iter = 20000;
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = ncu * wps;  // 256 threads = one wave per SIMD and workgroup
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, out, 1000, 1.0, 1.0000001, clk);
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, out, iter, 1.0, 1.0000001, clk);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[2];
        CHECK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
        const double mhz = (double)h[0] / ((double)h[1] / 100.0);
        const double per_simd = (double)iter * 16 * wps * (OP == 3 ? 2 : 1);  // wave-instructions a SIMD issued
        const double us = ms * 1e3;
        printf("%-10s %d wave(s) per SIMD: %7.1f us, %6.1f wave-instructions per SIMD and us, clock %4.0f MHz -> %.2f cycles per instruction\n",
               name, wps, us, per_simd / us, mhz, us * mhz / per_simd);
    }
    return 0;
}

int main()
{
    int ncu = 0;
    CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    double *out;
    unsigned long long *clk;
    CHECK(hipMalloc(&out, sizeof(double) * 256 * ncu * 8));
    CHECK(hipMalloc(&clk, 16));
    printf("%d CUs\n", ncu);
    if (run<0>("v_add_f64", ncu, out, clk)) return 1;
    if (run<1>("v_mul_f64", ncu, out, clk)) return 1;
    if (run<2>("v_fma_f64", ncu, out, clk)) return 1;
    if (run<4>("fma chain", ncu, out, clk)) return 1;
    if (run<3>("dpp mov x2", ncu, out, clk)) return 1;
    if (run<5>("add v,v", ncu, out, clk)) return 1;
    if (run<6>("mul v,v", ncu, out, clk)) return 1;
    if (run<7>("fma v,v,v", ncu, out, clk)) return 1;
    if (run<8>("fma v,s,v", ncu, out, clk)) return 1;
    return 0;
}
