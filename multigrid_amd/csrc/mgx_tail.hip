// mgx_tail.hip -- the bottom of the V-cycle in ONE launch.
//
// Levels of at most 65x65 points are latency-bound: a launch per leg costs more
// than its arithmetic.  k_tail runs every such level of a cycle -- down legs,
// the coarsest solve and up legs (reference main.cpp:317-346 restricted to
// those levels) -- inside one 1024-thread workgroup with the whole sub-hierarchy
// resident in LDS: per level one f and two v arrays (Jacobi ping-pong; the idle
// one doubles as the residual), 3*8*(65^2+33^2+17^2+9^2+5^2+3^2) = 137 KB of
// the CU's 160 KB.  The top level's f is read from HBM once and its final v is
// written once; everything in between is LDS traffic separated by workgroup
// barriers.  Arrays are column-major with odd leading dimension, so lanes
// running along i hit distinct banks.
//
// The top level starts from a zero iterate, as every coarse level does when the
// down sweep reaches it (main.cpp:295-299,345).  Expressions are those of
// mgx_kernels.hip, evaluated in the same order: identical bits.
#include "mgx_internal.h"

namespace mgx {

namespace {

struct LevelView {
    double *f, *v[2];
    int m, cur;
    double h2, invh2;
};

__device__ __forceinline__ void tail_sweep(LevelView &L, double omega, double nomega, int tid, int nt)
{
    const int m = L.m, mi = m - 2;
    const double *__restrict__ u = L.v[L.cur];
    double *__restrict__ o = L.v[L.cur ^ 1];
    for (int k = tid; k < mi * mi; k += nt) {
        const int j = 1 + k / mi, i = 1 + k - (j - 1) * mi;
        const int x = i + j * m;
        double s = u[x - 1] + u[x + 1];  // main.cpp:194-195
        s = s + u[x - m];
        s = s + u[x + m];
        s = s - L.f[x] * L.h2;
        s = s * 0.25;
        s = s * omega;
        o[x] = u[x] * nomega + s;
    }
    __syncthreads();
    L.cur ^= 1;
}

// residual of the live iterate into the idle v array (its boundary is zero)
__device__ __forceinline__ void tail_residual(const LevelView &L, int tid, int nt)
{
    const int m = L.m, mi = m - 2;
    const double *__restrict__ u = L.v[L.cur];
    double *__restrict__ r = L.v[L.cur ^ 1];
    for (int k = tid; k < mi * mi; k += nt) {
        const int j = 1 + k / mi, i = 1 + k - (j - 1) * mi;
        const int x = i + j * m;
        double s = u[x - 1] + u[x + 1];  // main.cpp:277
        s = s + u[x - m];
        s = s + u[x + m];
        s = s - u[x] * 4.0;
        s = s * L.invh2;
        r[x] = L.f[x] - s;
    }
    __syncthreads();
}

__device__ __forceinline__ void tail_restrict(const LevelView &F, LevelView &C, int tid, int nt)
{
    const int mf = F.m, mc = C.m, mi = mc - 2;
    const double *__restrict__ r = F.v[F.cur ^ 1];
    for (int k = tid; k < mi * mi; k += nt) {
        const int J = 1 + k / mi, I = 1 + k - (J - 1) * mi;
        const int x = 2 * I + 2 * J * mf;
        const double a = r[x] * 0.25;  // main.cpp:216-218, corner quirk included
        double nb = r[x - 1] + r[x + 1];
        nb = nb + r[x - mf];
        nb = nb + r[x + mf];
        nb = nb * 0.125;
        double cn = r[x - 1 - mf] + r[x + 1 - mf];
        cn = cn + r[x + 1 - mf];
        cn = cn + r[x + 1 + mf];
        cn = cn * 0.0625;
        C.f[I + J * mc] = (a + nb) + cn;
    }
    __syncthreads();
}

__device__ __forceinline__ void tail_prolong_add(LevelView &F, const LevelView &C, int tid, int nt)
{
    const int mf = F.m, mc = C.m, mi = mf - 2;
    const double *__restrict__ c = C.v[C.cur];
    double *__restrict__ u = F.v[F.cur];
    for (int k = tid; k < mi * mi; k += nt) {
        const int j = 1 + k / mi, i = 1 + k - (j - 1) * mi;
        const int y = (i >> 1) + (j >> 1) * mc;
        const bool io = i & 1, jo = j & 1;
        double p;  // main.cpp:234-246
        if (!io && !jo) {
            p = c[y];
        } else if (!io && jo) {
            p = 0.5 * (c[y] + c[y + mc]);
        } else if (io && !jo) {
            p = 0.5 * (c[y] + c[y + 1]);
        } else {
            double s = c[y] + c[y + 1];
            s = s + c[y + mc];
            s = s + c[y + mc + 1];
            p = 0.25 * s;
        }
        u[i + j * mf] = u[i + j * mf] + p;  // main.cpp:341
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void k_tail(const TailArgs A)
{
    extern __shared__ __align__(16) double lds[];
    const int tid = threadIdx.x, nt = blockDim.x;
    LevelView L[TAIL_MAX_LEVELS];
    int total = 0;
#pragma unroll
    for (int l = 0; l < TAIL_MAX_LEVELS; l++) {
        if (l < A.nlev) {
            const int n = A.m[l] * A.m[l];
            L[l].f = lds + total;
            L[l].v[0] = lds + total + n;
            L[l].v[1] = lds + total + 2 * n;
            L[l].m = A.m[l];
            L[l].cur = 0;
            L[l].h2 = A.h2[l];
            L[l].invh2 = A.invh2[l];
            total += 3 * n;
        }
    }
    for (int k = tid; k < total; k += nt) lds[k] = 0.0;  // main.cpp:295-299
    __syncthreads();
    {
        const int m0 = A.m[0];
        for (int k = tid; k < m0 * m0; k += nt) {
            const int j = k / m0, i = k - j * m0;
            L[0].f[k] = A.f[(size_t)i + (size_t)j * (size_t)A.ld];
        }
    }
    __syncthreads();
    const double omega = A.omega, nomega = A.nomega;
#pragma unroll
    for (int l = 0; l < TAIL_MAX_LEVELS - 1; l++) {  // main.cpp:317-328
        if (l < A.nlev - 1) {
            for (int it = 0; it < A.Ndown; it++) tail_sweep(L[l], omega, nomega, tid, nt);
            tail_residual(L[l], tid, nt);
            tail_restrict(L[l], L[l + 1], tid, nt);
        }
    }
#pragma unroll
    for (int l = 0; l < TAIL_MAX_LEVELS; l++) {  // main.cpp:331
        if (l == A.nlev - 1 && L[l].m == 3) {
            if (tid == 0) {
                double *u = L[l].v[L[l].cur];
                double s = u[0 + 1 * 3] + u[2 + 1 * 3];
                s = s + u[1 + 0 * 3];
                s = s + u[1 + 2 * 3];
                s = s - L[l].f[1 + 1 * 3] * L[l].h2;
                u[1 + 1 * 3] = s * 0.25;
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int l = TAIL_MAX_LEVELS - 2; l >= 0; l--) {  // main.cpp:335-346
        if (l < A.nlev - 1) {
            tail_prolong_add(L[l], L[l + 1], tid, nt);
            for (int it = 0; it < A.Nup; it++) tail_sweep(L[l], omega, nomega, tid, nt);
        }
    }
    {
        const int m0 = A.m[0], mi = m0 - 2;
        const double *__restrict__ u = L[0].v[L[0].cur];
        for (int k = tid; k < mi * mi; k += nt) {
            const int j = 1 + k / mi, i = 1 + k - (j - 1) * mi;
            A.v[(size_t)i + (size_t)j * (size_t)A.ld] = u[i + j * m0];
        }
    }
}

} // namespace

size_t tail_lds_bytes(const TailArgs &A)
{
    size_t n = 0;
    for (int l = 0; l < A.nlev; l++) n += 3 * (size_t)A.m[l] * (size_t)A.m[l];
    return n * sizeof(double);
}

hipError_t launch_tail(const TailArgs &A, hipStream_t s)
{
    if (A.nlev < 1 || A.nlev > TAIL_MAX_LEVELS) return hipErrorInvalidValue;
    const size_t bytes = tail_lds_bytes(A);
    if (bytes > TAIL_MAX_LDS_BYTES) return hipErrorInvalidValue;
    // more than 64 KB of dynamic LDS must be requested once per device
    static bool configured[64] = {false};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !configured[dev]) {
        e = hipFuncSetAttribute((const void *)k_tail, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)TAIL_MAX_LDS_BYTES);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) configured[dev] = true;
    }
    hipLaunchKernelGGL(k_tail, dim3(1), dim3(1024), bytes, s, A);
    return hipGetLastError();
}

} // namespace mgx
