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
// barriers.  Lane t of wave w handles rows 1+t, 65+t, ... of columns 1+w,
// 17+w, ...: arrays are column-major with odd leading dimension, so the 64
// lanes of a wave hit distinct banks, and no index needs a division.
//
// The levels are walked by a compile-time recursion so that each level's
// pointers and sizes live in registers.  The top level starts from a zero
// iterate, as every coarse level does when the down sweep reaches it
// (main.cpp:295-299,345).  Expressions are those of mgx_kernels.hip, evaluated
// in the same order: identical bits.
#include "mgx_coarse.h"
#include "mgx_internal.h"

#include <atomic>
#include <cstdlib>

namespace mgx {

namespace {

// u is the live iterate, w the idle array of the ping-pong pair.  (Two named
// pointers that are swapped, not an array indexed by a run-time "current"
// index: that would put the view in private memory.)
struct LevelView {
    double *f, *u, *w;
    int m;
    double h2, invh2;
};

// interior points (i, j), 1 <= i, j <= m-2, dealt over the workgroup: 16, 32 or 64 lanes
// along i, whatever covers a column, so that a 33x33 level takes one pass per phase, not two
__device__ __forceinline__ int tail_lanes_log2(int m) { return m - 2 <= 16 ? 4 : (m - 2 <= 32 ? 5 : 6); }
#define MGX_TAIL_FOR_INTERIOR(m_, i_, j_)                                                               \
    for (int lw_ = tail_lanes_log2(m_), j_ = 1 + (int)(threadIdx.x >> lw_); j_ < (m_)-1;                \
         j_ += (int)(blockDim.x >> lw_))                                                                \
        for (int i_ = 1 + (int)(threadIdx.x & ((1u << lw_) - 1)); i_ < (m_)-1; i_ += (1 << lw_))

__device__ __forceinline__ void tail_sweep(LevelView &L, double omega, double nomega)
{
    const int m = L.m;
    const double *__restrict__ u = L.u;
    double *__restrict__ o = L.w;
    MGX_TAIL_FOR_INTERIOR(m, i, j)
    {
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
    double *t = L.u;
    L.u = L.w;
    L.w = t;
}

// residual of the live iterate into the idle v array (its boundary is zero)
__device__ __forceinline__ void tail_residual(const LevelView &L)
{
    const int m = L.m;
    const double *__restrict__ u = L.u;
    double *__restrict__ r = L.w;
    MGX_TAIL_FOR_INTERIOR(m, i, j)
    {
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

__device__ __forceinline__ void tail_restrict(const LevelView &F, const LevelView &C)
{
    const int mf = F.m, mc = C.m;
    const double *__restrict__ r = F.w;
    MGX_TAIL_FOR_INTERIOR(mc, I, J)
    {
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

__device__ __forceinline__ void tail_prolong_add(const LevelView &F, const LevelView &C)
{
    const int mf = F.m, mc = C.m;
    const double *__restrict__ c = C.u;
    double *__restrict__ u = F.u;
    MGX_TAIL_FOR_INTERIOR(mf, i, j)
    {
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

// l is a compile-time index: a run-time index into the kernel-argument arrays would
// force the whole argument block into private (scratch) memory
template <int l>
__device__ __forceinline__ LevelView tail_view(const TailArgs &A, double *lds, int off)
{
    LevelView L;
    const int n = A.m[l] * A.m[l];
    L.f = lds + off;
    L.u = lds + off + n;
    L.w = lds + off + 2 * n;
    L.m = A.m[l];
    L.h2 = A.h2[l];
    L.invh2 = A.invh2[l];
    return L;
}

// Level l (whose f is in place and whose iterate is zero) and everything below
// it: down leg, recursion, up leg.  Returns with L.u naming the final iterate.
template <int l>
__device__ __forceinline__ void tail_level(const TailArgs &A, double *lds, int off, LevelView &L)
{
    if (l == A.nlev - 1) {  // main.cpp:331
        if (L.m == 3) {
            if (threadIdx.x == 0) {
                double *u = L.u;
                double s = u[0 + 1 * 3] + u[2 + 1 * 3];
                s = s + u[1 + 0 * 3];
                s = s + u[1 + 2 * 3];
                s = s - L.f[1 + 1 * 3] * L.h2;
                u[1 + 1 * 3] = s * 0.25;
            }
            __syncthreads();
        } else if (A.exact_tables) {
            // extension: exact solve of a coarsest grid that is not 3x3.  The level's idle
            // iterate array and, once read, its right-hand side serve as scratch.
            const int n = L.m - 2;
            const double *S = lds + A.tables_off;
            coarse_exact_solve(L.u, L.m, L.f, L.m, L.m, L.h2, S, S + n * n, L.w, L.f);
        }
        return;
    }
    if constexpr (l + 1 < TAIL_MAX_LEVELS) {
        for (int it = 0; it < A.Ndown; it++) tail_sweep(L, A.omega, A.nomega);  // main.cpp:321
        tail_residual(L);                                                       // main.cpp:325
        const int offc = off + 3 * L.m * L.m;
        LevelView C = tail_view<l + 1>(A, lds, offc);
        tail_restrict(L, C);                                                    // main.cpp:327
        tail_level<l + 1>(A, lds, offc, C);
        tail_prolong_add(L, C);                                                 // main.cpp:337-341
        for (int it = 0; it < A.Nup; it++) tail_sweep(L, A.omega, A.nomega);    // main.cpp:343
    }
}

__global__ __launch_bounds__(1024) void k_tail(const TailArgs A)
{
    extern __shared__ __align__(16) double lds[];
    const int tid = threadIdx.x;
    int total = 0;
#pragma unroll
    for (int l = 0; l < TAIL_MAX_LEVELS; l++)
        if (l < A.nlev) total += 3 * A.m[l] * A.m[l];
    for (int k = tid; k < total; k += (int)blockDim.x) lds[k] = 0.0;  // main.cpp:295-299
    if (A.exact_tables)
        for (int k = tid; k < A.tables_len; k += (int)blockDim.x) lds[A.tables_off + k] = A.exact_tables[k];
    __syncthreads();
    LevelView L0 = tail_view<0>(A, lds, 0);
    {
        const int m0 = A.m[0];
        for (int j = tid >> 6; j < m0; j += (int)(blockDim.x >> 6))
            for (int i = tid & 63; i < m0; i += 64) L0.f[i + j * m0] = A.f[(size_t)i + (size_t)j * (size_t)A.ld];
    }
    __syncthreads();
    tail_level<0>(A, lds, 0, L0);
    {
        const int m0 = A.m[0];
        const double *__restrict__ u = L0.u;
        MGX_TAIL_FOR_INTERIOR(m0, i, j) A.v[(size_t)i + (size_t)j * (size_t)A.ld] = u[i + j * m0];
    }
}

} // namespace

size_t tail_lds_bytes(const TailArgs &A)
{
    size_t n = 0;
    for (int l = 0; l < A.nlev; l++) n += 3 * (size_t)A.m[l] * (size_t)A.m[l];
    if (A.exact_tables && A.nlev >= 1) {
        const size_t nc = (size_t)A.m[A.nlev - 1] - 2;
        n += nc * nc + nc;
    }
    return n * sizeof(double);
}

hipError_t launch_tail(const TailArgs &A0, hipStream_t s)
{
    TailArgs A = A0;
    if (A.nlev < 1 || A.nlev > TAIL_MAX_LEVELS) return hipErrorInvalidValue;
    A.tables_off = 0;
    for (int l = 0; l < A.nlev; l++) A.tables_off += 3 * A.m[l] * A.m[l];
    const int nc = A.m[A.nlev - 1] - 2;
    A.tables_len = A.exact_tables ? nc * nc + nc : 0;
    const size_t bytes = tail_lds_bytes(A);
    if (bytes > TAIL_MAX_LDS_BYTES) return hipErrorInvalidValue;
    // a launch with more than 64 KB of dynamic LDS must be allowed once per device
    static std::atomic<unsigned long long> configured{0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = (dev >= 0 && dev < 64) ? (1ull << dev) : 0ull;
    if (!(configured.load(std::memory_order_acquire) & bit)) {
        e = hipFuncSetAttribute((const void *)k_tail, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TAIL_MAX_LDS_BYTES);
        if (e != hipSuccess) return e;
        configured.fetch_or(bit, std::memory_order_release);
    }
    static const int threads = [] {
        const char *e = getenv("MGX_TAIL_THREADS");
        int t = e ? atoi(e) : 1024;
        t = (t / 64) * 64;
        return t < 64 ? 64 : (t > 1024 ? 1024 : t);
    }();
    hipLaunchKernelGGL(k_tail, dim3(1), dim3((unsigned)threads), bytes, s, A);
    return hipGetLastError();
}

} // namespace mgx
