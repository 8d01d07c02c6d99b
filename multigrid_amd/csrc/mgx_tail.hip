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
// barriers.  Arrays are column-major with odd leading dimension, so the lanes of
// a wave hit distinct banks, and no index needs a division.
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

// -DMGX_TAIL_TRACE: thread 0 leaves the time (100 MHz clock) of every step of the launch in a
// device array, which mgx_debug_tail_trace() copies out -- how the cost of a phase was found
#ifdef MGX_TAIL_TRACE
__device__ unsigned long long g_tail_trace[128];
#define TT(k) do { if (threadIdx.x == 0) g_tail_trace[k] = wall_clock64(); } while (0)
#else
#define TT(k) do { } while (0)
#endif

namespace {

// u is the live iterate, w the idle array of the ping-pong pair.  (Two named
// pointers that are swapped, not an array indexed by a run-time "current"
// index: that would put the view in private memory.)
struct LevelView {
    double *f, *u, *w;
    int m;
    double h2, invh2;
};

// Interior points (i, j), 1 <= i, j <= m-2, dealt over the workgroup: 16, 32 or 64 lanes
// along i, whatever covers a column, the rest of the thread index along j.  With 1024 threads
// every level of at most 33x33 points is covered in ONE pass -- a thread has one point of the
// level or none -- and that is the case the phases are written for: each of them is a chain
// of dependent instructions of a single wave (index arithmetic, LDS round trip, eight
// dependent fp64 operations, barrier) and takes as long for a 5x5 level as for a 33x33 one:
// what a phase costs is the number of instructions on that chain (0.16-0.3 us as written
// here; 0.3-0.45 us with a loop nest over a thread's points in every phase).  A thread's
// point of a level is therefore worked out once per level (Pt), not once per phase, and the
// loops over further passes exist only where a level needs them.
__device__ __forceinline__ int tail_lanes_log2(int m) { return m - 2 <= 16 ? 4 : (m - 2 <= 32 ? 5 : 6); }

struct Pt {
    int i, j;  // the thread's first point of the level
    bool on;   // ... which lies inside the level
};

struct TailPar {
    int nthr, ndown, nup;
    double omega, nomega;
};

__device__ __forceinline__ Pt tail_point(int m, int nthr)
{
    const int lw = tail_lanes_log2(m);
    Pt P;
    P.i = 1 + (int)(threadIdx.x & ((1u << lw) - 1));
    P.j = 1 + (int)(threadIdx.x >> lw);
    P.on = P.i < m - 1 && P.j < m - 1;
    return P;
}

// does one pass of `nthr` threads cover a level of m x m points?  (host and device)
__host__ __device__ __forceinline__ bool tail_single_pass(int m, int nthr)
{
    const int lw = m - 2 <= 16 ? 4 : (m - 2 <= 32 ? 5 : 6);
    return m - 2 <= (1 << lw) && m - 2 <= (nthr >> lw);
}

// SINGLE: the launcher has checked that one pass covers the level
template <bool SINGLE, class Body>
__device__ __forceinline__ void tail_interior(int m, const Pt &P, int nthr, Body body)
{
    if constexpr (SINGLE) {
        if (P.on) body(P.i, P.j);
    } else {
        const int lw = tail_lanes_log2(m);
        for (int j = P.j; j < m - 1; j += nthr >> lw)
            for (int i = P.i; i < m - 1; i += 1 << lw) body(i, j);
    }
}

// a sweep applied to an iterate that is zero everywhere (main.cpp:194-195 with u = 0): it
// needs f at the point only, so whoever produces f can produce the first iterate with it
__device__ __forceinline__ double tail_sweep_from_zero(double f, double h2, double omega, double nomega)
{
    double s = 0.0 + 0.0;
    s = s + 0.0;
    s = s + 0.0;
    s = s - f * h2;
    s = s * 0.25;
    s = s * omega;
    return 0.0 * nomega + s;
}

template <bool SINGLE>
__device__ __forceinline__ void tail_sweep(LevelView &L, const Pt &P, const TailPar &T)
{
    const int m = L.m;
    const double *__restrict__ u = L.u;
    const double *__restrict__ f = L.f;
    double *__restrict__ o = L.w;
    const double h2 = L.h2, omega = T.omega, nomega = T.nomega;
    tail_interior<SINGLE>(m, P, T.nthr, [&](int i, int j) {
        const int x = i + j * m;
        double s = u[x - 1] + u[x + 1];  // main.cpp:194-195
        s = s + u[x - m];
        s = s + u[x + m];
        s = s - f[x] * h2;
        s = s * 0.25;
        s = s * omega;
        o[x] = u[x] * nomega + s;
    });
    __syncthreads();
    double *t = L.u;
    L.u = L.w;
    L.w = t;
}

// residual of the live iterate into the idle v array (its boundary is zero)
template <bool SINGLE>
__device__ __forceinline__ void tail_residual(const LevelView &L, const Pt &P, const TailPar &T)
{
    const int m = L.m;
    const double *__restrict__ u = L.u;
    const double *__restrict__ f = L.f;
    double *__restrict__ r = L.w;
    const double invh2 = L.invh2;
    tail_interior<SINGLE>(m, P, T.nthr, [&](int i, int j) {
        const int x = i + j * m;
        double s = u[x - 1] + u[x + 1];  // main.cpp:277
        s = s + u[x - m];
        s = s + u[x + m];
        s = s - u[x] * 4.0;
        s = s * invh2;
        r[x] = f[x] - s;
    });
    __syncthreads();
}

// PC: the thread's point of the coarse level.  seed: also store the coarse level's iterate
// after its first sweep from zero, so that the level's own first phase is its second sweep.
template <bool SINGLE>
__device__ __forceinline__ void tail_restrict(const LevelView &F, const LevelView &C, const Pt &PC, const TailPar &T, bool seed)
{
    const int mf = F.m, mc = C.m;
    const double *__restrict__ r = F.w;
    double *__restrict__ cf = C.f;
    double *__restrict__ cu = C.u;
    const double ch2 = C.h2, omega = T.omega, nomega = T.nomega;
    tail_interior<SINGLE>(mc, PC, T.nthr, [&](int I, int J) {
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
        const double val = (a + nb) + cn;
        cf[I + J * mc] = val;
        if (seed) cu[I + J * mc] = tail_sweep_from_zero(val, ch2, omega, nomega);
    });
    __syncthreads();
}

template <bool SINGLE>
__device__ __forceinline__ void tail_prolong_add(const LevelView &F, const LevelView &C, const Pt &P, const TailPar &T)
{
    const int mf = F.m, mc = C.m;
    const double *__restrict__ c = C.u;
    double *__restrict__ u = F.u;
    tail_interior<SINGLE>(mf, P, T.nthr, [&](int i, int j) {
        // main.cpp:234-246.  The four coarse points around (i, j) are always inside the coarse
        // level, so they are read whatever the parities and the case is picked by selects: the
        // sums are those of the reference's four branches, term by term.
        const int y = (i >> 1) + (j >> 1) * mc;
        const bool io = i & 1, jo = j & 1;
        const double c00 = c[y], c10 = c[y + 1], c01 = c[y + mc], c11 = c[y + mc + 1];
        const double t1 = c00 + (io ? c10 : c01);  // both half-way cases, and the first sum of the centre case
        double t3 = t1 + c01;
        t3 = t3 + c11;
        const double p = (io && jo) ? 0.25 * t3 : ((io || jo) ? 0.5 * t1 : c00);
        u[i + j * mf] = u[i + j * mf] + p;  // main.cpp:341
    });
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

// Level l (whose f is in place and whose iterate is zero or, with `seeded`, has had its
// first sweep) and everything below it: down leg, recursion, up leg.  Returns with L.u
// naming the final iterate.
// MODE: 1 = every level is covered by one pass of the workgroup, 2 = every level but the
// top one, 0 = no such promise.
template <int l, int MODE>
__device__ __forceinline__ void tail_level(const TailArgs &A, double *lds, int off, LevelView &L, const Pt &P,
                                           const TailPar &T, bool seeded)
{
    constexpr bool S = MODE == 1 || (MODE == 2 && l > 0);  // this level
    constexpr bool SC = MODE != 0;                          // the next coarser one
    TT(10 * l + 10);
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
        for (int it = seeded ? 1 : 0; it < T.ndown; it++) tail_sweep<S>(L, P, T);  // main.cpp:321
        TT(10 * l + 11);
        tail_residual<S>(L, P, T);                                              // main.cpp:325
        TT(10 * l + 12);
        const int offc = off + 3 * L.m * L.m;
        LevelView C = tail_view<l + 1>(A, lds, offc);
        const Pt PC = tail_point(C.m, T.nthr);
        // the coarse level sweeps next unless it is the coarsest one (main.cpp:331)
        const bool seed = T.ndown > 0 && l + 2 < A.nlev;
        tail_restrict<SC>(L, C, PC, T, seed);                                    // main.cpp:327
        TT(10 * l + 13);
        tail_level<l + 1, MODE>(A, lds, offc, C, PC, T, seed);
        TT(10 * l + 14);
        tail_prolong_add<S>(L, C, P, T);                                        // main.cpp:337-341
        TT(10 * l + 15);
        for (int it = 0; it < T.nup; it++) tail_sweep<S>(L, P, T);              // main.cpp:343
        TT(10 * l + 16);
    }
}

template <int MODE>
__global__ __launch_bounds__(1024) void k_tail(const TailArgs A)
{
    extern __shared__ __align__(16) double lds[];
    const int tid = threadIdx.x;
    TT(0);
    const TailPar T{(int)blockDim.x, A.Ndown, A.Nup, A.omega, A.nomega};
    // The top level's right-hand side is the only thing the launch waits for from memory: its
    // loads are issued first, all of a thread's at once, and land while LDS is being cleared.
    const int m0 = A.m[0];
    const size_t ld = (size_t)A.ld;
    const int li = tid & 63, lj = tid >> 6, dj = T.nthr >> 6;
    constexpr int NPRE = 5;  // 5 x 16 columns of 64 rows with 1024 threads: a 65 x 65 level but for its last row
    double pre[NPRE];
#pragma unroll
    for (int p = 0; p < NPRE; p++) {
        const int j = lj + p * dj;
        pre[p] = (li < m0 && j < m0) ? A.f[(size_t)li + (size_t)j * ld] : 0.0;
    }
    int total = 0;
#pragma unroll
    for (int l = 0; l < TAIL_MAX_LEVELS; l++)
        if (l < A.nlev) total += 3 * A.m[l] * A.m[l];
    for (int k = tid; k < total; k += T.nthr) lds[k] = 0.0;  // main.cpp:295-299
    if (A.exact_tables)
        for (int k = tid; k < A.tables_len; k += T.nthr) lds[A.tables_off + k] = A.exact_tables[k];
    __syncthreads();
    TT(1);
    LevelView L0 = tail_view<0>(A, lds, 0);
    const bool seed0 = T.ndown > 0 && A.nlev > 1;
    {
        // the top level's first sweep from zero comes with its right-hand side
        auto put = [&](int i, int j, double f) {
            L0.f[i + j * m0] = f;
            if (seed0 && i >= 1 && i < m0 - 1 && j >= 1 && j < m0 - 1)
                L0.u[i + j * m0] = tail_sweep_from_zero(f, L0.h2, T.omega, T.nomega);
        };
#pragma unroll
        for (int p = 0; p < NPRE; p++) {
            const int j = lj + p * dj;
            if (li < m0 && j < m0) put(li, j, pre[p]);
        }
        // what the registers above did not cover: rows from 64 on, columns from NPRE * dj on
        for (int j = lj; j < m0; j += dj)
            for (int i = li + (j < lj + NPRE * dj ? 64 : 0); i < m0; i += 64) put(i, j, A.f[(size_t)i + (size_t)j * ld]);
    }
    __syncthreads();
    TT(2);
    const Pt P0 = tail_point(L0.m, T.nthr);
    tail_level<0, MODE>(A, lds, 0, L0, P0, T, seed0);
    TT(3);
    {
        const double *__restrict__ u = L0.u;
        double *__restrict__ v = A.v;
        tail_interior<MODE == 1>(m0, P0, T.nthr, [&](int i, int j) { v[(size_t)i + (size_t)j * ld] = u[i + j * m0]; });
    }
    TT(4);
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
    static const int threads = [] {
        const char *e = getenv("MGX_TAIL_THREADS");
        int t = e ? atoi(e) : 1024;
        t = (t / 64) * 64;
        return t < 64 ? 64 : (t > 1024 ? 1024 : t);
    }();
    // which levels one pass of the workgroup covers decides the instance (see tail_level)
    bool all = true, below_top = true;
    for (int l = 0; l < A.nlev; l++) {
        const bool one = tail_single_pass(A.m[l], threads);
        all = all && one;
        if (l > 0) below_top = below_top && one;
    }
    const int mode = all ? 1 : (below_top ? 2 : 0);
    const void *kernel = mode == 1 ? (const void *)k_tail<1> : (mode == 2 ? (const void *)k_tail<2> : (const void *)k_tail<0>);
    // a launch with more than 64 KB of dynamic LDS must be allowed once per kernel and device
    static std::atomic<unsigned long long> configured[3];
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = (dev >= 0 && dev < 64) ? (1ull << dev) : 0ull;
    if (!(configured[mode].load(std::memory_order_acquire) & bit)) {
        e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TAIL_MAX_LDS_BYTES);
        if (e != hipSuccess) return e;
        configured[mode].fetch_or(bit, std::memory_order_release);
    }
    switch (mode) {
    case 1: hipLaunchKernelGGL(k_tail<1>, dim3(1), dim3((unsigned)threads), bytes, s, A); break;
    case 2: hipLaunchKernelGGL(k_tail<2>, dim3(1), dim3((unsigned)threads), bytes, s, A); break;
    default: hipLaunchKernelGGL(k_tail<0>, dim3(1), dim3((unsigned)threads), bytes, s, A); break;
    }
    return hipGetLastError();
}

} // namespace mgx

#ifdef MGX_TAIL_TRACE
extern "C" int mgx_debug_tail_trace(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mgx::g_tail_trace), sizeof(unsigned long long) * 128);
}
#endif
