// mgx_mid.hip -- several small levels of the V-cycle in ONE launch.
//
// Between the big levels (bandwidth-bound, one streaming launch per leg) and the
// LDS-resident tail (one workgroup) lie levels of a few hundred to a thousand
// points per side.  A launch per leg costs them 6-15 us each although they hold
// almost no work: the column march of the streaming kernel is a chain of
// dependent loads.  The two kernels here take a GROUP of consecutive levels per
// launch; both are for levels whose iterate is zero when the down sweep arrives
// (every level >= 1, reference main.cpp:295-299,345), with the "lean" bookkeeping
// of the streaming schedule: the down sweep stores no iterate, the up sweep
// recomputes the pre-smoothing from the right-hand side.
//
//   k_mid_down<NL>  f of level 0  ->  f of levels 1..NL
//       per level: Ndown sweeps from zero, residual, restriction
//       (relax + residual + coarsen, main.cpp:321-327)
//   k_mid_up<NL>    v of level NL, f of levels 0..NL-1  ->  v of level 0
//       per level, coarse to fine: Ndown sweeps from zero, += P(coarse v),
//       Nup sweeps (refine + `cur += refined` + relax, main.cpp:337-344)
//
// A workgroup owns a rectangular tile of the OUTPUT level and computes every point
// of every level that its tile depends on (overlapping tiles recompute their
// common points; Jacobi makes the copies bit-identical).  Going down the
// dependence region roughly doubles per level (a tile of T coarse points needs
// 2T + 2(Ndown+1) fine points), so down launches take two levels; going up it
// converges (a tile of T fine points needs T/2 + ~4 coarse points), so one up
// launch can take every level of the group.  A thread keeps the right-hand side
// and the iterate of its (at most Q) points of a level in registers from phase to
// phase; LDS carries the iterate to the neighbours, one workgroup barrier per
// sweep.  The kernels are bound by the latency of that chain of phases (about
// 0.4 us each on MI355X), not by throughput: shapes are chosen so that every
// workgroup of a launch is resident at once.
//
// LDS arrays hold a region with a one-cell border of zeros.  Where the region
// touches the edge of the grid the border IS the boundary (zero: the iterate
// and the residual of a coarse level have a zero boundary); elsewhere it only
// keeps neighbour reads inside the array, and what is computed from it never
// reaches a point the tile is responsible for: the regions are grown by exactly
// the number of cells each sweep, residual and transfer consumes.
//
// Expressions are those of mgx_kernels.hip / mgx_tail.hip, evaluated in the same
// order: identical bits.
#include "../../include/mgx.h"
#include "mgx_internal.h"

#include <map>
#include <mutex>
#include <tuple>

#include <cstdlib>

namespace mgx {

// -DMGX_MID_TRACE: thread 0 of the middle workgroup of the launches whose level 0 has
// g_mid_trace_m rows leaves the time (100 MHz clock) of every step in a device array
// (row 0: down launch, row 1: up launch); mgx_debug_mid_trace() selects and copies out
#ifdef MGX_MID_TRACE
__device__ unsigned long long g_mid_trace[2][64];
__device__ int g_mid_trace_m;
#define MT(row, k)                                                                                          \
    do {                                                                                                    \
        if (threadIdx.x == 0 && (int)blockIdx.x == A.nti / 2 && (int)blockIdx.y == A.ntj / 2 &&             \
            A.m[0] == g_mid_trace_m)                                                                        \
            g_mid_trace[row][k] = wall_clock64();                                                           \
    } while (0)
#else
#define MT(row, k) do { } while (0)
#endif

namespace {

struct Span {
    int lo, hi;  // inclusive
};

__host__ __device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__host__ __device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__host__ __device__ __forceinline__ int width(const Span &s) { return s.hi - s.lo + 1; }

// Along one axis, tile t of `tile` points of level NL: the points of each level whose
// right-hand side the workgroup needs (need[k]) and the points it writes (own[k]; the
// tiles' own ranges partition every level: first point of a tile = 2 * (first point on
// the level below) - 1).
template <int NL>
__host__ __device__ __forceinline__ void down_spans(const MidArgs &A, int t, int tile, int nt, Span (&need)[NL + 1],
                                                    Span (&own)[NL + 1])
{
    own[NL].lo = 1 + t * tile;
    own[NL].hi = (t == nt - 1) ? A.m[NL] - 2 : own[NL].lo + tile - 1;
    need[NL] = own[NL];
#pragma unroll
    for (int k = NL - 1; k >= 0; k--) {
        own[k].lo = (t == 0) ? 1 : 2 * own[k + 1].lo - 1;
        own[k].hi = (t == nt - 1) ? A.m[k] - 2 : 2 * own[k + 1].hi;
        // restriction reads the residual at 2I-1..2I+1, the residual the iterate one further
        // out, and each sweep after the first (which needs f at the point only) one more
        need[k].lo = imax(1, 2 * need[k + 1].lo - 1 - A.ndown);
        need[k].hi = imin(A.m[k] - 2, 2 * need[k + 1].hi + 1 + A.ndown);
    }
}

// Along one axis, tile t of `tile` points of level 0: the points of each level the
// workgroup computes (G[k]) and the points of level NL it reads (cload, boundary included).
template <int NL>
__host__ __device__ __forceinline__ void up_spans(const MidArgs &A, int t, int tile, int nt, Span (&G)[NL], Span &cload,
                                                  Span &own)
{
    // Nup sweeps after the correction, and the Ndown sweeps before it (the first is pointwise)
    const int halo = A.nup + (A.ndown > 1 ? A.ndown - 1 : 0);
    own.lo = 1 + t * tile;
    own.hi = (t == nt - 1) ? A.m[0] - 2 : own.lo + tile - 1;
    Span res = own;
#pragma unroll
    for (int k = 0; k < NL; k++) {
        G[k].lo = imax(1, res.lo - halo);
        G[k].hi = imin(A.m[k] - 2, res.hi + halo);
        const Span cr{G[k].lo >> 1, (G[k].hi + 1) >> 1};  // coarse points the interpolation reads
        if (k + 1 < NL) {
            res.lo = imax(1, cr.lo);
            res.hi = imin(A.m[k + 1] - 2, cr.hi);
        } else {
            cload = cr;
        }
    }
}

// ---- device side ---------------------------------------------------------------------

struct Lay {
    int i0, j0, ld;  // grid index of the border cell (0, 0); leading dimension
};
__device__ __forceinline__ Lay lay_of(const Span &si, const Span &sj) { return Lay{si.lo - 1, sj.lo - 1, width(si) + 2}; }
__device__ __forceinline__ int at(const Lay &L, int i, int j) { return (i - L.i0) + (j - L.j0) * L.ld; }

// The points of a region a thread is responsible for: (i, j + q*dj), q < Q.  Lanes run
// along i (16, 32 or 64 of them per column, whatever covers the region's width), the rest
// of the thread index along j; the launcher sizes the tiles so that Q passes cover the
// region.  The thread keeps f and the iterate at its points in registers from phase to
// phase; LDS only carries the iterate to the neighbours.
struct Pts {
    int i, j, dj, jhi;  // grid indices of point 0, column step, last column of the region
    int x, xs;          // LDS index of point 0 and its step per q
    bool in_i;
    __device__ __forceinline__ bool ok(int q) const { return in_i && j + q * dj <= jhi; }
    __device__ __forceinline__ int col(int q) const { return j + q * dj; }
};
__device__ __forceinline__ Pts pts_of(const Span &si, const Span &sj, const Lay &L)
{
    const int wx = width(si);
    const int lw = wx <= 16 ? 4 : (wx <= 32 ? 5 : 6);
    const int t = (int)threadIdx.x;
    Pts P;
    P.i = si.lo + (t & ((1 << lw) - 1));
    P.j = sj.lo + (t >> lw);
    P.dj = (int)blockDim.x >> lw;
    P.jhi = sj.hi;
    P.in_i = P.i <= si.hi;
    P.x = at(L, P.i, P.j);
    P.xs = P.dj * L.ld;
    return P;
}

__device__ __forceinline__ void zero_border(double *U, double *W, const Span &sj, const Lay &L)
{
    const int nx = L.ld, ny = width(sj) + 2;
    for (int t = (int)threadIdx.x; t < nx; t += (int)blockDim.x) {
        U[t] = 0.0;
        W[t] = 0.0;
        U[t + (ny - 1) * nx] = 0.0;
        W[t + (ny - 1) * nx] = 0.0;
    }
    for (int t = (int)threadIdx.x; t < ny; t += (int)blockDim.x) {
        U[t * nx] = 0.0;
        W[t * nx] = 0.0;
        U[nx - 1 + t * nx] = 0.0;
        W[nx - 1 + t * nx] = 0.0;
    }
}

// a sweep applied to an iterate that is zero everywhere (main.cpp:194-195 with u = 0)
__device__ __forceinline__ double jacobi_from_zero(double f, const SweepCoef &c)
{
    double s = 0.0 + 0.0;
    s = s + 0.0;
    s = s + 0.0;
    s = s - f * c.h2;
    s = s * 0.25;
    s = s * c.omega;
    return 0.0 * c.nomega + s;
}

// Interpolated coarse value at fine point (i, j); C holds the coarse level in layout LC.
// The four coarse cells around the point lie inside the array whatever the parities (at worst
// in its border), so all four are read at once and the case is picked by selects instead of
// four divergent branches with an LDS round trip each: the sums are those of the reference's
// branches, term by term (main.cpp:234-246).
__device__ __forceinline__ double prolong_at(const double *__restrict__ C, const Lay &LC, int i, int j)
{
    const int y = at(LC, i >> 1, j >> 1);
    const bool io = i & 1, jo = j & 1;
    const double c00 = C[y], c10 = C[y + 1], c01 = C[y + LC.ld], c11 = C[y + LC.ld + 1];
    const double t1 = c00 + (io ? c10 : c01);  // both half-way cases, and the first sum of the centre case
    double t3 = t1 + c01;
    t3 = t3 + c11;
    return (io && jo) ? 0.25 * t3 : ((io || jo) ? 0.5 * t1 : c00);
}

// the four neighbours of every point of the thread, requested together so that their LDS
// latencies overlap (points outside the region read cell (1, 1) and are ignored)
template <int Q>
struct Nbr {
    double w[Q], e[Q], s[Q], n[Q];
};
template <int Q>
__device__ __forceinline__ void read_neighbours(const double *__restrict__ U, const Pts &P, const Lay &L, Nbr<Q> &N)
{
#pragma unroll
    for (int q = 0; q < Q; q++) {
        const int x = P.ok(q) ? P.x + q * P.xs : L.ld + 1;
        N.w[q] = U[x - 1];
        N.e[q] = U[x + 1];
        N.s[q] = U[x - L.ld];
        N.n[q] = U[x + L.ld];
    }
}

// one sweep U -> O at the thread's points (main.cpp:194-195); with C, the coarse-grid
// correction is added to the result (main.cpp:341)
template <int Q>
__device__ __forceinline__ void sweep_pts(const double *__restrict__ U, double *__restrict__ O, const Pts &P, const Lay &L,
                                          const SweepCoef &c, const double (&f)[Q], double (&u)[Q],
                                          const double *__restrict__ C, const Lay &LC)
{
    Nbr<Q> N;
    read_neighbours(U, P, L, N);
#pragma unroll
    for (int q = 0; q < Q; q++) {
        if (P.ok(q)) {
            double s = N.w[q] + N.e[q];
            s = s + N.s[q];
            s = s + N.n[q];
            s = s - f[q] * c.h2;
            s = s * 0.25;
            s = s * c.omega;
            double v = u[q] * c.nomega + s;
            if (C) v = v + prolong_at(C, LC, P.i, P.col(q));
            u[q] = v;
            O[P.x + q * P.xs] = v;
        }
    }
    __syncthreads();
}

template <int NL, int Q>
__global__ __launch_bounds__(1024) void k_mid_down(const MidArgs A)
{
    extern __shared__ __align__(16) double lds[];
    double *U = lds;
    double *W = lds + A.stride;
    MT(0, 0);
    Span ni[NL + 1], oi[NL + 1], nj[NL + 1], oj[NL + 1];
    down_spans<NL>(A, (int)blockIdx.x, A.tile_i, A.nti, ni, oi);
    down_spans<NL>(A, (int)blockIdx.y, A.tile_j, A.ntj, nj, oj);
    double f[Q], u[Q];
    {
        const Lay L = lay_of(ni[0], nj[0]);
        const Pts P = pts_of(ni[0], nj[0], L);
        const double *__restrict__ g = A.f[0];
        const size_t m0 = (size_t)A.m[0];
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const bool ok = P.ok(q);
            const int ic = ok ? P.i : ni[0].lo, jc = ok ? P.col(q) : nj[0].lo;
            f[q] = g[(size_t)ic + (size_t)jc * m0];
        }
    }
#pragma unroll
    for (int k = 0; k < NL; k++) {
        const Span si = ni[k], sj = nj[k];
        const Lay L = lay_of(si, sj);
        const Pts P = pts_of(si, sj, L);
        const SweepCoef c{A.h2[k], A.invh2[k], A.omega, A.nomega};
        MT(0, 10 * k + 1);
        zero_border(U, W, sj, L);
        // the first sweep from zero needs f at the point only (main.cpp:321)
#pragma unroll
        for (int q = 0; q < Q; q++) {
            if (P.ok(q)) {
                u[q] = A.ndown > 0 ? jacobi_from_zero(f[q], c) : 0.0;
                U[P.x + q * P.xs] = u[q];
            }
        }
        __syncthreads();
        MT(0, 10 * k + 2);
        for (int s = 1; s < A.ndown; s++) {
            sweep_pts(U, W, P, L, c, f, u, nullptr, L);
            double *t = U;
            U = W;
            W = t;
        }
        MT(0, 10 * k + 3);
        {  // main.cpp:325: the residual, into the idle array
            Nbr<Q> N;
            read_neighbours(U, P, L, N);
#pragma unroll
            for (int q = 0; q < Q; q++) {
                if (P.ok(q)) {
                    double s = N.w[q] + N.e[q];  // main.cpp:277
                    s = s + N.s[q];
                    s = s + N.n[q];
                    s = s - u[q] * 4.0;
                    s = s * c.invh2;
                    W[P.x + q * P.xs] = f[q] - s;
                }
            }
            __syncthreads();
        }
        MT(0, 10 * k + 4);
        // main.cpp:327: every thread restricts at ITS points of the next level, whose
        // right-hand side therefore never leaves the registers
        const Span ci = ni[k + 1], cj = nj[k + 1];
        const Lay LC = lay_of(ci, cj);
        const Pts PC = pts_of(ci, cj, LC);
        double *__restrict__ go = A.fo[k + 1];
        const size_t mc = (size_t)A.m[k + 1];
        const double *__restrict__ r = W;
        const int ld = L.ld;
#pragma unroll
        for (int q = 0; q < Q; q++) {
            if (PC.ok(q)) {
                const int I = PC.i, J = PC.col(q);
                const int x = at(L, 2 * I, 2 * J);
                const double a = r[x] * 0.25;  // main.cpp:216-218, corner quirk included
                double nb = r[x - 1] + r[x + 1];
                nb = nb + r[x - ld];
                nb = nb + r[x + ld];
                nb = nb * 0.125;
                double cn = r[x - 1 - ld] + r[x + 1 - ld];
                cn = cn + r[x + 1 - ld];
                cn = cn + r[x + 1 + ld];
                cn = cn * 0.0625;
                const double val = (a + nb) + cn;
                f[q] = val;
                if (I >= oi[k + 1].lo && I <= oi[k + 1].hi && J >= oj[k + 1].lo && J <= oj[k + 1].hi)
                    go[(size_t)I + (size_t)J * mc] = val;
            }
        }
        if (k + 1 < NL) __syncthreads();  // the residual is spent: its array serves the next level
        MT(0, 10 * k + 5);
    }
}

template <int NL, int Q>
__global__ __launch_bounds__(1024) void k_mid_up(const MidArgs A)
{
    extern __shared__ __align__(16) double lds[];
    double *C = lds;
    double *U = lds + A.stride;
    double *W = lds + 2 * A.stride;
    MT(1, 0);
    Span gi[NL], gj[NL], cli, clj, owni, ownj;
    up_spans<NL>(A, (int)blockIdx.x, A.tile_i, A.nti, gi, cli, owni);
    up_spans<NL>(A, (int)blockIdx.y, A.tile_j, A.ntj, gj, clj, ownj);

    // Everything the workgroup reads from memory is requested before anything is used, so
    // that the levels' load latencies overlap; the right-hand sides stay in registers.
    double pf[NL][Q], pc[Q];
#pragma unroll
    for (int k = 0; k < NL; k++) {
        const Lay L = lay_of(gi[k], gj[k]);
        const Pts P = pts_of(gi[k], gj[k], L);
        const double *__restrict__ g = A.f[k];
        const size_t mk = (size_t)A.m[k];
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const bool ok = P.ok(q);
            const int ic = ok ? P.i : gi[k].lo, jc = ok ? P.col(q) : gj[k].lo;
            pf[k][q] = g[(size_t)ic + (size_t)jc * mk];
        }
    }
    // level NL in the layout of its interior points: boundary values land in the border
    const Span cii{imax(1, cli.lo), imin(A.m[NL] - 2, cli.hi)};
    const Span cjj{imax(1, clj.lo), imin(A.m[NL] - 2, clj.hi)};
    Lay LC = lay_of(cii, cjj);
    {
        const Pts P = pts_of(cli, clj, LC);
        const double *__restrict__ g = A.cv;
        const size_t mk = (size_t)A.m[NL];
#pragma unroll
        for (int q = 0; q < Q; q++) {
            const bool ok = P.ok(q);
            const int ic = ok ? P.i : cli.lo, jc = ok ? P.col(q) : clj.lo;
            pc[q] = g[(size_t)ic + (size_t)jc * mk];
        }
#pragma unroll
        for (int q = 0; q < Q; q++)
            if (P.ok(q)) C[P.x + q * P.xs] = pc[q];
    }

    MT(1, 1);
    double u[Q];
#pragma unroll
    for (int k = NL - 1; k >= 0; k--) {
        MT(1, 10 * k + 2);
        const Span si = gi[k], sj = gj[k];
        const Lay L = lay_of(si, sj);
        const Pts P = pts_of(si, sj, L);
        const SweepCoef c{A.h2[k], A.invh2[k], A.omega, A.nomega};
        zero_border(U, W, sj, L);
        if (k == NL - 1) __syncthreads();  // the coarse values are in place
        // first sweep from zero (pointwise); with Ndown <= 1 the correction follows at once
#pragma unroll
        for (int q = 0; q < Q; q++) {
            if (P.ok(q)) {
                double v = A.ndown > 0 ? jacobi_from_zero(pf[k][q], c) : 0.0;
                if (A.ndown <= 1) v = v + prolong_at(C, LC, P.i, P.col(q));  // main.cpp:341
                u[q] = v;
                U[P.x + q * P.xs] = v;
            }
        }
        __syncthreads();
        MT(1, 10 * k + 3);
        for (int s = 2; s <= A.ndown; s++) {
            sweep_pts(U, W, P, L, c, pf[k], u, s == A.ndown ? C : nullptr, LC);
            double *t = U;
            U = W;
            W = t;
        }
        MT(1, 10 * k + 4);
        for (int s = 0; s < A.nup; s++) {  // main.cpp:343
            sweep_pts(U, W, P, L, c, pf[k], u, nullptr, LC);
            double *t = U;
            U = W;
            W = t;
        }
        // this level's iterate is the next one's coarse level; the spent coarse array joins the pair
        MT(1, 10 * k + 5);
        double *t = C;
        C = U;
        U = t;
        LC = L;
        if (k == 0) {  // the thread's points of level 0 that lie in the tile
            double *__restrict__ g = A.v;
            const size_t m0 = (size_t)A.m[0];
#pragma unroll
            for (int q = 0; q < Q; q++) {
                const int j = P.col(q);
                if (P.ok(q) && P.i >= owni.lo && P.i <= owni.hi && j >= ownj.lo && j <= ownj.hi)
                    g[(size_t)P.i + (size_t)j * m0] = u[q];
            }
            MT(1, 60);
        }
    }
}

// ---- host side: tile sizes ------------------------------------------------------------

int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e && *e ? atoi(e) : dflt;
}

// Threads per workgroup and points per thread.  The kernels are bound by the latency of
// their chain of phases, not by throughput: workgroups should all be resident at once.
struct Config {
    int threads, q;
};
Config mid_config(bool down, const MidArgs &A)
{
    // measured on MI355X (tools/mid_exp.sh): 1024 threads, except 512 for the up legs of a group
    Config c{down || A.nl == 1 ? 1024 : 512, 2};
    c.threads = env_int(down ? "MGX_MID_DOWN_THREADS" : "MGX_MID_UP_THREADS", c.threads);
    c.q = env_int(down ? "MGX_MID_DOWN_Q" : "MGX_MID_UP_Q", c.q);
    c.threads = imin(imax((c.threads / 64) * 64, 64), 1024);
    c.q = c.q <= 1 ? 1 : (c.q <= 2 ? 2 : 4);
    return c;
}

size_t mid_lds_budget()
{
    return (size_t)imin(imax(env_int("MGX_MID_LDS_KB", 78), 8), 160) * 1024;
}

int lanes_log2(int wx) { return wx <= 16 ? 4 : (wx <= 32 ? 5 : 6); }

struct Shape {
    bool ok = false;
    int threads = 0, q = 0;
    int tile_i = 0, tile_j = 0, nti = 0, ntj = 0, stride = 0;
    size_t bytes = 0;
};

// widest region of every level over the tiles of one axis
template <int NL>
void down_widths(const MidArgs &A, int tile, int (&wmax)[NL + 1])
{
    const int nt = (A.m[NL] - 2 + tile - 1) / tile;
    for (int k = 0; k <= NL; k++) wmax[k] = 0;
    for (int t = 0; t < nt; t++) {
        Span need[NL + 1], own[NL + 1];
        down_spans<NL>(A, t, tile, nt, need, own);
        for (int k = 0; k <= NL; k++) wmax[k] = imax(wmax[k], width(need[k]));
    }
}

template <int NL>
Shape down_try(const MidArgs &A, const Config &cf, int ti, int tj)
{
    Shape S;
    int wi[NL + 1], wj[NL + 1];
    down_widths<NL>(A, ti, wi);
    down_widths<NL>(A, tj, wj);
    int stride = 0;
    bool fits = true;
    S.threads = cf.threads;
    S.q = cf.q;
    for (int k = 0; k <= NL; k++) {
        // one pass of lanes along i, at most q passes along j (the registers of a thread)
        if (wi[k] > 64) fits = false;
        const int dj = S.threads >> lanes_log2(wi[k]);
        if ((wj[k] + dj - 1) / dj > S.q) fits = false;
        if (k < NL) stride = imax(stride, (wi[k] + 2) * (wj[k] + 2));
    }
    S.tile_i = ti;
    S.tile_j = tj;
    S.nti = (A.m[NL] - 2 + ti - 1) / ti;
    S.ntj = (A.m[NL] - 2 + tj - 1) / tj;
    S.stride = (stride + 1) & ~1;
    S.bytes = 2 * (size_t)S.stride * sizeof(double);
    S.ok = fits && S.bytes <= mid_lds_budget() && S.nti <= 65535 && S.ntj <= 65535;
    return S;
}

template <int NL>
Shape down_shape(const MidArgs &A, const Config &cf)
{
    const int fi = env_int("MGX_MID_DOWN_TI", 0), fj = env_int("MGX_MID_DOWN_TJ", 0);
    if (fi > 0 && fj > 0) return down_try<NL>(A, cf, fi, fj);
    const int n = A.m[NL] - 2;
    // lanes run along i: the widest tile whose level-0 region still fits 64 lanes
    int ti = 1;
    for (int t = imin(n, 32); t >= 1; t--) {
        int wi[NL + 1];
        down_widths<NL>(A, t, wi);
        if (wi[0] <= 64) {
            ti = t;
            break;
        }
    }
    Shape best;
    for (int tj = imin(n, ti); tj >= 1; tj--) {
        Shape S = down_try<NL>(A, cf, ti, tj);
        if (!S.ok) continue;
        best = S;
        // enough workgroups for every CU, as long as the tiles stay reasonably square
        if ((long)S.nti * S.ntj >= 256 || tj <= 3) break;
    }
    return best;
}

template <int NL>
void up_widths(const MidArgs &A, int tile, int (&wmax)[NL + 1])
{
    const int nt = (A.m[0] - 2 + tile - 1) / tile;
    for (int k = 0; k <= NL; k++) wmax[k] = 0;
    for (int t = 0; t < nt; t++) {
        Span G[NL], cload, own;
        up_spans<NL>(A, t, tile, nt, G, cload, own);
        for (int k = 0; k < NL; k++) wmax[k] = imax(wmax[k], width(G[k]));
        wmax[NL] = imax(wmax[NL], width(cload));
    }
}

template <int NL>
Shape up_try(const MidArgs &A, const Config &cf, int ti, int tj)
{
    Shape S;
    int wi[NL + 1], wj[NL + 1];
    up_widths<NL>(A, ti, wi);
    up_widths<NL>(A, tj, wj);
    bool fits = !(NL == 5 && cf.q == 4);  // that instance does not fit the registers of 1024 threads
    int stride = 0;
    S.threads = cf.threads;
    S.q = cf.q;
    for (int k = 0; k <= NL; k++) {
        // one pass of lanes along i, at most q passes along j (the registers of a thread)
        if (wi[k] > 64) fits = false;
        const int dj = S.threads >> lanes_log2(wi[k]);
        if ((wj[k] + dj - 1) / dj > S.q) fits = false;
        stride = imax(stride, (wi[k] + 2) * (wj[k] + 2));
    }
    S.tile_i = ti;
    S.tile_j = tj;
    S.nti = (A.m[0] - 2 + ti - 1) / ti;
    S.ntj = (A.m[0] - 2 + tj - 1) / tj;
    S.stride = (stride + 1) & ~1;
    S.bytes = 3 * (size_t)S.stride * sizeof(double);
    S.ok = fits && S.bytes <= mid_lds_budget() && S.nti <= 65535 && S.ntj <= 65535;
    return S;
}

template <int NL>
Shape up_shape(const MidArgs &A, const Config &cf)
{
    const int fi = env_int("MGX_MID_UP_TI", 0), fj = env_int("MGX_MID_UP_TJ", 0);
    if (fi > 0 && fj > 0) return up_try<NL>(A, cf, fi, fj);
    const int n = A.m[0] - 2;
    int ti = 1;
    for (int t = imin(n, 64); t >= 1; t--) {
        int wi[NL + 1];
        up_widths<NL>(A, t, wi);
        if (wi[0] <= 64) {
            ti = t;
            break;
        }
    }
    Shape best;
    for (int tj = imin(n, ti); tj >= 1; tj--) {
        Shape S = up_try<NL>(A, cf, ti, tj);
        if (!S.ok) continue;
        best = S;
        if ((long)S.nti * S.ntj >= 512 || tj <= 8) break;
    }
    return best;
}

// shapes are a function of the level sizes and the sweep counts: looked up per launch
// (plus the tile sizes forced through the environment, which experiments change between plans)
// down/up, levels, rows of level 0, Ndown, Nup, forced tile, LDS budget, threads and points per thread
using Key = std::tuple<int, int, int, int, int, int, int, int, int>;
std::mutex g_mu;
std::map<Key, Shape> g_shapes;

template <int NL>
Shape shape_of(bool down, const MidArgs &A)
{
    const Config cf = mid_config(down, A);
    const Key key{down ? 0 : 1, NL, A.m[0], A.ndown, A.nup, env_int(down ? "MGX_MID_DOWN_TI" : "MGX_MID_UP_TI", 0),
                  env_int(down ? "MGX_MID_DOWN_TJ" : "MGX_MID_UP_TJ", 0), (int)(mid_lds_budget() >> 10), cf.threads * 8 + cf.q};
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_shapes.find(key);
    if (it != g_shapes.end()) return it->second;
    const Shape S = down ? down_shape<NL>(A, cf) : up_shape<NL>(A, cf);
    g_shapes.emplace(key, S);
    return S;
}

Shape shape_any(bool down, const MidArgs &A)
{
    switch (A.nl) {
    case 1: return shape_of<1>(down, A);
    case 2: return shape_of<2>(down, A);
    case 3: return shape_of<3>(down, A);
    case 4: return shape_of<4>(down, A);
    case 5: return shape_of<5>(down, A);
    default: return Shape();
    }
}

bool levels_nest(const MidArgs &A)
{
    if (A.nl < 1 || A.nl > MID_MAX_LEVELS) return false;
    for (int k = 0; k < A.nl; k++) {
        if (A.m[k] < 3 || (A.m[k] & 1) == 0) return false;
        if (A.m[k + 1] != (A.m[k] + 1) / 2) return false;
    }
    return A.m[A.nl] >= 3;
}

// a launch with more than 64 KB of dynamic LDS must be allowed once per kernel and device
hipError_t allow_lds(const void *kernel, size_t bytes)
{
    static std::mutex mu;
    static std::map<const void *, unsigned long long> done;
    if (bytes <= 64 * 1024) return hipSuccess;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = (dev >= 0 && dev < 64) ? (1ull << dev) : 0ull;
    std::lock_guard<std::mutex> lock(mu);
    unsigned long long &mask = done[kernel];
    if (mask & bit) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) mask |= bit;
    return e;
}

template <int NL, int Q>
hipError_t launch_down_nq(const MidArgs &A, const Shape &S, hipStream_t s)
{
    hipError_t e = allow_lds((const void *)k_mid_down<NL, Q>, S.bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_mid_down<NL, Q>), dim3((unsigned)S.nti, (unsigned)S.ntj), dim3((unsigned)S.threads), S.bytes, s, A);
    return hipGetLastError();
}

template <int NL, int Q>
hipError_t launch_up_nq(const MidArgs &A, const Shape &S, hipStream_t s)
{
    hipError_t e = allow_lds((const void *)k_mid_up<NL, Q>, S.bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_mid_up<NL, Q>), dim3((unsigned)S.nti, (unsigned)S.ntj), dim3((unsigned)S.threads), S.bytes, s, A);
    return hipGetLastError();
}

template <int NL>
hipError_t launch_down_n(const MidArgs &A, const Shape &S, hipStream_t s)
{
    switch (S.q) {
    case 1: return launch_down_nq<NL, 1>(A, S, s);
    case 2: return launch_down_nq<NL, 2>(A, S, s);
    case 4: return launch_down_nq<NL, 4>(A, S, s);
    }
    return hipErrorInvalidValue;
}

template <int NL>
hipError_t launch_up_n(const MidArgs &A, const Shape &S, hipStream_t s)
{
    switch (S.q) {
    case 1: return launch_up_nq<NL, 1>(A, S, s);
    case 2: return launch_up_nq<NL, 2>(A, S, s);
    case 4: return launch_up_nq<NL, 4>(A, S, s);
    }
    return hipErrorInvalidValue;
}

MidArgs with_shape(const MidArgs &A0, const Shape &S)
{
    MidArgs A = A0;
    A.tile_i = S.tile_i;
    A.tile_j = S.tile_j;
    A.nti = S.nti;
    A.ntj = S.ntj;
    A.stride = S.stride;
    return A;
}

} // namespace

bool mid_fits(bool down, const MidArgs &A)
{
    if (!levels_nest(A) || A.ndown < 0 || A.nup < 0) return false;
    return shape_any(down, A).ok;
}

hipError_t launch_mid_down(const MidArgs &A0, hipStream_t s)
{
    if (!levels_nest(A0)) return hipErrorInvalidValue;
    const Shape S = shape_any(true, A0);
    if (!S.ok) return hipErrorInvalidValue;
    const MidArgs A = with_shape(A0, S);
    switch (A.nl) {
    case 1: return launch_down_n<1>(A, S, s);
    case 2: return launch_down_n<2>(A, S, s);
    case 3: return launch_down_n<3>(A, S, s);
    case 4: return launch_down_n<4>(A, S, s);
    case 5: return launch_down_n<5>(A, S, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_mid_up(const MidArgs &A0, hipStream_t s)
{
    if (!levels_nest(A0)) return hipErrorInvalidValue;
    const Shape S = shape_any(false, A0);
    if (!S.ok) return hipErrorInvalidValue;
    const MidArgs A = with_shape(A0, S);
    switch (A.nl) {
    case 1: return launch_up_n<1>(A, S, s);
    case 2: return launch_up_n<2>(A, S, s);
    case 3: return launch_up_n<3>(A, S, s);
    case 4: return launch_up_n<4>(A, S, s);
    case 5: return launch_up_n<5>(A, S, s);
    }
    return hipErrorInvalidValue;
}

} // namespace mgx

// ---- host-only introspection of the tiling (tests, tuning tools) ------------------------

namespace {

mgx::MidArgs geometry_args(int nl, int m0, int Ndown, int Nup)
{
    mgx::MidArgs A;
    A.nl = nl;
    int m = m0;
    for (int k = 0; k <= nl && k <= mgx::MID_MAX_LEVELS; k++) {
        A.m[k] = m;
        m = (m + 1) / 2;
    }
    A.ndown = Ndown;
    A.nup = Nup;
    return A;
}

template <int NL>
void spans_out(bool down, const mgx::MidArgs &A, int tile, int t, int *lo, int *hi)
{
    using namespace mgx;
    if (down) {
        const int nt = (A.m[NL] - 2 + tile - 1) / tile;
        Span need[NL + 1], own[NL + 1];
        down_spans<NL>(A, t, tile, nt, need, own);
        for (int k = 0; k <= NL; k++) {
            lo[k] = need[k].lo;
            hi[k] = need[k].hi;
            lo[NL + 1 + k] = own[k].lo;
            hi[NL + 1 + k] = own[k].hi;
        }
    } else {
        const int nt = (A.m[0] - 2 + tile - 1) / tile;
        Span G[NL], cload, own;
        up_spans<NL>(A, t, tile, nt, G, cload, own);
        for (int k = 0; k < NL; k++) {
            lo[k] = G[k].lo;
            hi[k] = G[k].hi;
        }
        lo[NL] = cload.lo;
        hi[NL] = cload.hi;
        lo[NL + 1] = own.lo;
        hi[NL + 1] = own.hi;
    }
}

} // namespace

#ifdef MGX_MID_TRACE
// m >= 0: trace the launches whose level 0 has m rows from now on; out: the 2 x 64 marks so far
extern "C" int mgx_debug_mid_trace(int m, unsigned long long *out)
{
    hipError_t e = hipSuccess;
    if (m >= 0) {
        static const unsigned long long zeros[2 * 64] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(mgx::g_mid_trace), zeros, sizeof(zeros));
        if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(mgx::g_mid_trace_m), &m, sizeof(int));
    }
    if (e == hipSuccess && out) e = hipMemcpyFromSymbol(out, HIP_SYMBOL(mgx::g_mid_trace), sizeof(unsigned long long) * 2 * 64);
    return (int)e;
}
#endif

extern "C" {

int mgx_mid_shape(int down, int nl, int m0, int Ndown, int Nup, mgx_mid_info *out)
{
    if (!out) return mgx::api_fail(MGX_ERR_ARG, "mgx_mid_shape: null result");
    *out = mgx_mid_info();
    if (nl < 1 || nl > mgx::MID_MAX_LEVELS || Ndown < 0 || Nup < 0) return mgx::api_fail(MGX_ERR_ARG, "mgx_mid_shape: bad argument");
    const mgx::MidArgs A = geometry_args(nl, m0, Ndown, Nup);
    if (!mgx::levels_nest(A)) return MGX_OK;  // out->ok stays 0
    const mgx::Shape S = mgx::shape_any(down != 0, A);
    out->ok = S.ok ? 1 : 0;
    out->threads = S.threads;
    out->points_per_thread = S.q;
    out->tile_i = S.tile_i;
    out->tile_j = S.tile_j;
    out->ntiles_i = S.nti;
    out->ntiles_j = S.ntj;
    out->lds_bytes = S.bytes;
    return MGX_OK;
}

int mgx_mid_spans(int down, int nl, int m0, int Ndown, int Nup, int tile, int t, int *lo, int *hi)
{
    if (!lo || !hi || nl < 1 || nl > mgx::MID_MAX_LEVELS || tile < 1 || Ndown < 0 || Nup < 0)
        return mgx::api_fail(MGX_ERR_ARG, "mgx_mid_spans: bad argument");
    const mgx::MidArgs A = geometry_args(nl, m0, Ndown, Nup);
    if (!mgx::levels_nest(A)) return mgx::api_fail(MGX_ERR_HIERARCHY, "mgx_mid_spans: %d levels below %d rows do not nest", nl, m0);
    const int n = (down ? A.m[nl] : A.m[0]) - 2;
    if (t < 0 || t >= (n + tile - 1) / tile) return mgx::api_fail(MGX_ERR_ARG, "mgx_mid_spans: no tile %d", t);
    switch (nl) {
    case 1: spans_out<1>(down != 0, A, tile, t, lo, hi); break;
    case 2: spans_out<2>(down != 0, A, tile, t, lo, hi); break;
    case 3: spans_out<3>(down != 0, A, tile, t, lo, hi); break;
    case 4: spans_out<4>(down != 0, A, tile, t, lo, hi); break;
    case 5: spans_out<5>(down != 0, A, tile, t, lo, hi); break;
    }
    return MGX_OK;
}

} // extern "C"

