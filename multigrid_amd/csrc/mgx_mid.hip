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
// A workgroup owns a rectangular tile of the OUTPUT level and computes, in LDS,
// every point of every level that its tile depends on (overlapping tiles
// recompute their common points; Jacobi makes the copies bit-identical).  Going
// down the dependence region roughly doubles per level (a tile of T coarse
// points needs 2T + 2(Ndown+1) fine points), so down launches take two or three
// levels; going up it converges (a tile of T fine points needs T/2 + ~4 coarse
// points), so one up launch can take every level of the group.
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
#include "mgx_internal.h"

#include <map>
#include <mutex>
#include <tuple>

#include <cstdlib>

namespace mgx {

namespace {

constexpr int MID_Q = 4;  // columns per thread the up kernel prefetches per level

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

// Which points of a region a thread handles: lanes run along i (16, 32 or 64 of them per
// column, whatever covers the region's width), the rest of the thread index along j.
struct Walk {
    int i, j, di, dj;
};
__device__ __forceinline__ Walk walk_of(const Span &si, const Span &sj)
{
    const int wx = width(si);
    const int lw = wx <= 16 ? 4 : (wx <= 32 ? 5 : 6);
    const int t = (int)threadIdx.x;
    return Walk{si.lo + (t & ((1 << lw) - 1)), sj.lo + (t >> lw), 1 << lw, (int)blockDim.x >> lw};
}
#define MID_FOR(w_, si_, sj_, i_, j_)                              \
    for (int j_ = (w_).j; j_ <= (sj_).hi; j_ += (w_).dj)          \
        for (int i_ = (w_).i; i_ <= (si_).hi; i_ += (w_).di)

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

__device__ __forceinline__ double jacobi_at(const double *__restrict__ U, const double *__restrict__ F, int x, int ld,
                                            const SweepCoef &c)
{
    double s = U[x - 1] + U[x + 1];  // main.cpp:194-195
    s = s + U[x - ld];
    s = s + U[x + ld];
    s = s - F[x] * c.h2;
    s = s * 0.25;
    s = s * c.omega;
    return U[x] * c.nomega + s;
}

// the same sweep applied to an iterate that is zero everywhere
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

__device__ __forceinline__ void mid_sweep(const double *__restrict__ F, const double *__restrict__ U,
                                          double *__restrict__ O, const Span &si, const Span &sj, const Lay &L,
                                          const Walk &w, const SweepCoef &c)
{
    MID_FOR(w, si, sj, i, j)
    {
        const int x = at(L, i, j);
        O[x] = jacobi_at(U, F, x, L.ld, c);
    }
    __syncthreads();
}

// interpolated coarse value at fine point (i, j); C holds the coarse level in layout LC
__device__ __forceinline__ double prolong_at(const double *__restrict__ C, const Lay &LC, int i, int j)
{
    const int y = at(LC, i >> 1, j >> 1);
    const bool io = i & 1, jo = j & 1;
    double p;  // main.cpp:234-246
    if (!io && !jo) {
        p = C[y];
    } else if (!io && jo) {
        p = 0.5 * (C[y] + C[y + LC.ld]);
    } else if (io && !jo) {
        p = 0.5 * (C[y] + C[y + 1]);
    } else {
        double s = C[y] + C[y + 1];
        s = s + C[y + LC.ld];
        s = s + C[y + LC.ld + 1];
        p = 0.25 * s;
    }
    return p;
}

template <int NL>
__global__ __launch_bounds__(1024) void k_mid_down(const MidArgs A)
{
    extern __shared__ __align__(16) double lds[];
    double *F = lds;
    double *U = lds + A.stride;
    double *W = lds + 2 * A.stride;
    Span ni[NL + 1], oi[NL + 1], nj[NL + 1], oj[NL + 1];
    down_spans<NL>(A, (int)blockIdx.x, A.tile_i, A.nti, ni, oi);
    down_spans<NL>(A, (int)blockIdx.y, A.tile_j, A.ntj, nj, oj);
#pragma unroll
    for (int k = 0; k < NL; k++) {
        const Span si = ni[k], sj = nj[k];
        const Lay L = lay_of(si, sj);
        const Walk w = walk_of(si, sj);
        const SweepCoef c{A.h2[k], A.invh2[k], A.omega, A.nomega};
        zero_border(U, W, sj, L);
        // the first sweep from zero needs f at the point only (main.cpp:321)
        if (k == 0) {
            const double *__restrict__ g = A.f[0];
            const size_t m0 = (size_t)A.m[0];
            MID_FOR(w, si, sj, i, j)
            {
                const int x = at(L, i, j);
                const double fv = g[(size_t)i + (size_t)j * m0];
                F[x] = fv;
                U[x] = A.ndown > 0 ? jacobi_from_zero(fv, c) : 0.0;
            }
        } else {
            MID_FOR(w, si, sj, i, j)
            {
                const int x = at(L, i, j);
                U[x] = A.ndown > 0 ? jacobi_from_zero(F[x], c) : 0.0;
            }
        }
        __syncthreads();
        for (int s = 1; s < A.ndown; s++) {
            mid_sweep(F, U, W, si, sj, L, w, c);
            double *t = U;
            U = W;
            W = t;
        }
        MID_FOR(w, si, sj, i, j)  // main.cpp:325
        {
            const int x = at(L, i, j);
            double s = U[x - 1] + U[x + 1];  // main.cpp:277
            s = s + U[x - L.ld];
            s = s + U[x + L.ld];
            s = s - U[x] * 4.0;
            s = s * c.invh2;
            W[x] = F[x] - s;
        }
        __syncthreads();
        // main.cpp:327; the next level's f takes the place of this level's, which is spent
        const Span ci = ni[k + 1], cj = nj[k + 1];
        const Lay LC = lay_of(ci, cj);
        const Walk wc = walk_of(ci, cj);
        double *__restrict__ go = A.fo[k + 1];
        const size_t mc = (size_t)A.m[k + 1];
        const double *__restrict__ r = W;
        const int ld = L.ld;
        MID_FOR(wc, ci, cj, I, J)
        {
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
            if (k + 1 < NL) F[at(LC, I, J)] = val;
            if (I >= oi[k + 1].lo && I <= oi[k + 1].hi && J >= oj[k + 1].lo && J <= oj[k + 1].hi)
                go[(size_t)I + (size_t)J * mc] = val;
        }
        __syncthreads();
    }
}

template <int NL>
__global__ __launch_bounds__(1024) void k_mid_up(const MidArgs A)
{
    extern __shared__ __align__(16) double lds[];
    double *C = lds;
    double *U = lds + A.stride;
    double *W = lds + 2 * A.stride;
    Span gi[NL], gj[NL], cli, clj, owni, ownj;
    up_spans<NL>(A, (int)blockIdx.x, A.tile_i, A.nti, gi, cli, owni);
    up_spans<NL>(A, (int)blockIdx.y, A.tile_j, A.ntj, gj, clj, ownj);

    // Everything the workgroup reads from memory is requested before anything is used, so
    // that the levels' load latencies overlap: each thread holds at most MID_Q values per
    // level (the launcher sizes the tiles for that).
    double pf[NL][MID_Q], pc[MID_Q];
#pragma unroll
    for (int k = 0; k < NL; k++) {
        const Walk w = walk_of(gi[k], gj[k]);
        const double *__restrict__ g = A.f[k];
        const size_t mk = (size_t)A.m[k];
#pragma unroll
        for (int q = 0; q < MID_Q; q++) {
            const int j = w.j + q * w.dj;
            const bool ok = w.i <= gi[k].hi && j <= gj[k].hi;
            const int ic = ok ? w.i : gi[k].lo, jc = ok ? j : gj[k].lo;
            pf[k][q] = g[(size_t)ic + (size_t)jc * mk];
        }
    }
    {
        const Walk w = walk_of(cli, clj);
        const double *__restrict__ g = A.cv;
        const size_t mk = (size_t)A.m[NL];
#pragma unroll
        for (int q = 0; q < MID_Q; q++) {
            const int j = w.j + q * w.dj;
            const bool ok = w.i <= cli.hi && j <= clj.hi;
            const int ic = ok ? w.i : cli.lo, jc = ok ? j : clj.lo;
            pc[q] = g[(size_t)ic + (size_t)jc * mk];
        }
    }
#pragma unroll
    for (int k = 0; k < NL; k++) {
        const Walk w = walk_of(gi[k], gj[k]);
        const Lay L = lay_of(gi[k], gj[k]);
        double *__restrict__ Fk = lds + A.foff[k];
#pragma unroll
        for (int q = 0; q < MID_Q; q++) {
            const int j = w.j + q * w.dj;
            if (w.i <= gi[k].hi && j <= gj[k].hi) Fk[at(L, w.i, j)] = pf[k][q];
        }
    }
    // level NL in the layout of its interior points: boundary values land in the border
    Lay LC;
    {
        const Span ii{imax(1, cli.lo), imin(A.m[NL] - 2, cli.hi)};
        const Span jj{imax(1, clj.lo), imin(A.m[NL] - 2, clj.hi)};
        LC = lay_of(ii, jj);
        const Walk w = walk_of(cli, clj);
#pragma unroll
        for (int q = 0; q < MID_Q; q++) {
            const int j = w.j + q * w.dj;
            if (w.i <= cli.hi && j <= clj.hi) C[at(LC, w.i, j)] = pc[q];
        }
    }

#pragma unroll
    for (int k = NL - 1; k >= 0; k--) {
        const Span si = gi[k], sj = gj[k];
        const Lay L = lay_of(si, sj);
        const Walk w = walk_of(si, sj);
        const SweepCoef c{A.h2[k], A.invh2[k], A.omega, A.nomega};
        const double *__restrict__ Fk = lds + A.foff[k];
        zero_border(U, W, sj, L);
        if (k == NL - 1) __syncthreads();  // the coarse values are in place
        // first sweep from zero (pointwise); with Ndown <= 1 the correction follows at once
        MID_FOR(w, si, sj, i, j)
        {
            const int x = at(L, i, j);
            double v = A.ndown > 0 ? jacobi_from_zero(Fk[x], c) : 0.0;
            if (A.ndown <= 1) v = v + prolong_at(C, LC, i, j);  // main.cpp:341
            U[x] = v;
        }
        __syncthreads();
        for (int s = 2; s <= A.ndown; s++) {
            MID_FOR(w, si, sj, i, j)
            {
                const int x = at(L, i, j);
                double v = jacobi_at(U, Fk, x, L.ld, c);
                if (s == A.ndown) v = v + prolong_at(C, LC, i, j);  // main.cpp:341
                W[x] = v;
            }
            __syncthreads();
            double *t = U;
            U = W;
            W = t;
        }
        for (int s = 0; s < A.nup; s++) {  // main.cpp:343
            mid_sweep(Fk, U, W, si, sj, L, w, c);
            double *t = U;
            U = W;
            W = t;
        }
        // this level's iterate is the next one's coarse level; the spent coarse array joins the pair
        double *t = C;
        C = U;
        U = t;
        LC = L;
    }
    {
        const Walk w = walk_of(owni, ownj);
        double *__restrict__ g = A.v;
        const size_t m0 = (size_t)A.m[0];
        MID_FOR(w, owni, ownj, i, j) g[(size_t)i + (size_t)j * m0] = C[at(LC, i, j)];
    }
}

// ---- host side: tile sizes ------------------------------------------------------------

int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e && *e ? atoi(e) : dflt;
}

// threads per workgroup: the down kernel's workgroups are few and big, the up kernel's are
// many and should all be resident at once (two or three per CU)
int mid_threads(bool down)
{
    int t = env_int(down ? "MGX_MID_DOWN_THREADS" : "MGX_MID_UP_THREADS", down ? 1024 : 512);
    t = (t / 64) * 64;
    return t < 64 ? 64 : (t > 1024 ? 1024 : t);
}

size_t mid_lds_budget()
{
    return (size_t)imin(imax(env_int("MGX_MID_LDS_KB", 78), 8), 160) * 1024;
}

int lanes_log2(int wx) { return wx <= 16 ? 4 : (wx <= 32 ? 5 : 6); }

struct Shape {
    bool ok = false;
    int threads = 0;
    int tile_i = 0, tile_j = 0, nti = 0, ntj = 0, stride = 0;
    int foff[MID_MAX_LEVELS] = {0};
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
Shape down_try(const MidArgs &A, int ti, int tj)
{
    Shape S;
    int wi[NL + 1], wj[NL + 1];
    down_widths<NL>(A, ti, wi);
    down_widths<NL>(A, tj, wj);
    int stride = 0;
    for (int k = 0; k < NL; k++) stride = imax(stride, (wi[k] + 2) * (wj[k] + 2));
    S.tile_i = ti;
    S.tile_j = tj;
    S.nti = (A.m[NL] - 2 + ti - 1) / ti;
    S.ntj = (A.m[NL] - 2 + tj - 1) / tj;
    S.stride = (stride + 1) & ~1;
    S.bytes = 3 * (size_t)S.stride * sizeof(double);
    S.threads = mid_threads(true);
    S.ok = S.bytes <= mid_lds_budget() && S.nti <= 65535 && S.ntj <= 65535;
    return S;
}

template <int NL>
Shape down_shape(const MidArgs &A)
{
    const int fi = env_int("MGX_MID_DOWN_TI", 0), fj = env_int("MGX_MID_DOWN_TJ", 0);
    if (fi > 0 && fj > 0) return down_try<NL>(A, fi, fj);
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
        Shape S = down_try<NL>(A, ti, tj);
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
Shape up_try(const MidArgs &A, int ti, int tj)
{
    Shape S;
    int wi[NL + 1], wj[NL + 1];
    up_widths<NL>(A, ti, wi);
    up_widths<NL>(A, tj, wj);
    bool fits = true;
    int stride = 0;
    S.threads = mid_threads(false);
    for (int k = 0; k <= NL; k++) {
        // one pass of lanes along i, at most MID_Q passes along j (the prefetch registers)
        if (wi[k] > 64) fits = false;
        const int dj = S.threads >> lanes_log2(wi[k]);
        if ((wj[k] + dj - 1) / dj > MID_Q) fits = false;
        stride = imax(stride, (wi[k] + 2) * (wj[k] + 2));
    }
    S.tile_i = ti;
    S.tile_j = tj;
    S.nti = (A.m[0] - 2 + ti - 1) / ti;
    S.ntj = (A.m[0] - 2 + tj - 1) / tj;
    S.stride = (stride + 1) & ~1;
    int off = 3 * S.stride;
    for (int k = 0; k < NL; k++) {
        S.foff[k] = off;
        off += ((wi[k] + 2) * (wj[k] + 2) + 1) & ~1;
    }
    S.bytes = (size_t)off * sizeof(double);
    S.ok = fits && S.bytes <= mid_lds_budget() && S.nti <= 65535 && S.ntj <= 65535;
    return S;
}

template <int NL>
Shape up_shape(const MidArgs &A)
{
    const int fi = env_int("MGX_MID_UP_TI", 0), fj = env_int("MGX_MID_UP_TJ", 0);
    if (fi > 0 && fj > 0) return up_try<NL>(A, fi, fj);
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
        Shape S = up_try<NL>(A, ti, tj);
        if (!S.ok) continue;
        best = S;
        if ((long)S.nti * S.ntj >= 512 || tj <= 8) break;
    }
    return best;
}

// shapes are a function of the level sizes and the sweep counts: looked up per launch
// (plus the tile sizes forced through the environment, which experiments change between plans)
// down/up, levels, rows of level 0, Ndown, Nup, forced tile, LDS budget and threads
using Key = std::tuple<int, int, int, int, int, int, int, int>;
std::mutex g_mu;
std::map<Key, Shape> g_shapes;

template <int NL>
Shape shape_of(bool down, const MidArgs &A)
{
    const Key key{down ? 0 : 1, NL, A.m[0], A.ndown, A.nup, env_int(down ? "MGX_MID_DOWN_TI" : "MGX_MID_UP_TI", 0),
                  env_int(down ? "MGX_MID_DOWN_TJ" : "MGX_MID_UP_TJ", 0), (int)(mid_lds_budget() >> 10) * 2048 + mid_threads(down)};
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_shapes.find(key);
    if (it != g_shapes.end()) return it->second;
    const Shape S = down ? down_shape<NL>(A) : up_shape<NL>(A);
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
hipError_t allow_lds(const void *kernel, int slot, size_t bytes)
{
    static std::mutex mu;
    static unsigned long long done[2 * (MID_MAX_LEVELS + 1)] = {0};
    if (bytes <= 64 * 1024) return hipSuccess;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = (dev >= 0 && dev < 64) ? (1ull << dev) : 0ull;
    std::lock_guard<std::mutex> lock(mu);
    if (done[slot] & bit) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) done[slot] |= bit;
    return e;
}

template <int NL>
hipError_t launch_down_n(const MidArgs &A, const Shape &S, hipStream_t s)
{
    hipError_t e = allow_lds((const void *)k_mid_down<NL>, NL, S.bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_mid_down<NL>), dim3((unsigned)S.nti, (unsigned)S.ntj), dim3((unsigned)S.threads), S.bytes, s, A);
    return hipGetLastError();
}

template <int NL>
hipError_t launch_up_n(const MidArgs &A, const Shape &S, hipStream_t s)
{
    hipError_t e = allow_lds((const void *)k_mid_up<NL>, MID_MAX_LEVELS + 1 + NL, S.bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_mid_up<NL>), dim3((unsigned)S.nti, (unsigned)S.ntj), dim3((unsigned)S.threads), S.bytes, s, A);
    return hipGetLastError();
}

MidArgs with_shape(const MidArgs &A0, const Shape &S)
{
    MidArgs A = A0;
    A.tile_i = S.tile_i;
    A.tile_j = S.tile_j;
    A.nti = S.nti;
    A.ntj = S.ntj;
    A.stride = S.stride;
    for (int k = 0; k < MID_MAX_LEVELS; k++) A.foff[k] = S.foff[k];
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
