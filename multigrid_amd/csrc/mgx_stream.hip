// mgx_stream.hip -- fused streaming kernels for gfx950.
//
// One kernel template covers every fused step of the V-cycle.  A launch is a
// pipeline over the NS+1 "stages" of the iterate (stage 0 = the input, stage s
// = after s damped-Jacobi sweeps, reference relax(), main.cpp:171-199):
//
//   k_stream<NS, PA, NM, RES, ZERO_IN, STORE>
//     NS       number of sweeps applied in registers
//     PA       >= 0: the coarse-grid correction P(coarse v) is added to stage PA
//              (refine + `cur += refined`, main.cpp:339-341); -1: none
//     NM       >= 0: the infinity norm of the residual of stage NM is merged into
//              a slot (calcNorm(residual), main.cpp:307,351); -1: none
//     RES      1: the residual of stage NS is restricted into the coarse f
//              (residual + coarsen, main.cpp:325-327)
//     ZERO_IN  the input iterate is all zeros and is not read (a coarse level
//              when the down sweep reaches it, main.cpp:295-299,345)
//     STORE    stage NS is written to the output iterate
//
// The legs of a cycle are instances of it:
//   down            <nu1,-1,-1,1,*,*>       relax + residual + coarsen
//   up              <nu2, 0,-1|nu2,0,0,1>   refine + add + relax (+ norm on level 0)
//   up+down         <nu2+nu1,0,nu2,1,0,1>   level 0: the up leg of cycle k and the
//                                           down leg of cycle k+1 in ONE pass
//   recomputing up  <nu1+nu2,nu1,-1,0,1,1>  levels >= 1: the down sweeps started
//                                           from zero, so they are recomputed from
//                                           f instead of being stored and re-read
//
// Mapping.  A wavefront owns a strip of 128 consecutive rows i (lane t holds
// rows r0+2t and r0+2t+1 in registers) and marches along the columns j of a
// chunk.  Column j of the input is loaded once (coalesced: the 64 lanes read
// 1 KiB of one column, 16 B per lane); stage s of column j-s is computed as
// soon as stage s-1 of columns j-s-1..j-s+1 exists, so the sweeps, the residual
// and the restriction form a software pipeline that lags the load front by
// NS+2 columns and never touches memory in between.  North/south neighbours
// (j-1, j+1) are the lane's own registers of the previous/next column;
// west/east neighbours (i-1, i+1) are the lane's other row or one cross-lane
// shift.  No LDS; the waves of a workgroup meet at an s_barrier every few columns
// only so that they stay close enough for their shared halo rows to hit in L2.
// Strips overlap by the halo the pipeline consumes (NS+2 rows below, NS+1 above),
// chunks by the same number of columns; the overlap is recomputed, not communicated.
//
// The march is unrolled by two so that the parity of the column (which decides
// the interpolation formula and when a coarse column is complete) is static,
// and every load in the loop is unconditional (out-of-range columns and rows
// are clamped to a valid address and the value is ignored): the compiler can
// then count outstanding loads exactly and keep PF columns in flight per wave
// instead of draining the queue every step.
//
// Every expression is evaluated exactly as in the stepwise kernels
// (mgx_kernels.hip), so both schedules produce identical bits.
#include "mgx_internal.h"

#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace mgx {

namespace {

struct P2 {
    double a, b;  // rows ra (even offset in the strip) and ra+1
};

template <int NS, int PA, int NM, int RES>
struct Geo {
    static constexpr bool RESID_END = (RES != 0) || (NM == NS);  // a residual of stage NS is taken
    // rows of halo the pipeline consumes below / above the rows a strip owns
    static constexpr int HL = NS + (RES ? 2 : (NM == NS ? 1 : 0));
    static constexpr int HH = NS + (RESID_END ? 1 : 0) + (PA == 0 ? 1 : 0);
    static constexpr int PADLO = (HL + 1) & ~1;            // even, so that r0 is even
#ifndef MGX_STRIP_SHRINK
#define MGX_STRIP_SHRINK 0  // experiment knob: rows a strip gives up (a larger share of halo rows)
#endif
    static constexpr int S = ((128 - PADLO - HH) & ~1) - MGX_STRIP_SHRINK;  // rows owned per strip (even)
    // columns of halo a chunk consumes before its first owned column
    static constexpr int HLC = HL;
    // last input column the march must reach, relative to the chunk end jb
    static constexpr int JLAST = NS + (RES ? 1 : (NM == NS ? 0 : -1));
};

__device__ __forceinline__ double jacobi_pt(double uW, double uE, double uS, double uN, double uC, double f,
                                            const SweepCoef &c)
{
    double s = uW + uE;  // main.cpp:194-195
    s = s + uS;
    s = s + uN;
    s = s - f * c.h2;
    s = s * 0.25;
    s = s * c.omega;
    return uC * c.nomega + s;
}

__device__ __forceinline__ double residual_pt(double uW, double uE, double uS, double uN, double uC, double f,
                                              const SweepCoef &c)
{
    double s = uW + uE;  // main.cpp:277
    s = s + uS;
    s = s + uN;
    s = s - uC * 4.0;
    s = s * c.invh2;
    return f - s;
}

// Neighbour lanes through the wavefront-wide DPP shifts of gfx9 (wave_shr:1 / wave_shl:1, two
// VALU moves per double) instead of __shfl_up/__shfl_down (two ds_bpermute through the LDS
// crossbar and a wait each).  Lane 0 / 63, which has no such neighbour, reads 0.0
// (bound_ctrl; keeping its own value instead would cost a second move per half): that lane's
// outer row is the outermost halo row of the strip, whose neighbour never reaches an owned point.
#ifndef MGX_DPP_SHIFT
#define MGX_DPP_SHIFT 1
#endif
template <int CTRL>
__device__ __forceinline__ double dpp_shift(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
#ifndef MGX_DPP_BOUND
#define MGX_DPP_BOUND 1
#endif
#if MGX_DPP_BOUND
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
#else
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
#endif
    return __hiloint2double(hi, lo);
}
#if MGX_DPP_SHIFT
__device__ __forceinline__ double from_lower_lane(double v) { return dpp_shift<0x138>(v); }  // lane t-1 -> t
__device__ __forceinline__ double from_upper_lane(double v) { return dpp_shift<0x130>(v); }  // lane t+1 -> t
#else
__device__ __forceinline__ double from_lower_lane(double v) { return __shfl_up(v, 1, 64); }    // lane t-1 -> t
__device__ __forceinline__ double from_upper_lane(double v) { return __shfl_down(v, 1, 64); }  // lane t+1 -> t
#endif

__device__ __forceinline__ int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

// EDGE: the strip touches row 0 or m-1 (rows are clamped and masked).  CF: the chunk marches
// over a Dirichlet column or past an end of the array (columns are tested one by one);
// without it no column of the march is special and the tests fold away.
template <int NS, int PA, int NM, int RES, bool ZERO_IN, bool STORE, bool EDGE, bool CF>
__device__ __forceinline__ void stream_wave(const StreamArgs &A, const int strip, const int chunk)
{
    using G = Geo<NS, PA, NM, RES>;
    constexpr bool PRO = PA >= 0;
    constexpr bool NORM = NM >= 0;
#ifndef MGX_PF
#define MGX_PF 2
#endif
    constexpr int PF = MGX_PF;  // columns of load-ahead
    const int lane = threadIdx.x & 63;
    const int m = A.m;
    const size_t ld = (size_t)A.ld;
    const SweepCoef c = A.c;

    // ---- rows of this lane -------------------------------------------------
    const int own_lo = strip * G::S, own_hi = own_lo + G::S;
    const int r0 = own_lo - G::PADLO;
    const int ra = r0 + 2 * lane, rb = ra + 1;
    const bool fixa = EDGE && (ra <= 0 || ra >= m - 1);  // Dirichlet or outside: never updated
    const bool fixb = EDGE && (rb <= 0 || rb >= m - 1);
    // own_lo, own_hi and ra are even, so away from the edges rows a and b are owned together
    const bool owna = ra >= own_lo && ra < own_hi && ra >= 1 && ra <= m - 2;
    const bool ownb = rb >= own_lo && rb < own_hi && rb >= 1 && rb <= m - 2;
    // rows used for addressing: clamped into the array on edge strips, so that
    // every load is unconditional (a clamped value belongs to a row that does
    // not exist and never reaches an owned point)
    const int ra_c = EDGE ? clampi(ra, 0, m - 1) : ra;
    const int rb_c = EDGE ? clampi(rb, 0, m - 1) : rb;

    // ---- columns of this chunk -----------------------------------------------
    const bool second = chunk >= A.nch1;  // a launch may produce two column ranges
    const int ja = second ? A.jlo2 + (chunk - A.nch1) * A.CJ : A.jlo + chunk * A.CJ;
    const int jb = min(ja + A.CJ, second ? A.jhi2 : A.jhi);
    int jstart = ja - G::HLC;
    if (jstart < 0) jstart = 0;
    if ((jstart + A.gf0) & 1) jstart -= 1;  // start the march on an even global column
    const int jlast = jb + G::JLAST;
    const int ncols = A.ncols;
    const int jfl = CF ? A.jfix_lo : -1, jfh = CF ? A.jfix_hi : -1;

    const double *__restrict__ pua = A.uin + ra_c;
    const double *__restrict__ pub = A.uin + rb_c;
    const double *__restrict__ pfa = A.f + ra_c;
    const double *__restrict__ pfb = A.f + rb_c;
    double *__restrict__ po = A.uout + ra;

#ifndef MGX_NT_LOAD
#define MGX_NT_LOAD 0
#endif
// The iterate is written once per launch and read again a whole launch later (2 GB at
// 16385^2): a non-temporal store keeps it from displacing the halo rows and columns that
// neighbouring waves still share through the L2 (level-0 up+down launch 1.570 -> 1.554 ms).
// Non-temporal LOADS of the streamed arrays lose (1.64 ms): they give up exactly that sharing.
#ifndef MGX_NT_STORE
#define MGX_NT_STORE 1
#endif
#ifndef MGX_NT_LOAD_F
#define MGX_NT_LOAD_F 0  // non-temporal loads of the right-hand side only: level-0 launch equal, cycle +1 % (measured)
#endif
#ifndef MGX_NT_CSTORE
#define MGX_NT_CSTORE 0  // non-temporal stores of the coarse right-hand side: level-0 launch +1.7 % (measured)
#endif
    auto load_col = [&](const double *__restrict__ pa, const double *__restrict__ pb, int j, bool is_f = false) -> P2 {
        const size_t off = (size_t)clampi(j, 0, ncols - 1) * ld;
        P2 r;
        if (MGX_NT_LOAD || (MGX_NT_LOAD_F && is_f)) {
            r.a = __builtin_nontemporal_load(pa + off);
            r.b = __builtin_nontemporal_load(pb + off);
        } else {
            r.a = pa[off];
            r.b = pb[off];  // adjacent to r.a on interior strips: one 16-byte load
        }
        return r;
    };
    auto store_to = [&](double *__restrict__ base, int j, const P2 &v) {
        if (EDGE) {
#if MGX_NT_STORE
            if (owna) __builtin_nontemporal_store(v.a, base + (size_t)j * ld);
            if (ownb) __builtin_nontemporal_store(v.b, base + (size_t)j * ld + 1);
#else
            if (owna) base[(size_t)j * ld] = v.a;
            if (ownb) base[(size_t)j * ld + 1] = v.b;
#endif
        } else if (owna) {
            double *q = base + (size_t)j * ld;
#if MGX_NT_STORE
            __builtin_nontemporal_store(v.a, q);
            __builtin_nontemporal_store(v.b, q + 1);
#else
            q[0] = v.a;
            q[1] = v.b;
#endif
        }
    };
    auto store_col = [&](int j, const P2 &v) { store_to(po, j, v); };

    // ---- prolongation state (PRO) -------------------------------------------
    // The correction is added to stage PA, whose newest column at input column
    // jin is jin-PA.  Coarse row of this lane: I = ra/2 (ra is even); its upper
    // neighbour I+1 comes from lane t+1.
    const int I = ra >> 1;
    const int I_c = EDGE ? clampi(I, 0, A.mc - 1) : I;
    const double *__restrict__ pc = PRO ? (A.cin + I_c) : nullptr;
    auto load_c = [&](int J) -> double { return pc[(size_t)clampi(J, 0, A.nccols - 1) * (size_t)A.ldc]; };
    double c0 = 0, c0u = 0, c1 = 0, c1u = 0, cnext = 0;  // cols J, J+1 (own row, row I+1) and raw col J+2
    int Jl = 0;                                          // local coarse column J of the current column pair
    if (PRO) {
        // state as it stands just before the first corrected column g0 = jstart-PA
        // (global index): an even column rotates first, an odd one does not
        const int g0 = jstart - PA + A.gf0;
        const bool g0odd = (g0 & 1) != 0;
        Jl = (g0 >> 1) - A.gc0 - (g0odd ? 0 : 1);
        if (g0odd) {
            c0 = load_c(Jl);
            c0u = from_upper_lane(c0);
        }
        c1 = load_c(Jl + 1);
        cnext = load_c(Jl + 2);
        c1u = from_upper_lane(c1);
    }

    // ---- pipeline registers ---------------------------------------------------
    P2 U[NS + 1][3];  // U[s][2] = newest column of stage s, [1], [0] the two before it
    P2 Fq[NS + 2];    // Fq[k] = f of column jin-k (k >= 1)
    P2 R[3];          // residual columns jr-2, jr-1, jr of stage NS
#pragma unroll
    for (int s = 0; s <= NS; s++)
#pragma unroll
        for (int k = 0; k < 3; k++) U[s][k] = P2{0.0, 0.0};
#pragma unroll
    for (int k = 0; k < NS + 2; k++) Fq[k] = P2{0.0, 0.0};
#pragma unroll
    for (int k = 0; k < 3; k++) R[k] = P2{0.0, 0.0};

    // load-ahead queues: Pu[k] = u of column jin+k, Pf[k] = f of column jin-1+k
    P2 Pu[PF], Pf[PF];
#pragma unroll
    for (int k = 0; k < PF; k++) {
        Pu[k] = ZERO_IN ? P2{0.0, 0.0} : load_col(pua, pub, jstart + k);
        Pf[k] = load_col(pfa, pfb, jstart - 1 + k, true);
    }

    double nacc = 0.0;  // NORM: running max |r| (boundary residuals are +0.0)

    // One column step on column jin of the input (u0, f1: the iterate of column jin and the
    // right-hand side of column jin-1); ODD is the parity of the global column index of jin.
    auto step_core = [&](const int jin, auto odd_tag, P2 u0, const P2 f1) {
        constexpr bool ODD = decltype(odd_tag)::value;

        // P(coarse) added to column jp (parity ODDP) of stage PA
        auto add_correction = [&](P2 &x, const int jp, auto oddp_tag) {
            constexpr bool ODDP = decltype(oddp_tag)::value;
            if (!ODDP) {  // entered a new coarse column
                Jl += 1;
                c0 = c1;
                c0u = c1u;
                c1 = cnext;
                c1u = from_upper_lane(c1);
                cnext = load_c(Jl + 2);
            }
            double pa, pb;
            if (!ODDP) {  // main.cpp:236, :242
                pa = c0;
                pb = 0.5 * (c0 + c0u);
            } else {  // main.cpp:239, :245
                pa = 0.5 * (c0 + c1);
                double s4 = c0 + c0u;
                s4 = s4 + c1;
                s4 = s4 + c1u;
                pb = 0.25 * s4;
            }
            const bool colfix = CF && ((jp == jfl) || (jp == jfh) || jp < 0 || jp >= ncols);
            const double na = x.a + pa;  // main.cpp:341
            const double nb = x.b + pb;
            x.a = (colfix || fixa) ? x.a : na;
            x.b = (colfix || fixb) ? x.b : nb;
        };
        // residual of stage s at column jin-s-1 (its three columns are in U[s])
        auto residual_col = [&](auto s_tag) -> P2 {
            constexpr int s = decltype(s_tag)::value;
            const P2 uS = U[s][0], uC = U[s][1], uN = U[s][2];
            const double wa = from_lower_lane(uC.b);
            const double eb = from_upper_lane(uC.a);
            P2 r;
            r.a = residual_pt(wa, uC.b, uS.a, uN.a, uC.a, Fq[s + 1].a, c);
            r.b = residual_pt(uC.a, eb, uS.b, uN.b, uC.b, Fq[s + 1].b, c);
            return r;
        };
        auto norm_merge = [&](const P2 &r, const int jr) {
            const bool colok = jr >= ja && jr < jb && !(CF && (jr == jfl || jr == jfh));
            const double xa = __builtin_fabs(r.a), xb = __builtin_fabs(r.b);
            if (colok && owna) nacc = (nacc < xa) ? xa : nacc;
            if (colok && ownb) nacc = (nacc < xb) ? xb : nacc;
        };

        // ---- stage 0 ---------------------------------------------------------
        if (PA == 0) add_correction(u0, jin, std::integral_constant<bool, ODD>{});
        U[0][0] = U[0][1];
        U[0][1] = U[0][2];
        U[0][2] = u0;
#pragma unroll
        for (int k = NS + 1; k > 1; k--) Fq[k] = Fq[k - 1];
        Fq[1] = f1;
        if (NS == 0 && PRO && STORE) {  // prolong + correct only: stage 0 is the output
            if (jin >= ja && jin < jb && !(CF && (jin == jfl || jin == jfh))) store_col(jin, u0);
        }
        if (NORM && NM == 0 && NS > 0) norm_merge(residual_col(std::integral_constant<int, 0>{}), jin - 1);

        // ---- stages 1..NS: damped Jacobi sweeps ----------------------------------
        auto sweep = [&](auto s_tag) {
            constexpr int s = decltype(s_tag)::value;
            const int jc = jin - s;
            const P2 uS = U[s - 1][0], uC = U[s - 1][1], uN = U[s - 1][2];
            const bool colfix = CF && ((jc == jfl) || (jc == jfh));  // Dirichlet column: untouched by relax
            const double wa = from_lower_lane(uC.b);         // row ra-1
            const double eb = from_upper_lane(uC.a);         // row rb+1
            const double na = jacobi_pt(wa, uC.b, uS.a, uN.a, uC.a, Fq[s].a, c);
            const double nb = jacobi_pt(uC.a, eb, uS.b, uN.b, uC.b, Fq[s].b, c);
            P2 nw;
            nw.a = (colfix || fixa) ? uC.a : na;
            nw.b = (colfix || fixb) ? uC.b : nb;
            if (PA == s) add_correction(nw, jc, std::integral_constant<bool, (ODD != ((s & 1) != 0))>{});
            U[s][0] = U[s][1];
            U[s][1] = U[s][2];
            U[s][2] = nw;
            if (s == NS && STORE && jc >= ja && jc < jb && !colfix) store_col(jc, nw);
            if (NORM && NM == s && s < NS) norm_merge(residual_col(std::integral_constant<int, s>{}), jc - 1);
        };
        if constexpr (NS >= 1) sweep(std::integral_constant<int, 1>{});
        if constexpr (NS >= 2) sweep(std::integral_constant<int, 2>{});
        if constexpr (NS >= 3) sweep(std::integral_constant<int, 3>{});
        if constexpr (NS >= 4) sweep(std::integral_constant<int, 4>{});
        static_assert(NS <= 4, "at most 4 sweeps per launch");

        // ---- residual of stage NS at column jr -----------------------------------
        if constexpr (G::RESID_END) {
            const int jr = jin - NS - 1;
            const P2 r = residual_col(std::integral_constant<int, NS>{});
            if (NORM && NM == NS) norm_merge(r, jr);
            if (RES) {
                R[0] = R[1];
                R[1] = R[2];
                R[2] = r;
                // coarse column J is centred on the even fine column 2J = jr-1,
                // complete once the odd column jr = 2J+1 has its residual
                constexpr bool JR_ODD = ((NS + 1) & 1) ? !ODD : ODD;
                if (JR_ODD) {
                    const int jcen = jr - 1;
                    if (jcen >= ja && jcen < jb && !(CF && (jcen == jfl || jcen == jfh))) {
                        const double m1 = from_lower_lane(R[1].b);  // row 2I-1, column 2J
                        const double m0 = from_lower_lane(R[0].b);  // row 2I-1, column 2J-1
                        const double t0 = R[1].a * 0.25;            // main.cpp:216-218
                        double nb = m1 + R[1].b;
                        nb = nb + R[0].a;
                        nb = nb + R[2].a;
                        nb = nb * 0.125;
                        double cn = m0 + R[0].b;
                        cn = cn + R[0].b;
                        cn = cn + R[2].b;
                        cn = cn * 0.0625;
                        const double outc = (t0 + nb) + cn;
                        const int Jc = ((jcen + A.gf0) >> 1) - A.gc0;
                        if (owna && I >= 1 && I <= A.mc - 2) {
                            double *q = A.cout + ((size_t)I + (size_t)Jc * (size_t)A.ldc);
                            if (MGX_NT_CSTORE)
                                __builtin_nontemporal_store(outc, q);
                            else
                                *q = outc;
                        }
                    }
                }
            }
        }
    };
    // The same with column jin taken from its place in the load-ahead queues (SLOT: the step
    // number modulo PF) and that place refilled at once.
    auto step = [&](const int jin, auto odd_tag, auto slot_tag) {
        constexpr int SLOT = decltype(slot_tag)::value;
        const P2 u0 = Pu[SLOT];
        const P2 f1 = Pf[SLOT];
        if (!ZERO_IN) Pu[SLOT] = load_col(pua, pub, jin + PF);
        Pf[SLOT] = load_col(pfa, pfb, jin - 1 + PF, true);
        step_core(jin, odd_tag, u0, f1);
    };

#ifndef MGX_LOCKSTEP
#define MGX_LOCKSTEP 8
#endif
#ifndef MGX_TRIP
// 1: trips of PF columns with the next trip's loads all issued at the start of the trip.  Measured
// (16385^2 cycle / level-0 launch / one rank's cycle of eight): column-by-column refill 2.100 ms /
// 1.471 ms / 0.418 ms; trips of 2: 2.14-2.25 / 1.50-1.62 / 0.418; of 4: 2.16 / 1.51 / 0.433; of 6 and 8:
// 2.23 / 1.56 / 0.444-0.456.  Longer queues do not pay, not even on the small latency-bound levels: off.
#define MGX_TRIP 0
#endif
#ifndef MGX_UNROLL6
// Measured on the level-0 launch: the six-column trip executes 20 % fewer vector instructions and
// is 7 % SLOWER (1.467 -> 1.578 ms): the moves it saves are not what the launch waits for, and its
// 170 registers and longer trip drain the load queue at a worse moment.  Off.
#define MGX_UNROLL6 0
#endif
    // Keep the workgroup's strips within a few columns of each other, so that the halo rows
    // two of them share are still in the XCD's L2 when the second wave reads them (measured on
    // the level-0 launch: no barrier 1.551 ms, every 64 columns 1.515, 32: 1.500, 16: 1.491,
    // 8: 1.487, 4: 1.487, 2: 1.497).  All waves of a workgroup march the same chunk, hence the
    // same trip count; a wave that has exited is not waited for.
    using std::integral_constant;
    if constexpr (MGX_TRIP != 0) {
        // Trips of PF columns: the queues hold the whole trip; at the start of a trip they are
        // emptied into the trip's own registers and refilled with the NEXT trip's columns, every
        // load at once, so that each has the whole trip to arrive (a column-by-column refill gives
        // the last one a single step).  The last trip may march up to PF-2 columns past jlast:
        // loads are clamped, nothing past jb is stored or merged.
        static_assert(PF % 2 == 0 && PF <= 8, "trips are 2, 4, 6 or 8 columns");
        for (int jin = jstart; jin <= jlast; jin += PF) {
#if MGX_LOCKSTEP > 0
            if (PF >= MGX_LOCKSTEP || ((jin - jstart) & (MGX_LOCKSTEP - 1)) == 0) __builtin_amdgcn_s_barrier();
#endif
            P2 cu[PF], cf[PF];
#pragma unroll
            for (int k = 0; k < PF; k++) {
                cu[k] = Pu[k];
                cf[k] = Pf[k];
            }
#pragma unroll
            for (int k = 0; k < PF; k++) {
                if (!ZERO_IN) Pu[k] = load_col(pua, pub, jin + PF + k);
                Pf[k] = load_col(pfa, pfb, jin - 1 + PF + k, true);
            }
            step_core(jin, std::false_type{}, cu[0], cf[0]);
            step_core(jin + 1, std::true_type{}, cu[1], cf[1]);
            if constexpr (PF >= 4) {
                step_core(jin + 2, std::false_type{}, cu[2], cf[2]);
                step_core(jin + 3, std::true_type{}, cu[3], cf[3]);
            }
            if constexpr (PF >= 6) {
                step_core(jin + 4, std::false_type{}, cu[4], cf[4]);
                step_core(jin + 5, std::true_type{}, cu[5], cf[5]);
            }
            if constexpr (PF >= 8) {
                step_core(jin + 6, std::false_type{}, cu[6], cf[6]);
                step_core(jin + 7, std::true_type{}, cu[7], cf[7]);
            }
        }
    } else if constexpr ((MGX_UNROLL6 && !EDGE) || PF != 2) {
        // Six columns per trip: the register queues rotate with periods 2 (load-ahead, column
        // parity), 3 (the three newest columns of a stage, the residual columns, the coarse
        // columns every other step) and NS+1, so after six steps every value is back in the
        // register it started in and the rotations cost no moves.  The last trip may march up to
        // four columns past jlast: loads are clamped, nothing past jb is stored or merged.
        static_assert(6 % PF == 0, "the load-ahead depth must divide the trip length");
        for (int jin = jstart; jin <= jlast; jin += 6) {
#if MGX_LOCKSTEP > 0
            __builtin_amdgcn_s_barrier();
#endif
            step(jin, std::false_type{}, integral_constant<int, 0 % PF>{});
            step(jin + 1, std::true_type{}, integral_constant<int, 1 % PF>{});
            step(jin + 2, std::false_type{}, integral_constant<int, 2 % PF>{});
            step(jin + 3, std::true_type{}, integral_constant<int, 3 % PF>{});
            step(jin + 4, std::false_type{}, integral_constant<int, 4 % PF>{});
            step(jin + 5, std::true_type{}, integral_constant<int, 5 % PF>{});
        }
    } else {
        for (int jin = jstart; jin <= jlast; jin += 2) {
#if MGX_LOCKSTEP > 0
            if (((jin - jstart) & (MGX_LOCKSTEP - 1)) == 0) __builtin_amdgcn_s_barrier();
#endif
            step(jin, std::false_type{}, integral_constant<int, 0>{});
            step(jin + 1, std::true_type{}, integral_constant<int, 1>{});
        }
    }

    if (NORM) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double o = __shfl_xor(nacc, off, 64);
            nacc = (nacc < o) ? o : nacc;
        }
        if (lane == 0) atomicMax(A.slot, norm_encode(nacc));
    }
}

// -DMGX_STREAM_TRACE: every wave of a four-sweep up+down launch of TWO column ranges (the edge
// bands of a slab in a V(2,2) cycle) leaves its start and end time (100 MHz clock) and where it ran; mgx_debug_stream_trace()
// copies the marks out (tools/trace_edge_launch.py)
#ifdef MGX_STREAM_TRACE
constexpr int STREAM_TRACE_WAVES = 16384;
__device__ unsigned long long g_stream_trace[4 * STREAM_TRACE_WAVES];
__device__ int g_stream_trace_ranges = 2;  // which launches: those of this many column ranges
#endif

template <int NS, int PA, int NM, int RES, bool ZERO_IN, bool STORE>
__global__ __launch_bounds__(512) void k_stream(const StreamArgs A)
{
    using G = Geo<NS, PA, NM, RES>;
    int bx = (int)blockIdx.x, by = (int)blockIdx.y;
#ifdef MGX_STREAM_TRACE
    const int trace_id = ((int)blockIdx.x + (int)blockIdx.y * (int)gridDim.x) * A.nwv + (int)(threadIdx.x >> 6);
    const bool traced = NS == 4 && RES == 1 && !ZERO_IN && (A.jhi2 > A.jlo2 ? 2 : 1) == g_stream_trace_ranges &&
                        (threadIdx.x & 63) == 0 && trace_id < STREAM_TRACE_WAVES;
    if (traced) {
        g_stream_trace[4 * trace_id + 0] = wall_clock64();
        g_stream_trace[4 * trace_id + 2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) |
                                           ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
        g_stream_trace[4 * trace_id + 3] = 0;
    }
#endif
    if (A.remap) {
        // Workgroups are dealt to the 8 XCDs round-robin by linear id; they are renumbered so
        // that one XCD (one L2) gets a contiguous run of the (strip group, chunk) plane, with
        // neighbours along the strips -- which load the same halo rows at the same time --
        // next to each other (MGX_XCD_REMAP=1, the default: level-0 launch -0.5 %, cycle
        // -0.6 %; 2 = neighbours along the chunks: +10 %; 0 = none)
        const int nsg = (int)gridDim.x, nch = (int)gridDim.y, N = nsg * nch;
        const int id = bx + by * nsg, x = id & 7, q = N >> 3, r = N & 7;
        const int L = x * q + (x < r ? x : r) + (id >> 3);
        if (A.remap == 1) {
            bx = L % nsg;
            by = L / nsg;
        } else {
            by = L % nch;
            bx = L / nch;
        }
    }
    const int strip = bx * A.nwv + (threadIdx.x >> 6);  // A.nwv strips (waves) per workgroup
    if (strip >= A.nstrips) return;
    const int r0 = strip * G::S - G::PADLO;
    const bool edge = (r0 < 1) || (r0 + 127 > A.m - 2);
    // does the march of this chunk (halo, load-ahead and the columns a six-step trip may add
    // included) meet a Dirichlet column or leave the array?
    const bool second = by >= A.nch1;
    const int ja = second ? A.jlo2 + (by - A.nch1) * A.CJ : A.jlo + by * A.CJ;
    const int jb = min(ja + A.CJ, second ? A.jhi2 : A.jhi);
    const int mlo = ja - G::HLC - 2, mhi = jb + G::JLAST + 8;
    const bool cf = mlo < 0 || mhi >= A.ncols || (A.jfix_lo >= 0 && mlo <= A.jfix_lo) || (A.jfix_hi >= 0 && mhi >= A.jfix_hi);
    if (edge)
        stream_wave<NS, PA, NM, RES, ZERO_IN, STORE, true, true>(A, strip, by);
    else if (cf)
        stream_wave<NS, PA, NM, RES, ZERO_IN, STORE, false, true>(A, strip, by);
    else
        stream_wave<NS, PA, NM, RES, ZERO_IN, STORE, false, false>(A, strip, by);
#ifdef MGX_STREAM_TRACE
    if (traced) {
        g_stream_trace[4 * trace_id + 1] = wall_clock64();
        g_stream_trace[4 * trace_id + 3] = 1 + (edge ? 1 : 0) + (cf ? 2 : 0) + ((unsigned long long)(jb - ja) << 8) +
                                           ((unsigned long long)strip << 16) + ((unsigned long long)by << 32);
    }
#endif
}

int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

// Shape of one launch: strips (waves) per workgroup, columns per chunk, LDS to reserve.
struct LaunchShape {
    int nwv = 4, CJ = 256;
    size_t lds = 0;
};

// Plain shape.  (For the tall slabs of the coarser levels of a partitioned run -- 8193 x 1100,
// 4097 x 540, the iterate known to be zero -- chunks two, four and eight times as long as this
// rule's 16 / 8 columns were measured as tuner candidates and lost, 64 -> 68 -> 87 -> 127 us per
// three launches: those launches want workgroups, not fewer halo columns.)
// Chunk length: long enough to amortise the pipeline warm-up on a large
// level (256 columns: 4 % of re-read), short enough that a small level still spreads
// over the chip (down to 4 columns: a small level is latency-bound, so a shorter march
// beats less redundant work).
int cu_count()
{
    static const int ncu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        return n;
    }();
    return ncu;
}

LaunchShape plain_shape(int nstrips, int ncolsout, int min_per_cu = 0, int min_cj = 0)
{
    static const int cj_max = env_int("MGX_CJ", 256), cj_min = env_int("MGX_CJ_MIN", 4);
    static const int min_blocks = env_int("MGX_MIN_BLOCKS", 1024);
    static const int bal_cj = env_int("MGX_BALANCE_CJ", 192);
    LaunchShape sh;
    const int nsg = (nstrips + sh.nwv - 1) / sh.nwv;
    sh.CJ = cj_max;
    while (sh.CJ > cj_min && (long)nsg * ((ncolsout + sh.CJ - 1) / sh.CJ) < min_blocks) {
        // below 16 columns the halo columns of a chunk outweigh its own: not while half as
        // many workgroups still cover the chip twice (2049^2: legs 26.4 + 19.1 -> 25.4 + 17.8 us)
        if (sh.CJ <= 16 && 2L * nsg * ((ncolsout + sh.CJ - 1) / sh.CJ) >= min_blocks) break;
        sh.CJ >>= 1;
    }
    // A launch big enough for chunks of ~192 columns at two or more workgroups per CU gets
    // the same number of workgroups on every CU (all are resident at once): the chunk
    // count is rounded down to a multiple of #CUs / strip groups.  8193^2, V(2,2): up leg
    // 274 -> 257 us, down leg 158 -> 148 us against 1152 workgroups of 128 columns.
    if (bal_cj > 0) {
        const long ncu = cu_count();
        const long want = (long)nsg * ((ncolsout + bal_cj - 1) / bal_cj);
        long q = (want + ncu / 2) / ncu;
        if (want >= 2 * ncu || (min_per_cu > 0 && q < min_per_cu)) {
            // (with min_per_cu also a smaller launch: that many workgroups per CU, if the
            // chunks then keep min_cj columns)
            if (q < min_per_cu) q = min_per_cu;
            const long nch = q * ncu / nsg;
            if (nch >= 1) {
                int cj = (int)((ncolsout + nch - 1) / nch);
                cj += cj & 1;
                if (want >= 2 * ncu || cj >= min_cj) sh.CJ = cj;
            }
        }
    }
    return sh;
}

// Shape for a big launch that streams BOTH the iterate and the right-hand side (the
// level-0 launches).  Two measured effects, 16385^2 V(2,2) up+down launch, 1.73 ms plain:
//  * few resident waves: reserving 64 KB of (unused) LDS per workgroup leaves two
//    four-wave workgroups per CU instead of the three the kernel's ~150 VGPRs allow (8
//    instead of 12 waves per CU), and eight strips per workgroup instead of four (one such
//    workgroup per CU) share more halo rows inside a workgroup (1.73 -> 1.62 -> 1.56 ms);
//  * whole rounds: capped workgroups run in rounds of 2 x #CUs, and a last round that is
//    nearly empty wastes the chip, so the number of chunks is made a whole number of
//    rounds (rounded down), at least three, with chunks of about 192 columns.
// Launches that read only f (levels >= 1) and launches too small for chunks of 64
// columns lose from the cap and keep the plain shape.  Defaults are the measured optimum
// at 8193^2..32769^2; the environment variables exist for re-tuning.
// A second variant, chunks of about 256 columns in at least two rounds, is a candidate of its
// own: at 16385^2 it wins (1.485 -> 1.473 ms, reads 19.2 -> 18.9 B/point), at 22529^2 and
// 32769^2 it loses 1 %.
bool capped_shape(int nstrips, int ncolsout, LaunchShape &sh, bool longer = false)
{
    static const int cap_lds = env_int("MGX_CAP_LDS", 65536), cap_cj0 = env_int("MGX_CAP_CJ", 192);
    static const int cap_rounds0 = env_int("MGX_CAP_ROUNDS", 3), cap_wgs = env_int("MGX_CAP_WGS", 2);
    static const int cap_minblk = env_int("MGX_CAP_MINBLK", 512);
    static const int cap_cj1 = env_int("MGX_CAP_CJ_LONG", 256), cap_rounds1 = env_int("MGX_CAP_ROUNDS_LONG", 2);
    const int cap_cj = longer ? cap_cj1 : cap_cj0, cap_rounds = longer ? cap_rounds1 : cap_rounds0;
    const int ncu = cu_count();
    if (cap_lds <= 0 || cap_lds > 65536) return false;
    const long slots = (long)cap_wgs * ncu;
    for (int nwv = 8; nwv >= 4; nwv >>= 1) {  // eight strips per workgroup if the chunks stay long
        const long nsg = (nstrips + nwv - 1) / nwv;
        const long want = nsg * ((ncolsout + cap_cj - 1) / cap_cj);
        long rounds = (want + slots / 2) / slots;
        if (rounds < cap_rounds) rounds = cap_rounds;
        const long nch = rounds * slots / nsg;
        if (nch < 1 || nsg * nch < cap_minblk) continue;
        int cj = (int)((ncolsout + nch - 1) / nch);
        cj += cj & 1;
        if (cj < (nwv == 8 ? 128 : 64)) continue;
        sh.nwv = nwv;
        sh.CJ = cj;
        sh.lds = (size_t)cap_lds;
        return true;
    }
    return false;
}

// Which shape is fastest for a big launch that streams both arrays depends on more than
// its size (a full 16385^2 level gains 8 % from the cap, a tall slab of the same level or a
// full 8193^2 level loses 5 %), so it is MEASURED where it runs.  The first launches of a
// given geometry run the candidate shapes in turn, WINDOW consecutive launches each, every
// launch between two events on the launch stream, and the shape with the smallest total is
// kept.  Every shape computes the same bits, so the choice affects time only.  The
// last launch of a window waits for its completion (the host would otherwise run far
// ahead of the GPU and never see a result): one stream synchronisation per candidate, once
// per geometry and process; after that nothing blocks.  MGX_TUNE=0 turns this off
// (square-ish launches of at least 12000 columns then take the cap, others the plain shape).
struct ShapeTuner {
    static constexpr int MAXC = 8, WINDOW = 3;
    struct Entry {
        int device = -1, m = 0, ncols = 0, nranges = 1;
        int ncand = 0;
        LaunchShape cand[MAXC];
        float time[MAXC] = {0};
        int chosen = -1, cur = 0, count = -1;  // count -1: one untimed launch first (cold start)
        hipEvent_t e0[WINDOW] = {nullptr}, e1[WINDOW] = {nullptr};
    };
    std::mutex mu;
    std::vector<Entry> entries;
    ShapeTuner()
    {
        std::lock_guard<std::mutex> lock(registry_mutex());
        registry().push_back(this);
    }
    static std::mutex &registry_mutex()
    {
        static std::mutex m;
        return m;
    }
    static std::vector<ShapeTuner *> &registry()
    {
        static std::vector<ShapeTuner *> all;
        return all;
    }
    void forget()
    {
        std::lock_guard<std::mutex> lock(mu);
        for (Entry &e : entries)
            for (int w = 0; w < WINDOW; w++) {
                if (e.e0[w]) (void)hipEventDestroy(e.e0[w]);
                if (e.e1[w]) (void)hipEventDestroy(e.e1[w]);
            }
        entries.clear();
    }
    Entry &find(int device, int m, int ncols, int nranges)
    {
        for (Entry &e : entries)
            if (e.device == device && e.m == m && e.ncols == ncols && e.nranges == nranges) return e;
        entries.emplace_back();
        Entry &e = entries.back();
        e.device = device;
        e.m = m;
        e.ncols = ncols;
        e.nranges = nranges;
        return e;
    }
    int sig[6] = {0, 0, 0, 0, 0, 0};  // NS, PA, NM, RES, ZERO_IN, STORE of the kernel instance this tuner serves
};

thread_local bool g_tune_now = false;
bool tune_now_flag() { return g_tune_now; }

template <int NS, int PA, int NM, int RES, bool ZERO_IN, bool STORE>
void report_choice(const ShapeTuner::Entry &e, int m, int ncolsout)
{
    static const int verbose = env_int("MGX_TUNE_VERBOSE", 0);
    if (!verbose) return;
    fprintf(stderr, "mgx tune k_stream<%d,%d,%d,%d,%d,%d> m=%d ncols=%d:", NS, PA, NM, RES, (int)ZERO_IN, (int)STORE, m,
            ncolsout);
    for (int k = 0; k < e.ncand; k++)
        fprintf(stderr, " [%d strips x %d cols, lds %zu: %.1f us / %d launches]%s", e.cand[k].nwv, e.cand[k].CJ,
                e.cand[k].lds, 1e3 * e.time[k], ShapeTuner::WINDOW, k == e.chosen ? "*" : "");
    fprintf(stderr, "\n");
}

bool same_shape(const LaunchShape &a, const LaunchShape &b) { return a.nwv == b.nwv && a.CJ == b.CJ && a.lds == b.lds; }

template <int NS, int PA, int NM, int RES, bool ZERO_IN, bool STORE>
hipError_t launch_shape(StreamArgs A, const LaunchShape &sh, hipStream_t s)
{
    static const int remap = env_int("MGX_XCD_REMAP", 1);
    A.CJ = sh.CJ;
    A.nwv = sh.nwv;
    A.remap = remap;
    A.nch1 = (A.jhi - A.jlo + sh.CJ - 1) / sh.CJ;
    A.nchunks = A.nch1 + (A.jhi2 > A.jlo2 ? (A.jhi2 - A.jlo2 + sh.CJ - 1) / sh.CJ : 0);
    const int nsg = (A.nstrips + sh.nwv - 1) / sh.nwv;
    hipLaunchKernelGGL((k_stream<NS, PA, NM, RES, ZERO_IN, STORE>), dim3((unsigned)nsg, (unsigned)A.nchunks),
                       dim3(64 * (unsigned)sh.nwv), sh.lds, s, A);
    return hipGetLastError();
}

// One workgroup slot per CU filled exactly once: `per_cu` workgroups of `nwv` strips on every CU,
// all resident from the start, chunks as long as that allows.  For a tall slab of a partitioned
// level (16385 rows x 2048 columns: 18 groups of eight strips x 14 chunks of 146 columns) the
// shapes above leave chunks of 64-90 columns, of which the 11 halo columns are a sixth.
bool one_round_shape(int nstrips, int ncolsout, int nwv, int per_cu, LaunchShape &sh, int min_cj = 48)
{
    const long nsg = (nstrips + nwv - 1) / nwv;
    const long nch = (long)per_cu * cu_count() / nsg;
    if (nch < 1) return false;
    int cj = (int)((ncolsout + nch - 1) / nch);
    cj += cj & 1;
    if (cj < min_cj) return false;
    sh.nwv = nwv;
    sh.CJ = cj;
    sh.lds = 0;
    return true;
}

template <int NS, int PA, int NM, int RES, bool ZERO_IN, bool STORE>
hipError_t launch_one(StreamArgs A, hipStream_t s)
{
    using G = Geo<NS, PA, NM, RES>;
    const int m = A.m;
    A.nstrips = (m - 1 + G::S - 1) / G::S;  // strips own rows 0..m-2
    const int nranges = A.jhi2 > A.jlo2 ? 2 : 1;
    const int ncolsout = (A.jhi - A.jlo) + (nranges == 2 ? A.jhi2 - A.jlo2 : 0);
    if (ncolsout <= 0 || m < 3) return hipSuccess;
    if (A.jhi <= A.jlo) {  // only the second range holds columns
        A.jlo = A.jlo2;
        A.jhi = A.jhi2;
        A.jlo2 = A.jhi2 = 0;
    }
    const LaunchShape plain = plain_shape(A.nstrips, ncolsout);
    // candidates: the static guess first (it is the fallback), then the others
    LaunchShape cands[ShapeTuner::MAXC];
    int ncands = 0;
    auto add = [&](const LaunchShape &c) {
        for (int k = 0; k < ncands; k++)
            if (same_shape(cands[k], c)) return;
        if (ncands < ShapeTuner::MAXC) cands[ncands++] = c;
    };
    add(plain);
    if (!ZERO_IN && (long)m * ncolsout >= (8L << 20)) {
        LaunchShape capped = plain;
        if (capped_shape(A.nstrips, ncolsout, capped)) {
            add(capped);
            if (ncands == 2 && 4L * ncolsout >= 3L * m && ncolsout >= 12000) std::swap(cands[0], cands[1]);  // a big square level
            LaunchShape longer = plain;
            if (capped_shape(A.nstrips, ncolsout, longer, true)) add(longer);
        }
        add(plain_shape(A.nstrips, ncolsout, 3, 60));
        if (4L * ncolsout < 3L * m) {  // a tall slab
            LaunchShape r = plain;
            if (one_round_shape(A.nstrips, ncolsout, 8, 1, r)) add(r);
            if (one_round_shape(A.nstrips, ncolsout, 4, 2, r)) add(r);
        }
    } else if (!ZERO_IN && A.nstrips >= 32 && (long)m * ncolsout >= (1L << 20)) {
        // A narrow band of a tall level (the columns next to a neighbour, launched ahead of the
        // rest so that the exchange can start): the plain rule would cut it into chunks of 4
        // columns, each marching its 11 halo columns too.  Longer chunks move fewer bytes, fewer
        // chunks march longer: measured (two, four and eight strips per workgroup give the same 40 us
        // at their best chunk length: four it is).
        const int width = std::max(A.jhi - A.jlo, A.jhi2 - A.jlo2);
        // Such a launch is bound by instruction issue, not by memory: a wave alone on its SIMD
        // marches a column in 0.7 us whatever the depth of its load queue, and two waves on one
        // SIMD take turns (one finishes as if alone, the other up to twice as late).  With more
        // workgroups than CUs the launch therefore lasts as long as two of them in a row: the
        // candidate that comes first leaves every CU at most one (16385 rows, two bands of 88
        // columns: three chunks of 30 per band instead of six of 16).
        {
            const long nsg4 = (A.nstrips + 3) / 4;
            const long per_range = cu_count() / nsg4 / nranges;
            if (per_range >= 1) {
                LaunchShape c = plain;
                c.nwv = 4;
                c.CJ = (int)((width + per_range - 1) / per_range);
                c.CJ += c.CJ & 1;
                if (c.CJ >= 6) add(c);
            }
        }
        for (int div : {2, 4, 6, 8, 12}) {
            LaunchShape c = plain;
            c.nwv = 4;
            c.CJ = (width + div - 1) / div;
            c.CJ += c.CJ & 1;
            if (c.CJ >= 6) add(c);
        }
    }
    if (ZERO_IN && nranges == 1 && (long)m * ncolsout >= (1L << 18) && (long)m * ncolsout <= (12L << 20)) {
        // A level of 0.25-12 M points whose iterate starts from zero (level 3 of a big problem,
        // the slabs of levels 1-3 of a partitioned one): the plain rule's short chunks (14
        // columns at 2049^2) march two thirds as many halo columns again.  One, two or three
        // workgroups on every CU, chunks as long as that allows: 2049^2 legs 22.7 + 15.4 ->
        // 21.3 + 14.5 us.  (Not for bigger launches: a 4097^2 level repeated for the
        // measurement stays in the memory-side cache and the winner of that contest, 50-column
        // chunks, then runs 62 us inside a cycle against the plain rule's 60.8.)
        for (int per_cu : {1, 2, 3}) {
            LaunchShape r = plain;
            if (one_round_shape(A.nstrips, ncolsout, 4, per_cu, r, 8)) add(r);
        }
    }
    static const int tune = env_int("MGX_TUNE", 1);
    if (ncands == 1 || !tune) return launch_shape<NS, PA, NM, RES, ZERO_IN, STORE>(A, cands[0], s);

    static ShapeTuner tuner;  // one per kernel instance
    tuner.sig[0] = NS, tuner.sig[1] = PA, tuner.sig[2] = NM, tuner.sig[3] = RES, tuner.sig[4] = ZERO_IN, tuner.sig[5] = STORE;
    std::lock_guard<std::mutex> lock(tuner.mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    ShapeTuner::Entry &e = tuner.find(dev, m, ncolsout, nranges);
    if (e.ncand == 0) {
        for (int k = 0; k < ncands; k++) e.cand[e.ncand++] = cands[k];
        for (int w = 0; w < ShapeTuner::WINDOW; w++)
            if (hipEventCreate(&e.e0[w]) != hipSuccess || hipEventCreate(&e.e1[w]) != hipSuccess) e.chosen = 0;
    }
    if (e.chosen >= 0) return launch_shape<NS, PA, NM, RES, ZERO_IN, STORE>(A, e.cand[e.chosen], s);
    if (tune_now_flag()) {
        // Decide at once: a launch reads uin/f/cin and writes uout/cout (never its own input)
        // and merges the norm by a max, so repeating it leaves the same result.  Every
        // candidate runs once untimed and WINDOW times between one event pair, three times.
        hipError_t rc = hipSuccess;
        bool ok = true;
        // three interleaved passes over the candidates, the smallest time of each counts: a
        // disturbance during one window (another tenant of the node, a clock ramp) must not
        // decide between shapes that differ by 2 %
        for (int pass = 0; pass < 3 && ok; pass++)
            for (int k = 0; k < e.ncand && ok; k++) {
                rc = launch_shape<NS, PA, NM, RES, ZERO_IN, STORE>(A, e.cand[k], s);
                ok = rc == hipSuccess && hipEventRecord(e.e0[0], s) == hipSuccess;
                for (int w = 0; ok && w < ShapeTuner::WINDOW; w++) {
                    rc = launch_shape<NS, PA, NM, RES, ZERO_IN, STORE>(A, e.cand[k], s);
                    ok = rc == hipSuccess;
                }
                float t = 0.f;
                ok = ok && hipEventRecord(e.e1[0], s) == hipSuccess && hipEventSynchronize(e.e1[0]) == hipSuccess &&
                     hipEventElapsedTime(&t, e.e0[0], e.e1[0]) == hipSuccess;
                if (ok && (pass == 0 || t < e.time[k])) e.time[k] = t;
            }
        if (!ok) {
            if (rc == hipSuccess) (void)hipGetLastError();
            e.chosen = 0;
            return rc != hipSuccess ? rc : launch_shape<NS, PA, NM, RES, ZERO_IN, STORE>(A, e.cand[0], s);
        }
        e.chosen = 0;
        for (int k = 1; k < e.ncand; k++)
            if (e.time[k] < e.time[e.chosen]) e.chosen = k;
        report_choice<NS, PA, NM, RES, ZERO_IN, STORE>(e, m, ncolsout);
        return hipSuccess;
    }
    if (e.count < 0) {
        e.count = 0;
        return launch_shape<NS, PA, NM, RES, ZERO_IN, STORE>(A, e.cand[0], s);
    }
    // (an event record fails e.g. on a stream under capture: tuning is then given up)
    bool ok = hipEventRecord(e.e0[e.count], s) == hipSuccess;
    const hipError_t rc = ok ? launch_shape<NS, PA, NM, RES, ZERO_IN, STORE>(A, e.cand[e.cur], s)
                             : launch_shape<NS, PA, NM, RES, ZERO_IN, STORE>(A, e.cand[0], s);
    ok = ok && rc == hipSuccess && hipEventRecord(e.e1[e.count], s) == hipSuccess;
    if (ok && ++e.count == ShapeTuner::WINDOW) {
        float ms = 0.f;
        ok = hipEventSynchronize(e.e1[ShapeTuner::WINDOW - 1]) == hipSuccess;
        for (int w = 0; ok && w < ShapeTuner::WINDOW; w++) {
            float t = 0.f;
            ok = hipEventElapsedTime(&t, e.e0[w], e.e1[w]) == hipSuccess;
            ms += t;
        }
        if (ok) {
            e.time[e.cur] = ms;
            e.count = 0;
            if (++e.cur == e.ncand) {
                e.chosen = 0;
                for (int k = 1; k < e.ncand; k++)
                    if (e.time[k] < e.time[e.chosen]) e.chosen = k;
                report_choice<NS, PA, NM, RES, ZERO_IN, STORE>(e, m, ncolsout);
            }
        }
    }
    if (!ok) {
        (void)hipGetLastError();
        e.chosen = 0;
    }
    return rc;
}

} // namespace

// One line per measured geometry: kernel instance, rows, columns produced, chosen shape.
std::string tune_report()
{
    std::string out;
    char buf[256];
    std::lock_guard<std::mutex> lock(ShapeTuner::registry_mutex());
    for (ShapeTuner *t : ShapeTuner::registry()) {
        std::lock_guard<std::mutex> lk(t->mu);
        for (const ShapeTuner::Entry &e : t->entries) {
            if (e.chosen < 0) continue;
            const LaunchShape &c = e.cand[e.chosen];
            snprintf(buf, sizeof(buf), "k_stream<%d,%d,%d,%d,%d,%d> dev %d m=%d cols=%d%s: %d strips x %d cols%s (%d of %d)\n",
                     t->sig[0], t->sig[1], t->sig[2], t->sig[3], t->sig[4], t->sig[5], e.device, e.m, e.ncols,
                     e.nranges == 2 ? " (2 ranges)" : "", c.nwv, c.CJ, c.lds ? ", capped" : "", e.chosen + 1, e.ncand);
            out += buf;
        }
    }
    return out;
}

void tune_forget()
{
    std::lock_guard<std::mutex> lock(ShapeTuner::registry_mutex());
    for (ShapeTuner *t : ShapeTuner::registry()) t->forget();
}

bool set_tune_now(bool on)
{
    const bool was = g_tune_now;
    g_tune_now = on;
    return was;
}

int stream_halo(const StreamOp &op)
{
    return op.ns + (op.res ? 2 : (op.nm == op.ns && op.nm >= 0 ? 1 : 0));
}

// The instances that exist.  Anything else is rejected (hipErrorInvalidValue),
// and the schedules in mgx_api.hip only compose these.
hipError_t launch_stream(const StreamOp &op, const StreamArgs &A, hipStream_t s)
{
    const int ns = op.ns, pa = op.pa, nm = op.nm, res = op.res;
    const bool z = op.zero_in != 0, st = op.store != 0;
#define MGX_TRY(NS, PA, NM, RES, Z, ST)                                                     \
    if (ns == (NS) && pa == (PA) && nm == (NM) && res == (RES) && z == (Z) && st == (ST))     \
        return launch_one<NS, PA, NM, RES, Z, ST>(A, s);
#define MGX_FOR_NS0(M, ...) M(0, __VA_ARGS__) M(1, __VA_ARGS__) M(2, __VA_ARGS__) M(3, __VA_ARGS__) M(4, __VA_ARGS__)
#define MGX_FOR_NS1(M, ...) M(1, __VA_ARGS__) M(2, __VA_ARGS__) M(3, __VA_ARGS__) M(4, __VA_ARGS__)

    // down leg: sweeps + residual + restriction (input read or known zero; output stored or not)
#define MGX_DOWN(NS, Z, ST) MGX_TRY(NS, -1, -1, 1, Z, ST)
    MGX_FOR_NS0(MGX_DOWN, false, true)
    MGX_DOWN(0, false, false)
    MGX_FOR_NS1(MGX_DOWN, true, true)
    MGX_FOR_NS0(MGX_DOWN, true, false)
    // the first down leg of a call together with the norm of the initial residual (main.cpp:307)
    MGX_TRY(0, -1, 0, 1, false, false)
    MGX_TRY(1, -1, 0, 1, false, true) MGX_TRY(2, -1, 0, 1, false, true)
    MGX_TRY(3, -1, 0, 1, false, true) MGX_TRY(4, -1, 0, 1, false, true)
    // sweeps only (links of a chain longer than 4 sweeps)
#define MGX_SWEEP(NS, Z) MGX_TRY(NS, -1, -1, 0, Z, true)
    MGX_FOR_NS1(MGX_SWEEP, false)
    MGX_FOR_NS1(MGX_SWEEP, true)
    // sweeps (or none) + residual norm
#define MGX_NORM(NS, ST) MGX_TRY(NS, -1, NS, 0, false, ST)
    MGX_NORM(0, false)
    MGX_FOR_NS1(MGX_NORM, true)
    // up leg: correction at stage 0, sweeps, optional norm
#define MGX_UP(NS, Z) MGX_TRY(NS, 0, -1, 0, Z, true) MGX_TRY(NS, 0, NS, 0, Z, true)
    MGX_FOR_NS0(MGX_UP, false)
    MGX_FOR_NS0(MGX_UP, true)
    // level 0, up leg of one cycle + down leg of the next: correction at stage 0,
    // norm after nu2 sweeps, restriction after nu2+nu1
#define MGX_UPDOWN(NS, NM) MGX_TRY(NS, 0, NM, 1, false, true)
    MGX_UPDOWN(1, 0) MGX_UPDOWN(1, 1)
    MGX_UPDOWN(2, 0) MGX_UPDOWN(2, 1) MGX_UPDOWN(2, 2)
    MGX_UPDOWN(3, 0) MGX_UPDOWN(3, 1) MGX_UPDOWN(3, 2) MGX_UPDOWN(3, 3)
    MGX_UPDOWN(4, 0) MGX_UPDOWN(4, 1) MGX_UPDOWN(4, 2) MGX_UPDOWN(4, 3) MGX_UPDOWN(4, 4)
    // levels >= 1, up leg that recomputes the nu1 down sweeps from zero: correction at stage nu1
#define MGX_REUP(NS, PA) MGX_TRY(NS, PA, -1, 0, true, true)
    MGX_REUP(1, 1)
    MGX_REUP(2, 1) MGX_REUP(2, 2)
    MGX_REUP(3, 1) MGX_REUP(3, 2) MGX_REUP(3, 3)
    MGX_REUP(4, 1) MGX_REUP(4, 2) MGX_REUP(4, 3) MGX_REUP(4, 4)
#undef MGX_TRY
    return hipErrorInvalidValue;
}

} // namespace mgx

#ifdef MGX_STREAM_TRACE
// ranges > 0: forget the marks and trace the launches of that many column ranges from now on;
// out: 4 x STREAM_TRACE_WAVES values (start, end, where, what) or NULL
extern "C" int mgx_debug_stream_trace(int ranges, unsigned long long *out)
{
    hipError_t e = hipSuccess;
    void *p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(mgx::g_stream_trace));
    if (e == hipSuccess && ranges > 0) e = hipMemset(p, 0, sizeof(unsigned long long) * 4 * mgx::STREAM_TRACE_WAVES);
    if (e == hipSuccess && ranges > 0) e = hipMemcpyToSymbol(HIP_SYMBOL(mgx::g_stream_trace_ranges), &ranges, sizeof(int));
    if (e == hipSuccess && out)
        e = hipMemcpy(out, p, sizeof(unsigned long long) * 4 * mgx::STREAM_TRACE_WAVES, hipMemcpyDeviceToHost);
    return (int)e;
}
#endif
