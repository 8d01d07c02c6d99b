// mgx_stream.hip -- fused streaming kernels for gfx950.
//
// One kernel template covers every fused step of the V-cycle:
//
//   k_stream<PRO, NS, RES, ZERO_IN>
//     PRO      1: stage 0 is v + P(coarse v)   (refine + `cur += refined`,
//                 reference main.cpp:339-341)
//     NS       number of damped-Jacobi sweeps applied in registers
//                 (relax, main.cpp:171-199)
//     RES      1: residual of the result, restricted into the coarse f
//                 (residual + coarsen, main.cpp:325-327)
//              2: residual of the result reduced to its infinity norm
//                 (calcNorm(residual), main.cpp:351)
//     ZERO_IN  1: the input iterate is known to be all zeros and is not read
//                 (a coarse level at the start of the down sweep, main.cpp:345)
//
// so "down" = <0,nu1,1,*> reads v,f once and writes v and the coarse f, and
// "up" = <1,nu2,0|2,0> reads the coarse v, v, f once and writes v: 26 B per
// fine point each instead of the 66 B the separate operators move.
//
// Mapping.  A wavefront owns a strip of 128 consecutive rows i (lane t holds
// rows r0+2t and r0+2t+1 in registers) and marches along the columns j of a
// chunk.  Column j of the input is loaded once (coalesced: the 64 lanes read
// 1 KiB of one column, 16 B per lane); sweep s of column j-s is computed as
// soon as sweep s-1 of columns j-s-1..j-s+1 exists, so the NS sweeps, the
// residual and the restriction form a software pipeline that lags the load
// front by NS+2 columns and never touches memory in between.  North/south
// neighbours (j-1, j+1) are the lane's own registers of the previous/next
// column; west/east neighbours (i-1, i+1) are the lane's other row or one
// cross-lane shift.  Waves are independent: no LDS, no barriers.  Strips
// overlap by the halo the pipeline consumes (NS+2 rows below, NS+1 above),
// chunks by the same number of columns; the overlap is recomputed, not
// communicated.
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

#include <type_traits>

namespace mgx {

namespace {

struct P2 {
    double a, b;  // rows ra (even offset in the strip) and ra+1
};

template <bool PRO, int NS, int RES>
struct Geo {
    // rows of halo the pipeline consumes below / above the rows a strip owns
    static constexpr int HL = NS + (RES == 1 ? 2 : RES == 2 ? 1 : 0);
    static constexpr int HH = NS + (RES >= 1 ? 1 : 0) + (PRO ? 1 : 0);
    static constexpr int PADLO = (HL + 1) & ~1;            // even, so that r0 is even
    static constexpr int S = (128 - PADLO - HH) & ~1;      // rows owned per strip (even)
    // columns of halo a chunk consumes before its first owned column
    static constexpr int HLC = HL;
    // last input column the march must reach, relative to the chunk end jb
    static constexpr int JLAST = NS + (RES == 1 ? 1 : RES == 2 ? 0 : -1);
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

__device__ __forceinline__ double from_lower_lane(double v) { return __shfl_up(v, 1, 64); }    // lane t-1 -> t
__device__ __forceinline__ double from_upper_lane(double v) { return __shfl_down(v, 1, 64); }  // lane t+1 -> t

__device__ __forceinline__ int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

template <bool PRO, int NS, int RES, bool ZERO_IN, bool EDGE>
__device__ __forceinline__ void stream_wave(const StreamArgs &A, const int strip, const int chunk)
{
    using G = Geo<PRO, NS, RES>;
    constexpr int PF = 2;  // columns of load-ahead
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
    const int ja = A.jlo + chunk * A.CJ;
    const int jb = min(ja + A.CJ, A.jhi);
    int jstart = ja - G::HLC;
    if (jstart < 0) jstart = 0;
    if ((jstart + A.gf0) & 1) jstart -= 1;  // start the march on an even global column
    const int jlast = jb + G::JLAST;
    const int ncols = A.ncols;
    const int jfl = A.jfix_lo, jfh = A.jfix_hi;

    const double *__restrict__ pua = A.uin + ra_c;
    const double *__restrict__ pub = A.uin + rb_c;
    const double *__restrict__ pfa = A.f + ra_c;
    const double *__restrict__ pfb = A.f + rb_c;
    double *__restrict__ po = A.uout + ra;

    auto load_col = [&](const double *__restrict__ pa, const double *__restrict__ pb, int j) -> P2 {
        const size_t off = (size_t)clampi(j, 0, ncols - 1) * ld;
        P2 r;
        r.a = pa[off];
        r.b = pb[off];  // adjacent to r.a on interior strips: one 16-byte load
        return r;
    };
    auto store_col = [&](int j, const P2 &v) {
        if (EDGE) {
            if (owna) po[(size_t)j * ld] = v.a;
            if (ownb) po[(size_t)j * ld + 1] = v.b;
        } else if (owna) {
            double *q = po + (size_t)j * ld;
            q[0] = v.a;
            q[1] = v.b;
        }
    };

    // ---- prolongation state (PRO) -------------------------------------------
    // coarse row of this lane: I = ra/2 (ra is even); its upper neighbour I+1
    // comes from lane t+1.
    const int I = ra >> 1;
    const int I_c = EDGE ? clampi(I, 0, A.mc - 1) : I;
    const double *__restrict__ pc = PRO ? (A.cin + I_c) : nullptr;
    auto load_c = [&](int J) -> double { return pc[(size_t)clampi(J, 0, A.nccols - 1) * (size_t)A.ldc]; };
    double c0 = 0, c0u = 0, c1 = 0, c1u = 0, cnext = 0;  // cols J, J+1 (own row, row I+1) and raw col J+2
    int Jl = 0;                                          // local coarse column J of the current even column
    if (PRO) {
        // primed one column back: the first (even) step rotates into place
        Jl = ((jstart + A.gf0) >> 1) - A.gc0 - 1;
        c1 = load_c(Jl + 1);
        cnext = load_c(Jl + 2);
        c1u = from_upper_lane(c1);
    }

    // ---- pipeline registers ---------------------------------------------------
    P2 U[NS + 1][3];  // U[s][2] = newest column of stage s, [1], [0] the two before it
    P2 Fq[NS + 2];    // Fq[k] = f of column jin-k (k >= 1)
    P2 R[3];          // residual columns jr-2, jr-1, jr
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
        Pf[k] = load_col(pfa, pfb, jstart - 1 + k);
    }

    double nacc = 0.0;  // RES==2: running max |r| (boundary residuals are +0.0)

    // One column step; ODD is the parity of the global column index of jin.
    auto step = [&](const int jin, auto odd_tag) {
        constexpr bool ODD = decltype(odd_tag)::value;
        // ---- take column jin from the load-ahead queue, refill it ----------
        P2 u0 = Pu[0];
        const P2 f1 = Pf[0];
#pragma unroll
        for (int k = 0; k + 1 < PF; k++) {
            Pu[k] = Pu[k + 1];
            Pf[k] = Pf[k + 1];
        }
        if (!ZERO_IN) Pu[PF - 1] = load_col(pua, pub, jin + PF);
        Pf[PF - 1] = load_col(pfa, pfb, jin - 1 + PF);

        // ---- stage 0: the input iterate, plus the coarse-grid correction ------
        if (PRO) {
            if (!ODD) {  // entered a new coarse column
                Jl += 1;
                c0 = c1;
                c0u = c1u;
                c1 = cnext;
                c1u = from_upper_lane(c1);
                cnext = load_c(Jl + 2);
            }
            double pa, pb;
            if (!ODD) {  // main.cpp:236, :242
                pa = c0;
                pb = 0.5 * (c0 + c0u);
            } else {  // main.cpp:239, :245
                pa = 0.5 * (c0 + c1);
                double s4 = c0 + c0u;
                s4 = s4 + c1;
                s4 = s4 + c1u;
                pb = 0.25 * s4;
            }
            const bool colfix = (jin == jfl) || (jin == jfh) || jin < 0 || jin >= ncols;
            const double na = u0.a + pa;  // main.cpp:341
            const double nb = u0.b + pb;
            u0.a = (colfix || fixa) ? u0.a : na;
            u0.b = (colfix || fixb) ? u0.b : nb;
        }
        U[0][0] = U[0][1];
        U[0][1] = U[0][2];
        U[0][2] = u0;
#pragma unroll
        for (int k = NS + 1; k > 1; k--) Fq[k] = Fq[k - 1];
        Fq[1] = f1;

        if (NS == 0 && PRO) {  // prolong + correct only: stage 0 is the output
            if (jin >= ja && jin < jb && jin != jfl && jin != jfh) store_col(jin, u0);
        }

        // ---- stages 1..NS: damped Jacobi sweeps ----------------------------------
#pragma unroll
        for (int s = 1; s <= NS; s++) {
            const int jc = jin - s;
            const P2 uS = U[s - 1][0], uC = U[s - 1][1], uN = U[s - 1][2];
            const bool colfix = (jc == jfl) || (jc == jfh);  // Dirichlet column: untouched by relax
            const double wa = from_lower_lane(uC.b);         // row ra-1
            const double eb = from_upper_lane(uC.a);         // row rb+1
            const double na = jacobi_pt(wa, uC.b, uS.a, uN.a, uC.a, Fq[s].a, c);
            const double nb = jacobi_pt(uC.a, eb, uS.b, uN.b, uC.b, Fq[s].b, c);
            P2 nw;
            nw.a = (colfix || fixa) ? uC.a : na;
            nw.b = (colfix || fixb) ? uC.b : nb;
            U[s][0] = U[s][1];
            U[s][1] = U[s][2];
            U[s][2] = nw;
            if (s == NS && jc >= ja && jc < jb && !colfix) store_col(jc, nw);
        }

        // ---- residual of stage NS at column jr ------------------------------------
        if (RES != 0) {
            const int jr = jin - NS - 1;
            const P2 uS = U[NS][0], uC = U[NS][1], uN = U[NS][2];
            const double wa = from_lower_lane(uC.b);
            const double eb = from_upper_lane(uC.a);
            P2 r;
            r.a = residual_pt(wa, uC.b, uS.a, uN.a, uC.a, Fq[NS + 1].a, c);
            r.b = residual_pt(uC.a, eb, uS.b, uN.b, uC.b, Fq[NS + 1].b, c);
            if (RES == 2) {
                const bool colok = jr >= ja && jr < jb && jr != jfl && jr != jfh;
                const double xa = __builtin_fabs(r.a), xb = __builtin_fabs(r.b);
                if (colok && owna) nacc = (nacc < xa) ? xa : nacc;
                if (colok && ownb) nacc = (nacc < xb) ? xb : nacc;
            } else {
                R[0] = R[1];
                R[1] = R[2];
                R[2] = r;
                // coarse column J is centred on the even fine column 2J = jr-1,
                // complete once the odd column jr = 2J+1 has its residual
                constexpr bool JR_ODD = ((NS + 1) & 1) ? !ODD : ODD;
                if (JR_ODD) {
                    const int jcen = jr - 1;
                    if (jcen >= ja && jcen < jb && jcen != jfl && jcen != jfh) {
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
                        if (owna && I >= 1 && I <= A.mc - 2) A.cout[(size_t)I + (size_t)Jc * (size_t)A.ldc] = outc;
                    }
                }
            }
        }
    };

    for (int jin = jstart; jin <= jlast; jin += 2) {
        step(jin, std::false_type{});
        step(jin + 1, std::true_type{});
    }

    if (RES == 2) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double o = __shfl_xor(nacc, off, 64);
            nacc = (nacc < o) ? o : nacc;
        }
        if (lane == 0) atomicMax(A.slot, norm_encode(nacc));
    }
}

template <bool PRO, int NS, int RES, bool ZERO_IN>
__global__ __launch_bounds__(256) void k_stream(const StreamArgs A)
{
    using G = Geo<PRO, NS, RES>;
    const int strip = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (strip >= A.nstrips) return;
    const int r0 = strip * G::S - G::PADLO;
    const bool edge = (r0 < 1) || (r0 + 127 > A.m - 2);
    if (edge)
        stream_wave<PRO, NS, RES, ZERO_IN, true>(A, strip, (int)blockIdx.y);
    else
        stream_wave<PRO, NS, RES, ZERO_IN, false>(A, strip, (int)blockIdx.y);
}

template <bool PRO, int NS, int RES, bool ZERO_IN>
hipError_t launch_one(StreamArgs A, hipStream_t s)
{
    using G = Geo<PRO, NS, RES>;
    const int m = A.m;
    A.nstrips = (m - 1 + G::S - 1) / G::S;  // strips own rows 0..m-2
    const int ncolsout = A.jhi - A.jlo;
    if (ncolsout <= 0 || m < 3) return hipSuccess;
    // chunk length: long enough to amortise the pipeline warm-up, short enough
    // that a small level still spreads over the chip
    int CJ = 128;
    const int nsg = (A.nstrips + 3) / 4;
    while (CJ > 16 && (long)nsg * ((ncolsout + CJ - 1) / CJ) < 2048) CJ >>= 1;
    A.CJ = CJ;
    A.nchunks = (ncolsout + CJ - 1) / CJ;
    hipLaunchKernelGGL((k_stream<PRO, NS, RES, ZERO_IN>), dim3((unsigned)nsg, (unsigned)A.nchunks), dim3(256), 0, s, A);
    return hipGetLastError();
}

template <bool PRO, int RES, bool ZERO_IN>
hipError_t launch_ns(int ns, const StreamArgs &A, hipStream_t s)
{
    switch (ns) {
    case 0: return launch_one<PRO, 0, RES, ZERO_IN>(A, s);
    case 1: return launch_one<PRO, 1, RES, ZERO_IN>(A, s);
    case 2: return launch_one<PRO, 2, RES, ZERO_IN>(A, s);
    case 3: return launch_one<PRO, 3, RES, ZERO_IN>(A, s);
    case 4: return launch_one<PRO, 4, RES, ZERO_IN>(A, s);
    default: return hipErrorInvalidValue;
    }
}

} // namespace

hipError_t launch_stream(int pro, int ns, int res, int zero_in, const StreamArgs &A, hipStream_t s)
{
    if (ns < 0 || ns > STREAM_MAX_SWEEPS) return hipErrorInvalidValue;
    if (pro) {
        if (zero_in) return hipErrorInvalidValue;
        if (res == 0) return launch_ns<true, 0, false>(ns, A, s);
        if (res == 2) return launch_ns<true, 2, false>(ns, A, s);
        return hipErrorInvalidValue;
    }
    if (ns == 0 && res == 0) return hipSuccess;  // nothing to do
    if (zero_in) {
        if (ns == 0) return hipErrorInvalidValue;
        if (res == 0) return launch_ns<false, 0, true>(ns, A, s);
        if (res == 1) return launch_ns<false, 1, true>(ns, A, s);
        return hipErrorInvalidValue;
    }
    if (res == 0) return launch_ns<false, 0, false>(ns, A, s);
    if (res == 1) return launch_ns<false, 1, false>(ns, A, s);
    return launch_ns<false, 2, false>(ns, A, s);
}

} // namespace mgx
