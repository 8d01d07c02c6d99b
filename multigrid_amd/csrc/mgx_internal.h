// Internal launcher interface between the C ABI (mgx_api.hip) and the gfx950
// kernels (mgx_kernels.hip, mgx_stream.hip).  Not installed; include/mgx.h is
// the public boundary.
//
// Geometry convention shared by every launcher: a level is an m-row
// column-major array (leading dimension m).  Launchers take a half-open
// column range [jlo, jhi) to update, so that the same kernels serve a whole
// grid (jlo = 1, jhi = m-1) and a j-slab of a partitioned grid.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <string>

namespace mgx {

// Order-preserving encoding of a norm value for unsigned atomic max:
// 0 <-> -infinity ("nothing seen"), otherwise bits(value >= +0.0) + 1.
__host__ __device__ inline unsigned long long norm_encode(double v)
{
    union { double d; unsigned long long u; } x;
    x.d = v;
    return (v == -__builtin_inf()) ? 0ull : x.u + 1ull;
}
__host__ __device__ inline double norm_decode(unsigned long long k)
{
    union { double d; unsigned long long u; } x;
    if (k == 0ull) return -__builtin_inf();
    x.u = k - 1ull;
    return x.d;
}

// ---- device helpers shared by the per-operator kernels ------------------------------
// residual at flat index k of an m-row grid, main.cpp:277: f - (uW+uE+uS+uN - u*4.0)*invh2
__device__ __forceinline__ double residual_at(const double *__restrict__ u, const double *__restrict__ f, size_t k,
                                              int m, double invh2)
{
    double s = u[k - 1] + u[k + 1];
    s = s + u[k - m];
    s = s + u[k + m];
    s = s - u[k] * 4.0;
    s = s * invh2;
    return f[k] - s;
}

__device__ __forceinline__ double absmax_step(double acc, double v)
{
    const double a = __builtin_fabs(v);
    return (acc < a) ? a : acc;  // NaN never wins, as `maxV < v ? v : maxV` (DTDoubleArray.cpp:895)
}

__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(v, off, 64);
        v = (v < o) ? o : v;
    }
    return v;
}

// sets the calling thread's mgx_last_error() text and returns `code` (mgx_api.hip)
int api_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

struct SweepCoef {
    double h2;      // dx*dx                     (main.cpp:180)
    double invh2;   // 1.0/h2                    (main.cpp:268)
    double omega;
    double nomega;  // 1 - omega                 (main.cpp:179)
};

// ---- one kernel per reference operator (mgx_kernels.hip) -------------------
hipError_t launch_jacobi(const double *uin, double *uout, const double *f, int m, int jlo, int jhi,
                         const SweepCoef &c, hipStream_t s);
hipError_t launch_residual(const double *u, const double *f, double *r, int m, int jlo, int jhi,
                           const SweepCoef &c, hipStream_t s);
// Level transfers on slabs: gf0 / gc0 are the GLOBAL column indices of local
// column 0 of the fine / coarse array (both 0 on a whole grid).  Global coarse
// column J pairs with global fine column 2J.
// restrict: writes coarse local columns [Jlo, Jhi), rows 1..mc-2.
hipError_t launch_restrict(const double *fine, int mf, double *coarse, int mc, int Jlo, int Jhi,
                           int gf0, int gc0, hipStream_t s);
// refine: fine local columns [jlo, jhi), rows 1..mf-2; add=1: fine += P(coarse),
// add=0: fine = P(coarse).
hipError_t launch_refine(const double *coarse, int mc, double *fine, int mf, int jlo, int jhi,
                         int gf0, int gc0, int add, hipStream_t s);
hipError_t launch_direct3x3(double *v, const double *f, double h2, hipStream_t s);
// boundary entries of an m x m array: x <- x + 0.0 (what main.cpp:341 does to them)
hipError_t launch_boundary_plus_zero(double *v, int m, hipStream_t s);
// the same for a slab of ncols columns; jfix_lo / jfix_hi: local physical boundary columns or -1
hipError_t launch_slab_boundary_plus_zero(double *v, int m, int ncols, int jfix_lo, int jfix_hi, hipStream_t s);
// full multigrid over slabs: Dirichlet values of n coarse columns (local row0..row0+n) from a fine slab
hipError_t launch_slab_inject_boundary(const double *fine, int mf, int ncols_f, int fl0, double *coarse, int mc,
                                       int row0, int n, int first, int last, hipStream_t s);
// rows 0 and m-1 of an m x m grid <- rows[0..m) and rows[m..2m) (device arrays)
hipError_t launch_set_boundary_rows(double *v, int m, const double *rows, hipStream_t s);
// full multigrid: coarse boundary point k <- fine boundary point 2k (all four sides)
hipError_t launch_inject_boundary(const double *fine, int mf, double *coarse, int mc, hipStream_t s);
hipError_t launch_norminf(const double *a, size_t len, unsigned long long *slot, hipStream_t s);
hipError_t launch_residual_norm(const double *u, const double *f, int m, int jlo, int jhi, const SweepCoef &c,
                                unsigned long long *slot, hipStream_t s);

// resolves, for the current device, the address of the fixed device array that
// mgx_residual_norm_noalloc() keeps its partial results in (mgx_noalloc.hip)
void noalloc_prepare_device();

// ---- ordinary (pageable) host arrays <-> device, several threads through pinned staging
// buffers (mgx_hostio.hip).  Synchronous; work already enqueued on `order` comes first.
hipError_t host_to_device(double *dst_dev, const double *src_host, size_t count, hipStream_t order);
hipError_t device_to_host(double *dst_host, const double *src_dev, size_t count, hipStream_t order);
// every interior entry of the m x m host array is +0.0 (all bits zero); threaded, stops at the first that is not
bool host_interior_is_zero(const double *v, int m);

// ---- the LDS-resident bottom of the cycle (mgx_tail.hip) -----------------------
static constexpr int TAIL_MAX_M = 65;
static constexpr int TAIL_MAX_LEVELS = 8;
static constexpr size_t TAIL_MAX_LDS_BYTES = 160 * 1024;
struct TailArgs {
    const double *f = nullptr;  // right-hand side of the tail's top level (m[0] x m[0], pitch ld)
    double *v = nullptr;        // receives the interior of the top level's iterate after the cycle
    int ld = 0;
    int nlev = 0;               // levels in the tail, top first
    int m[TAIL_MAX_LEVELS] = {0};
    double h2[TAIL_MAX_LEVELS] = {0}, invh2[TAIL_MAX_LEVELS] = {0};
    int Ndown = 0, Nup = 0;
    double omega = 0, nomega = 0;
    // extension: device tables of the exact coarsest-grid solve (mgx_coarse.h), or null for
    // the reference's behaviour (a coarsest grid that is not 3x3 is left as it is)
    const double *exact_tables = nullptr;
    int tables_off = 0, tables_len = 0;  // filled by the launcher: position (doubles) in LDS
};
size_t tail_lds_bytes(const TailArgs &A);  // hierarchy + tables
// One V(Ndown,Nup) cycle over the tail's levels from a zero iterate on its top level.
hipError_t launch_tail(const TailArgs &A, hipStream_t s);

// ---- groups of small levels in one launch (mgx_mid.hip) ------------------------
// For levels that start the down sweep from a zero iterate, with the lean bookkeeping (no
// iterate stored on the way down, pre-smoothing recomputed on the way up).
static constexpr int MID_MAX_LEVELS = 5;
struct MidArgs {
    int nl = 0;                                   // levels in the group; local level 0 is the finest
    int m[MID_MAX_LEVELS + 1] = {0};              // rows of local levels 0..nl (nl: the level below the group)
    double h2[MID_MAX_LEVELS] = {0}, invh2[MID_MAX_LEVELS] = {0};
    const double *f[MID_MAX_LEVELS] = {nullptr};  // right-hand sides of levels 0..nl-1 (down reads f[0] only)
    double *fo[MID_MAX_LEVELS + 1] = {nullptr};   // down: right-hand sides of levels 1..nl, written
    const double *cv = nullptr;                   // up: iterate of level nl, read with its boundary
    double *v = nullptr;                          // up: iterate of level 0, interior written
    int ndown = 0, nup = 0;
    double omega = 0, nomega = 0;
    // filled by the launcher: tile a workgroup owns (rows x columns of the output level),
    // tiles per axis, doubles per LDS array
    int tile_i = 0, tile_j = 0, nti = 0, ntj = 0, stride = 0;
};
// whether a launch for these levels and sweep counts exists (odd nested sizes, tiles that fit LDS)
bool mid_fits(bool down, const MidArgs &A);
hipError_t launch_mid_down(const MidArgs &A, hipStream_t s);
hipError_t launch_mid_up(const MidArgs &A, hipStream_t s);

// ---- fused streaming kernels (mgx_stream.hip) --------------------------------
// One launch = [prolong + correct] -> ns Jacobi sweeps -> [residual -> restrict | norm].
struct StreamArgs {
    const double *uin = nullptr;  // fine iterate read (unused with zero_in)
    double *uout = nullptr;       // fine iterate written (interior of [jlo,jhi) only)
    const double *f = nullptr;    // fine right-hand side
    const double *cin = nullptr;  // pro: coarse iterate to interpolate
    double *cout = nullptr;       // res==1: coarse right-hand side written
    unsigned long long *slot = nullptr;  // res==2: encoded norm, atomic max
    int m = 0, ld = 0;            // fine rows and pitch (doubles)
    int ncols = 0;                // fine local columns that exist: 0..ncols-1
    int mc = 0, ldc = 0;          // coarse rows and pitch
    int nccols = 0;               // coarse local columns that exist
    int jlo = 0, jhi = 0;         // fine local columns to produce: [jlo, jhi)
    int jlo2 = 0, jhi2 = 0;       // and, if jhi2 > jlo2, a second range [jlo2, jhi2) with jlo2 >= jhi
    int jfix_lo = -1, jfix_hi = -1;  // local index of the physical boundary columns j=0 / j=M-1, or -1
    int gf0 = 0, gc0 = 0;         // global column index of local column 0 (fine, coarse)
    int CJ = 0, nstrips = 0, nchunks = 0, nwv = 4;  // filled by the launcher
    int nch1 = 0;                 // filled by the launcher: chunks of the first range (the rest march the second)
    int remap = 0;                // filled by the launcher: renumbering of the workgroups over the XCDs (0: none)
    SweepCoef c;
};
static constexpr int STREAM_MAX_SWEEPS = 4;
// What one launch computes: ns sweeps; the coarse-grid correction added to
// stage pa (-1: none); the residual norm of stage nm merged into the slot
// (-1: none); res: the residual of stage ns restricted into cout; zero_in: the
// input iterate is all zeros and is not read; store: stage ns is written to uout.
struct StreamOp {
    int ns = 0, pa = -1, nm = -1, res = 0, zero_in = 0, store = 1;
};
hipError_t launch_stream(const StreamOp &op, const StreamArgs &A, hipStream_t s);
// While set (per host thread), a launch whose geometry has no measured shape yet runs every
// candidate shape on the spot (one untimed + three timed launches each, one stream
// synchronisation per candidate) and keeps the fastest.  Returns the previous setting.
bool set_tune_now(bool on);
// Forget every measured launch shape of this process (the next launches measure again).
void tune_forget();
// The shapes measured so far in this process, one text line per geometry.
std::string tune_report();
// halo columns (and rows) the pipeline of `op` consumes beyond the produced range
int stream_halo(const StreamOp &op);
// The four-flag form of the simple legs: pro in {0,1}; res 0 none, 1 restrict,
// 2 norm of the result.
inline StreamOp simple_op(int pro, int ns, int res, int zero_in)
{
    StreamOp op;
    op.ns = ns;
    op.pa = pro ? 0 : -1;
    op.nm = res == 2 ? ns : -1;
    op.res = res == 1 ? 1 : 0;
    op.zero_in = zero_in;
    op.store = (ns > 0 || pro) ? 1 : 0;
    return op;
}
inline hipError_t launch_stream(int pro, int ns, int res, int zero_in, const StreamArgs &A, hipStream_t s)
{
    return launch_stream(simple_op(pro, ns, res, zero_in), A, s);
}

} // namespace mgx
