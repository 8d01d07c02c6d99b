// mgx_api.hip -- the C ABI declared in include/mgx.h: argument checking, the
// device-resident level hierarchy ("plan") and the V-cycle schedule of the
// reference's MultiGrid() (main.cpp:283-357).  Kernels live in
// mgx_kernels.hip (one per reference operator) and mgx_stream.hip (fused).
#include "../../include/mgx.h"
#include "mgx_coarse.h"
#include "mgx_internal.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <future>
#include <new>
#include <utility>
#include <vector>

using namespace mgx;

namespace {

thread_local char g_err[512] = "";

void set_error(const char *fmt, va_list ap) { vsnprintf(g_err, sizeof(g_err), fmt, ap); }

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    set_error(fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(MGX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

SweepCoef make_coef(double dx, double omega)
{
    SweepCoef c;
    c.h2 = dx * dx;          // main.cpp:180
    c.invh2 = 1.0 / c.h2;    // main.cpp:268
    c.omega = omega;
    c.nomega = 1 - omega;    // main.cpp:179
    return c;
}

int check_grid(int m, const char *who)
{
    if (m < 3) return fail(MGX_ERR_ARG, "%s: grid size %d is below 3", who, m);
    if ((m % 2) == 0) return fail(MGX_ERR_ARG, "%s: Input has even rows and columns! (m=%d)", who, m);
    return MGX_OK;
}

struct Level {
    int m = 0;
    double dx = 0;
    double *f = nullptr;
    double *v[2] = {nullptr, nullptr};  // ping-pong pair; v[cur] is the live iterate
    int cur = 0;
    size_t n() const { return (size_t)m * (size_t)m; }
};

} // namespace

int mgx::api_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    set_error(fmt, ap);
    va_end(ap);
    return code;
}

struct mgx_plan {
    int device = 0;
    hipStream_t stream = nullptr;
    int schedule = MGX_SCHEDULE_FUSED;
    int verbose = 0;
    int coarsest = 2;
    int depth = 0;
    int coarse_solve = MGX_COARSE_REFERENCE;
    double *exact_tables = nullptr;      // device tables of the exact coarsest-grid solve, or null
    int use_tail = 1;  // levels <= 65x65 of a cycle in one LDS-resident launch (MGX_NO_TAIL=1 disables)
    // groups of levels of at most mid_max_m rows in one launch per direction (mgx_mid.hip;
    // MGX_NO_MID=1 disables, MGX_MID_MAX_M / MGX_MID_DOWN_NL / MGX_MID_UP_NL size the groups)
    int use_mid = 1, mid_max_m = 513, mid_nl_down = 2, mid_nl_up = 4;
    int mid_single_down_m = 1025, mid_single_up_m = 1025;  // MGX_MID_SINGLE_DOWN / MGX_MID_SINGLE_UP
    int tail_small_m = 33;  // a tail level above this size goes to a grouped launch where one exists (MGX_TAIL_SMALL_M)
    std::vector<Level> lv;
    std::vector<void *> allocs;          // what hipMalloc returned (the level arrays may start inside)
    double *r = nullptr;                 // residual scratch of the stepwise schedule (fine size)
    unsigned long long *slots = nullptr; // encoded norms, one per cycle (+1)
    int nslots = 0;
    std::vector<hipEvent_t> events;
    size_t bytes = 0;
    // per-launch timing of the fine-level kernels (mgx_plan_profile)
    int profile = 0;
    struct ProfRec { int kid, level; hipEvent_t a, b; };
    std::vector<ProfRec> recs;
    std::vector<double> lvl_ms;  // profile mode 2: device time of the launches that START at each level
    std::vector<long> lvl_n;
    std::vector<hipEvent_t> evpool;
    size_t evused = 0;
    double prof_ms[MGX_KERNEL_COUNT] = {0};
    long prof_n[MGX_KERNEL_COUNT] = {0};
    // stepwise schedule: event pairs around every relax() of a cycle (Times as main.cpp:320-322,342-344)
    struct RelaxRec { int cycle; hipEvent_t a, b; };
    std::vector<RelaxRec> relax_recs;
    int relax_cycle = -1;  // >= 0 while mgx_plan_cycles() wants relax() bracketed
    // Carry between calls (fused lean schedule, Ndown >= 1).  A caller that runs one cycle per
    // call would pay two level-0 launches per cycle: an opening down leg and a closing up leg,
    // which a longer call fuses into one launch per cycle.  So a call may end with the fused
    // up+down launch too, which leaves the iterate after the NEXT cycle's down sweeps and that
    // cycle's coarse right-hand side; the next call with the same Ndown, Nup and omega, and no
    // change of the data in between, starts from there.  The end-of-cycle iterate itself (what
    // mgx_plan_v_dev() names) is then not in memory: it is `lazy`, and materialize() produces
    // it when somebody asks -- one plain up leg from the closing launch's input, which is kept,
    // into the array `r` that the fused schedule does not use otherwise.  Ending a call this way
    // is a bet of one level-0 launch on the next call being a continuation.  It is placed when
    // the two calls before it were continuations, inside mgx_plan_cycles_until, or when the
    // call is short (at most HABIT_MAX_CYCLES cycles) and the plan's caller has the `habit`: a
    // count that goes up (to 3) whenever a carried leg is consumed and down whenever one left by
    // a short call is dropped unused; the bet is placed from 2 on.  A series of short calls
    // thus pays three extra launches the first time and none on the later right-hand sides of
    // the same plan, and a caller who stops calling that way loses two bets at most.
    bool carry_capable = true;  // false for the views full multigrid cycles on
    bool lazy = false;          // v[0] = swept iterate, v[1] = the closing launch's input, result not computed
    bool carry_valid = false;   // (not lazy) v[0] = result, r = swept iterate; lv[1].f its restricted residual
    bool force_carry = false;
    bool last_valid = false;    // the previous call's parameters are below and the data is as it left it
    int last_Ndown = 0, last_Nup = 0, streak = 0;
    int habit = 0;              // 0..3, see above; lives as long as the plan
    bool carry_short = false;   // what is carried was left by a short call
    static constexpr int HABIT_MAX_CYCLES = 4;
    double last_omega = 0;
    unsigned long long carry_key = 0;  // encoded norm after the previous call's last cycle
};

namespace {

// A plan lives on the device it was created on; its entry points run there whatever
// device the calling thread has current, and put the caller's device back on return.
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    explicit DeviceScope(const mgx_plan *p)
    {
        if (p && hipGetDevice(&prev) == hipSuccess && prev != p->device) switched = hipSetDevice(p->device) == hipSuccess;
    }
    ~DeviceScope()
    {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};

// `shift` bytes (a multiple of 256) are skipped at the start of the allocation: the arrays of a
// level can be placed at different phases of the memory's channel interleave
int plan_alloc(mgx_plan *p, void **ptr, size_t nbytes, size_t shift = 0)
{
    void *raw = nullptr;
    HIP_TRY(hipMalloc(&raw, nbytes + shift));
    p->allocs.push_back(raw);
    *ptr = (char *)raw + shift;
    p->bytes += nbytes + shift;
    return MGX_OK;
}

int ensure_slots(mgx_plan *p, int n)
{
    if (n <= p->nslots) return MGX_OK;
    if (p->slots) HIP_TRY(hipFree(p->slots));
    p->slots = nullptr;
    HIP_TRY(hipMalloc((void **)&p->slots, sizeof(unsigned long long) * (size_t)n));
    p->nslots = n;
    return MGX_OK;
}

int ensure_events(mgx_plan *p, int n)
{
    while ((int)p->events.size() < n) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        p->events.push_back(e);
    }
    return MGX_OK;
}

hipEvent_t prof_event(mgx_plan *p)
{
    if (p->evused == p->evpool.size()) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        p->evpool.push_back(e);
    }
    return p->evpool[p->evused++];
}

// Launch `call`; when profiling is on and the launch is on the fine level,
// bracket it with two events on the plan's stream.
#define LAUNCH(p, level, kid, call)                                              \
    do {                                                                         \
        const bool prof_ = ((p)->profile && (level) == 0) || (p)->profile == 2;  \
        hipEvent_t ea_ = nullptr, eb_ = nullptr;                                 \
        if (prof_ && (ea_ = prof_event(p))) HIP_TRY(hipEventRecord(ea_, (p)->stream)); \
        HIP_TRY(call);                                                           \
        if (prof_ && ea_ && (eb_ = prof_event(p))) {                             \
            HIP_TRY(hipEventRecord(eb_, (p)->stream));                           \
            (p)->recs.push_back({(kid), (level), ea_, eb_});                     \
        }                                                                        \
    } while (0)

int prof_collect(mgx_plan *p)
{
    for (const mgx_plan::ProfRec &r : p->recs) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, r.a, r.b));
        if (r.level == 0 && r.kid >= 0) {
            p->prof_ms[r.kid] += (double)ms;
            p->prof_n[r.kid] += 1;
        }
        if (p->profile == 2 && r.level >= 0 && r.level < (int)p->lvl_ms.size()) {
            p->lvl_ms[(size_t)r.level] += (double)ms;
            p->lvl_n[(size_t)r.level] += 1;
        }
    }
    p->recs.clear();
    p->evused = 0;
    return MGX_OK;
}

// niter Jacobi sweeps on one level, ping-ponging its two v buffers.
int level_relax(mgx_plan *p, int d, int niter, double omega)
{
    Level &L = p->lv[d];
    const SweepCoef c = make_coef(L.dx, omega);
    for (int it = 0; it < niter; it++) {
        LAUNCH(p, d, MGX_KERNEL_JACOBI, launch_jacobi(L.v[L.cur], L.v[L.cur ^ 1], L.f, L.m, 1, L.m - 1, c, p->stream));
        L.cur ^= 1;
    }
    return MGX_OK;
}

// relax() as the V-cycle loop calls it (main.cpp:320-322, 342-344): inside mgx_plan_cycles()
// the launches are bracketed by an event pair, the device-time counterpart of the
// reference's timer.Start()/Stop() around relax only.
int timed_relax(mgx_plan *p, int d, int niter, double omega)
{
    if (p->relax_cycle < 0 || niter <= 0) return level_relax(p, d, niter, omega);
    hipEvent_t a = prof_event(p), b = prof_event(p);
    if (!a || !b) return level_relax(p, d, niter, omega);
    HIP_TRY(hipEventRecord(a, p->stream));
    int rc = level_relax(p, d, niter, omega);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(b, p->stream));
    p->relax_recs.push_back({p->relax_cycle, a, b});
    return MGX_OK;
}

// main.cpp:331: the coarsest grid.  3x3: the reference's closed form; any other size:
// nothing (the reference prints a message, main.cpp:167), or with MGX_COARSE_EXACT the
// exact solve (an extension).
int coarsest_solve(mgx_plan *p)
{
    Level &B = p->lv[p->depth];
    if (B.m == 3) {
        HIP_TRY(launch_direct3x3(B.v[B.cur], B.f, B.dx * B.dx, p->stream));
    } else if (p->exact_tables) {
        HIP_TRY(launch_coarse_exact(B.v[B.cur], B.f, B.m, B.dx * B.dx, p->exact_tables, p->stream));
    } else if (p->verbose) {
        printf("Error: Direct solver not defined for %dx%d matrices!\n", B.m, B.m);
    }
    return MGX_OK;
}

// main.cpp:341 adds an array with a +0.0 boundary to the fine iterate: its Dirichlet values
// stay as they are, except that -0.0 turns into +0.0.  Both ping-pong buffers carry them.
int normalize_boundary(mgx_plan *p)
{
    Level &L = p->lv[0];
    HIP_TRY(launch_boundary_plus_zero(L.v[0], L.m, p->stream));
    HIP_TRY(launch_boundary_plus_zero(L.v[1], L.m, p->stream));
    // the third array a call's closing launch may write the iterate to (see `carry_valid`)
    if (p->carry_capable && p->schedule == MGX_SCHEDULE_FUSED) HIP_TRY(launch_boundary_plus_zero(p->r, L.m, p->stream));
    return MGX_OK;
}

// the four boundary lines of an m x m array from another (columns 0 and m-1 are contiguous,
// rows 0 and m-1 one double per column)
int copy_boundary(double *dst, const double *src, int m, hipStream_t s)
{
    const size_t col = sizeof(double) * (size_t)m;
    HIP_TRY(hipMemcpyAsync(dst, src, col, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpyAsync(dst + (size_t)(m - 1) * m, src + (size_t)(m - 1) * m, col, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpy2DAsync(dst, col, src, col, sizeof(double), (size_t)m, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hipMemcpy2DAsync(dst + (m - 1), col, src + (m - 1), col, sizeof(double), (size_t)m, hipMemcpyDeviceToDevice, s));
    return MGX_OK;
}

struct Level;
StreamArgs level_args(const Level &L, const Level *C, double omega);

// The end-of-cycle iterate of a call that ended with the fused up+down launch: the plain up leg
// (refine + add + Nup sweeps, main.cpp:337-344) from that launch's input and the level-1
// correction, both still in place, into `r`; afterwards v[0] names it and `r` the swept iterate.
int materialize(mgx_plan *p)
{
    if (!p->lazy) return MGX_OK;
    Level &L = p->lv[0];
    Level &C = p->lv[1];
    StreamArgs A = level_args(L, &C, p->last_omega);
    A.cin = C.v[C.cur];
    A.uin = L.v[1];
    A.uout = p->r;
    StreamOp op;
    op.ns = p->last_Nup;
    op.pa = 0;
    HIP_TRY(launch_stream(op, A, p->stream));
    double *result = p->r;
    p->r = L.v[0];
    L.v[0] = result;
    L.cur = 0;
    p->lazy = false;
    p->carry_valid = true;
    return MGX_OK;
}

// The data is about to change (or to be used in another way): nothing computed ahead survives.
// keep_iterate: the caller still needs the current iterate in v[0].
int forget_carry(mgx_plan *p, bool keep_iterate = true)
{
    int rc = MGX_OK;
    if ((p->lazy || p->carry_valid) && p->carry_short && p->habit > 0) p->habit--;  // a bet lost
    if (p->lazy && keep_iterate) rc = materialize(p);
    p->lazy = false;
    p->carry_valid = false;
    p->last_valid = false;
    p->streak = 0;
    return rc;
}

// One V(Ndown,Nup) cycle, one kernel per reference operator:
// the body of the loop at main.cpp:317-346.
int vcycle_stepwise(mgx_plan *p, int Ndown, int Nup, double omega)
{
    const int depth = p->depth;
    hipStream_t s = p->stream;
    for (int d = 0; d < depth; d++) {  // main.cpp:317-328
        Level &L = p->lv[d];
        Level &C = p->lv[d + 1];
        int rc = timed_relax(p, d, Ndown, omega);
        if (rc) return rc;
        const SweepCoef c = make_coef(L.dx, omega);
        // residual(): boundary columns 0 and m-1 are zero, the kernel fills the rest
        HIP_TRY(hipMemsetAsync(p->r, 0, sizeof(double) * (size_t)L.m, s));
        HIP_TRY(hipMemsetAsync(p->r + (size_t)(L.m - 1) * L.m, 0, sizeof(double) * (size_t)L.m, s));
        LAUNCH(p, d, MGX_KERNEL_RESIDUAL, launch_residual(L.v[L.cur], L.f, p->r, L.m, 1, L.m - 1, c, s));
        // coarsen(): interior of the coarse f; its boundary stays 0 from allocation
        LAUNCH(p, d, MGX_KERNEL_RESTRICT, launch_restrict(p->r, L.m, C.f, C.m, 1, C.m - 1, 0, 0, s));
    }
    {
        int rc = coarsest_solve(p);
        if (rc) return rc;
    }
    for (int d = depth - 1; d >= 0; d--) {  // main.cpp:335-346
        Level &L = p->lv[d];
        Level &C = p->lv[d + 1];
        LAUNCH(p, d, MGX_KERNEL_PROLONG_ADD, launch_refine(C.v[C.cur], C.m, L.v[L.cur], L.m, 1, L.m - 1, 0, 0, 1, s));
        int rc = d == 0 ? normalize_boundary(p) : MGX_OK;
        if (rc) return rc;
        rc = timed_relax(p, d, Nup, omega);
        if (rc) return rc;
        HIP_TRY(hipMemsetAsync(C.v[C.cur], 0, sizeof(double) * C.n(), s));  // main.cpp:345
    }
    return MGX_OK;
}


StreamArgs level_args(const Level &L, const Level *C, double omega)
{
    StreamArgs A;
    A.f = L.f;
    A.m = L.m;
    A.ld = L.m;
    A.ncols = L.m;
    A.jlo = 1;
    A.jhi = L.m - 1;
    A.jfix_lo = 0;
    A.jfix_hi = L.m - 1;
    A.c = make_coef(L.dx, omega);
    if (C) {
        A.mc = C->m;
        A.ldc = C->m;
        A.nccols = C->m;
    }
    return A;
}

// [prolong+correct] -> ns sweeps -> [residual -> restrict | norm] on one level
// as a chain of fused launches of at most STREAM_MAX_SWEEPS sweeps each.
// vbuf[cur] is the input iterate; every launch that writes the iterate writes
// the other buffer and flips cur.  p may be null (no profiling).
int stream_chain(mgx_plan *p, int level, int kid, hipStream_t s, int pro, int ns, int res, int zero_in,
                 StreamArgs A, double *vbuf[2], int &cur)
{
    int remaining = ns;
    bool first = true;
    do {
        const int k = remaining < STREAM_MAX_SWEEPS ? remaining : STREAM_MAX_SWEEPS;
        remaining -= k;
        const bool last = remaining == 0;
        const int pr = first ? pro : 0;
        const int rs = last ? res : 0;
        const int zi = first ? zero_in : 0;
        A.uin = vbuf[cur];
        A.uout = vbuf[cur ^ 1];
        if (k > 0 || pr || rs) {
            if (p) {
                LAUNCH(p, level, kid, launch_stream(pr, k, rs, zi, A, s));
            } else {
                HIP_TRY(launch_stream(pr, k, rs, zi, A, s));
            }
            if (k > 0 || pr) cur ^= 1;
        }
        first = false;
    } while (remaining > 0);
    return MGX_OK;
}

// ---- fused schedule ---------------------------------------------------------
// "lean" (Ndown + Nup <= STREAM_MAX_SWEEPS, the usual V(1..2,1..2)):
//   * a level that starts the down sweep from a zero iterate (every level >= 1,
//     main.cpp:295-299,345) stores nothing on the way down (f -> coarse f) and
//     its up leg recomputes the Ndown sweeps from f before adding the coarse
//     correction: 10 + 18 bytes per point instead of 18 + 26;
//   * on level 0 the up leg of cycle k and the down leg of cycle k+1 are ONE
//     launch (correction -> Nup sweeps -> norm -> Ndown sweeps -> residual ->
//     restriction): 28 bytes per point per cycle instead of 52.
// Otherwise every leg is a chain of launches of at most STREAM_MAX_SWEEPS sweeps.
struct FusedCtx {
    int Ndown, Nup;
    double omega;
    bool zero0;  // level 0 itself starts from a zero iterate (the tail of a multi-GPU run)
    bool lean() const { return Ndown + Nup <= STREAM_MAX_SWEEPS; }
    bool zero_start(int d) const { return d >= 1 || zero0; }
};

int launch_leg(mgx_plan *p, int d, int kid, const StreamOp &op, const StreamArgs &A)
{
    LAUNCH(p, d, kid, launch_stream(op, A, p->stream));
    return MGX_OK;
}

// down leg of level d: relax(Ndown) + residual + coarsen (main.cpp:321-327); with norm0
// (level 0, first cycle of a call, Ndown <= STREAM_MAX_SWEEPS) the same pass also yields the
// norm of the residual of the iterate it starts from (main.cpp:307)
int fused_down(mgx_plan *p, int d, const FusedCtx &cx, unsigned long long *norm0 = nullptr)
{
    Level &L = p->lv[d];
    Level &C = p->lv[d + 1];
    StreamArgs A = level_args(L, &C, cx.omega);
    A.cout = C.f;
    if (norm0) {
        StreamOp op;
        op.ns = cx.Ndown;
        op.nm = 0;
        op.res = 1;
        op.store = cx.Ndown > 0 ? 1 : 0;
        A.slot = norm0;
        A.uin = L.v[L.cur];
        A.uout = L.v[L.cur ^ 1];
        if (op.store) L.cur ^= 1;
        return launch_leg(p, d, MGX_KERNEL_FUSED_DOWN, op, A);
    }
    if (cx.zero_start(d)) {
        if (cx.lean()) {  // nothing stored: the up leg recomputes these sweeps
            StreamOp op;
            op.ns = cx.Ndown;
            op.res = 1;
            op.zero_in = 1;
            op.store = 0;
            return launch_leg(p, d, MGX_KERNEL_FUSED_DOWN, op, A);
        }
        if (cx.Ndown == 0) {  // residual of the zero iterate; nothing to store either
            StreamOp op;
            op.res = 1;
            op.zero_in = 1;
            op.store = 0;
            return launch_leg(p, d, MGX_KERNEL_FUSED_DOWN, op, A);
        }
        return stream_chain(p, d, MGX_KERNEL_FUSED_DOWN, p->stream, 0, cx.Ndown, 1, 1, A, L.v, L.cur);
    }
    return stream_chain(p, d, MGX_KERNEL_FUSED_DOWN, p->stream, 0, cx.Ndown, 1, 0, A, L.v, L.cur);
}

// up leg of level d: refine + add + relax(Nup) (main.cpp:337-344); with a slot
// also the residual norm of the result (main.cpp:351); with next_down (level 0,
// lean) also the down leg of the following cycle.
int fused_up(mgx_plan *p, int d, const FusedCtx &cx, unsigned long long *slot, bool next_down)
{
    Level &L = p->lv[d];
    Level &C = p->lv[d + 1];
    StreamArgs A = level_args(L, &C, cx.omega);
    A.cin = C.v[C.cur];
    A.slot = slot;
    if (cx.zero_start(d)) {
        if (cx.lean()) {  // sweeps from zero recomputed, correction added after Ndown of them
            StreamOp op;
            op.ns = cx.Ndown + cx.Nup;
            op.pa = cx.Ndown;
            op.nm = slot ? op.ns : -1;
            op.zero_in = 1;
            if (slot && cx.Ndown > 0)
                return fail(MGX_ERR_STATE, "fused_up: no norm on a recomputing leg");
            A.uout = L.v[0];
            L.cur = 0;
            return launch_leg(p, d, MGX_KERNEL_FUSED_UP, op, A);
        }
        if (cx.Ndown == 0)  // the iterate is still zero: not read
            return stream_chain(p, d, MGX_KERNEL_FUSED_UP, p->stream, 1, cx.Nup, slot ? 2 : 0, 1, A, L.v, L.cur);
    }
    if (next_down) {
        A.cout = C.f;
        StreamOp op;
        op.ns = cx.Nup + cx.Ndown;
        op.pa = 0;
        op.nm = cx.Nup;
        op.res = 1;
        A.uin = L.v[L.cur];
        A.uout = L.v[L.cur ^ 1];
        L.cur ^= 1;
        return launch_leg(p, d, MGX_KERNEL_FUSED_UPDOWN, op, A);
    }
    return stream_chain(p, d, MGX_KERNEL_FUSED_UP, p->stream, 1, cx.Nup, slot ? 2 : 0, 0, A, L.v, L.cur);
}

MidArgs mid_args(const mgx_plan *p, int d, int nl, const FusedCtx &cx);

// First level of the LDS-resident tail: the first zero-start level of at most
// TAIL_MAX_M points per side (level 0 only when it starts from zero), or -1.
// One workgroup spends 10 us of the tail's 25 on a 65x65 level, so where the grouped
// launches (mgx_mid.hip) can take that level along with the one above it, at next to no
// cost, the tail starts one level further down.
int tail_first_level(const mgx_plan *p, const FusedCtx &cx)
{
    if (!p->use_tail) return -1;
    const int first = cx.zero0 ? 0 : 1;
    for (int d = first; d <= p->depth; d++) {
        if (p->lv[d].m > TAIL_MAX_M) continue;
        if (p->depth - d + 1 > TAIL_MAX_LEVELS) return -1;
        size_t n = 0;
        for (int k = d; k <= p->depth; k++) n += 3 * p->lv[k].n() * sizeof(double);
        if (p->exact_tables) {
            const size_t nc = (size_t)p->lv[p->depth].m - 2;
            n += (nc * nc + nc) * sizeof(double);
        }
        if (n > TAIL_MAX_LDS_BYTES) return -1;
        if (p->use_mid && cx.lean() && d - 1 >= first && d < p->depth && p->lv[d].m > p->tail_small_m &&
            p->lv[d - 1].m <= p->mid_max_m && p->mid_nl_down >= 2 && p->mid_nl_up >= 2 &&
            mid_fits(true, mid_args(p, d - 1, 2, cx)) && mid_fits(false, mid_args(p, d - 1, 2, cx)))
            return d + 1;
        return d;
    }
    return -1;
}

// Levels T..depth of one cycle in a single launch; the result is lv[T].v[0].
int fused_tail(mgx_plan *p, int T, const FusedCtx &cx)
{
    TailArgs A;
    Level &L = p->lv[T];
    A.f = L.f;
    A.v = L.v[0];
    A.ld = L.m;
    A.nlev = p->depth - T + 1;
    for (int k = 0; k < A.nlev; k++) {
        const Level &K = p->lv[T + k];
        const SweepCoef c = make_coef(K.dx, cx.omega);
        A.m[k] = K.m;
        A.h2[k] = c.h2;
        A.invh2[k] = c.invh2;
    }
    A.Ndown = cx.Ndown;
    A.Nup = cx.Nup;
    A.omega = cx.omega;
    A.nomega = 1 - cx.omega;
    A.exact_tables = p->exact_tables;
    L.cur = 0;
    if (p->profile == 2) {
        LAUNCH(p, T, -1, launch_tail(A, p->stream));
    } else {
        HIP_TRY(launch_tail(A, p->stream));
    }
    if (p->verbose && p->lv[p->depth].m != 3 && !p->exact_tables)
        printf("Error: Direct solver not defined for %dx%d matrices!\n", p->lv[p->depth].m, p->lv[p->depth].m);
    return MGX_OK;
}

// ---- groups of small levels in one launch (mgx_mid.hip) -----------------------
MidArgs mid_args(const mgx_plan *p, int d, int nl, const FusedCtx &cx)
{
    MidArgs A;
    A.nl = nl;
    for (int k = 0; k <= nl; k++) A.m[k] = p->lv[d + k].m;
    for (int k = 0; k < nl; k++) {
        const SweepCoef c = make_coef(p->lv[d + k].dx, cx.omega);
        A.h2[k] = c.h2;
        A.invh2[k] = c.invh2;
        A.f[k] = p->lv[d + k].f;
    }
    for (int k = 1; k <= nl; k++) A.fo[k] = p->lv[d + k].f;
    A.ndown = cx.Ndown;
    A.nup = cx.Nup;
    A.omega = cx.omega;
    A.nomega = 1 - cx.omega;
    return A;
}

// How many of the levels d..last one launch takes, starting at d; 1: level d runs its own
// streaming leg.  Groups are cut so that a remainder stays at the fine end, where a
// level is big enough for the streaming kernel.
int mid_group(const mgx_plan *p, const FusedCtx &cx, int d, int last, bool down)
{
    if (!p->use_mid || !cx.lean() || !cx.zero_start(d) || p->lv[d].m > p->mid_max_m) return 1;
    const int cap = down ? p->mid_nl_down : p->mid_nl_up;
    const int n = last - d + 1;
    if (cap < 2 || n < 2) return 1;
    for (int g = (n % cap == 0) ? cap : n % cap; g >= 2; g--)
        if (mid_fits(down, mid_args(p, d, g, cx))) return g;
    return 1;
}

// A level that joins no group still gains from the tiled kernel while it is small enough
// for latency, not bandwidth, to set the time of its streaming leg (measured: 1025^2 down
// 9.8 -> 7.3 us, up 14.5 -> 11.1 us; at 2049^2 the streaming kernel wins both ways).
bool mid_single(const mgx_plan *p, const FusedCtx &cx, int d, bool down)
{
    const int most = down ? p->mid_single_down_m : p->mid_single_up_m;
    return p->use_mid && cx.lean() && cx.zero_start(d) && p->lv[d].m <= most && mid_fits(down, mid_args(p, d, 1, cx));
}

// down legs of levels d..d+nl-1 (main.cpp:321-327): lv[d].f -> lv[d+1..d+nl].f
int mid_down(mgx_plan *p, int d, int nl, const FusedCtx &cx)
{
    const MidArgs A = mid_args(p, d, nl, cx);
    LAUNCH(p, d, MGX_KERNEL_FUSED_DOWN, launch_mid_down(A, p->stream));
    return MGX_OK;
}

// up legs of levels d+nl-1..d (main.cpp:337-344): iterate of level d+nl -> lv[d].v[0]
int mid_up(mgx_plan *p, int d, int nl, const FusedCtx &cx)
{
    MidArgs A = mid_args(p, d, nl, cx);
    const Level &C = p->lv[d + nl];
    A.cv = C.v[C.cur];
    A.v = p->lv[d].v[0];
    p->lv[d].cur = 0;
    LAUNCH(p, d, MGX_KERNEL_FUSED_UP, launch_mid_up(A, p->stream));
    return MGX_OK;
}

// levels first..depth-1 down, the coarsest solve, levels depth-1..first up; every one of them
// starts from a zero iterate (first = 1, or 0 for a plan whose top level does too)
int fused_coarse_legs(mgx_plan *p, const FusedCtx &cx, int first = 1)
{
    const int depth = p->depth;
    const int T = tail_first_level(p, cx);
    const int last = (T >= 1) ? T - 1 : depth - 1;  // last level that runs its own legs
    for (int d = first; d <= last;) {
        const int g = mid_group(p, cx, d, last, true);
        const int rc = (g >= 2 || mid_single(p, cx, d, true)) ? mid_down(p, d, g, cx) : fused_down(p, d, cx);
        if (rc) return rc;
        d += g;
    }
    if (T >= 1) {
        int rc = fused_tail(p, T, cx);
        if (rc) return rc;
    } else {
        int rc = coarsest_solve(p);
        if (rc) return rc;
    }
    // the up legs: groups cut from the fine end like the down legs, run from the coarse end
    std::vector<std::pair<int, int>> groups;
    for (int d = first; d <= last;) {
        const int g = mid_group(p, cx, d, last, false);
        groups.push_back({d, g});
        d += g;
    }
    for (size_t q = groups.size(); q-- > 0;) {
        const int d = groups[q].first, g = groups[q].second;
        const int rc = (g >= 2 || mid_single(p, cx, d, false)) ? mid_up(p, d, g, cx) : fused_up(p, d, cx, nullptr, false);
        if (rc) return rc;
    }
    return MGX_OK;
}

// One whole V(Ndown,Nup) cycle (no cross-cycle fusion): used by mgx_plan_vcycle_async.
int vcycle_fused(mgx_plan *p, int Ndown, int Nup, double omega, unsigned long long *norm_slot, bool zero0 = false)
{
    FusedCtx cx{Ndown, Nup, omega, zero0};
    if (zero0 && tail_first_level(p, cx) == 0) return fused_tail(p, 0, cx);  // the whole plan fits the tail
    if (p->depth == 0) return coarsest_solve(p);
    if (zero0 && !norm_slot) return fused_coarse_legs(p, cx, 0);  // level 0 is one of the zero-start levels
    int rc = fused_down(p, 0, cx);
    if (!rc) rc = fused_coarse_legs(p, cx);
    if (!rc && !zero0) rc = normalize_boundary(p);
    if (!rc) rc = fused_up(p, 0, cx, (zero0 && cx.lean() && Ndown > 0) ? nullptr : norm_slot, false);
    return rc;
}

} // namespace

extern "C" {

int mgx_version(void) { return MGX_VERSION; }

const char *mgx_last_error(void) { return g_err; }

int mgx_depth(int m, int coarsest)
{
    // main.cpp:286, float arithmetic on purpose.
    const float ratio = 1.0f * (float)(m - 1) / (float)coarsest;
    return (int)(log2f(ratio) + 0.5f);
}

static int plan_create_on_device(mgx_plan **out, int m, double dx, int coarsest, const mgx_options *opt);

int mgx_plan_create(mgx_plan **out, int m, double dx, int coarsest, const mgx_options *opt)
{
    // options.device selects where the plan lives; the caller's current device is left as it was
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    const int rc = plan_create_on_device(out, m, dx, coarsest, opt);
    int now = -1;
    if (prev >= 0 && hipGetDevice(&now) == hipSuccess && now != prev) (void)hipSetDevice(prev);
    return rc;
}

static int plan_create_on_device(mgx_plan **out, int m, double dx, int coarsest, const mgx_options *opt)
{
    if (!out) return fail(MGX_ERR_ARG, "mgx_plan_create: null output pointer");
    *out = nullptr;
    int rc = check_grid(m, "mgx_plan_create");
    if (rc) return rc;
    if (coarsest < 1) return fail(MGX_ERR_ARG, "mgx_plan_create: coarsest must be >= 1 (got %d)", coarsest);
    if (!(dx > 0.0)) return fail(MGX_ERR_ARG, "mgx_plan_create: dx must be positive");
    const int depth = mgx_depth(m, coarsest);
    if (depth < 0) return fail(MGX_ERR_HIERARCHY, "mgx_plan_create: depth %d < 0 for m=%d coarsest=%d", depth, m, coarsest);
    std::vector<int> dims(depth + 1);
    dims[0] = m;
    for (int d = 1; d <= depth; d++) {
        dims[d] = (dims[d - 1] - 1) / 2 + 1;  // main.cpp:293
        // refine() between level d and d+1 needs m_d = 2 m_{d+1} - 1, i.e. every level above
        // the coarsest odd; the coarsest itself may be even
        if (dims[d] < 3 || ((dims[d] % 2) == 0 && d < depth))
            return fail(MGX_ERR_HIERARCHY,
                        "mgx_plan_create: level %d would be %dx%d (needs odd >= 3): the reference indexes out of "
                        "range there", d, dims[d], dims[d]);
    }

    mgx_plan *p = new (std::nothrow) mgx_plan();
    if (!p) return fail(MGX_ERR_NOMEM, "mgx_plan_create: out of host memory");
    if (opt) {
        if (opt->struct_size != (int)sizeof(mgx_options)) {
            delete p;
            return fail(MGX_ERR_ARG, "mgx_plan_create: options.struct_size %d != %d", opt->struct_size,
                        (int)sizeof(mgx_options));
        }
        p->stream = (hipStream_t)opt->stream;
        p->verbose = opt->verbose;
        p->schedule = opt->schedule == MGX_SCHEDULE_DEFAULT ? MGX_SCHEDULE_FUSED : opt->schedule;
        p->coarse_solve = opt->coarse_solve;
        if (opt->device >= 0) {
            hipError_t e = hipSetDevice(opt->device);
            if (e != hipSuccess) {
                delete p;
                return fail(MGX_ERR_HIP, "hipSetDevice(%d): %s", opt->device, hipGetErrorString(e));
            }
        }
    }
    if (p->schedule != MGX_SCHEDULE_STEPWISE && p->schedule != MGX_SCHEDULE_FUSED) {
        delete p;
        return fail(MGX_ERR_ARG, "mgx_plan_create: schedule %d is not available", opt ? opt->schedule : -1);
    }
    if (p->coarse_solve != MGX_COARSE_REFERENCE && p->coarse_solve != MGX_COARSE_EXACT) {
        delete p;
        return fail(MGX_ERR_ARG, "mgx_plan_create: coarse_solve %d is not available", opt ? opt->coarse_solve : -1);
    }
    if (p->coarse_solve == MGX_COARSE_EXACT && dims[depth] != 3 && dims[depth] > COARSE_EXACT_MAX_M) {
        delete p;
        return fail(MGX_ERR_ARG, "mgx_plan_create: the exact coarsest-grid solve takes at most %dx%d points, the "
                                 "coarsest level here has %dx%d (raise the depth with a smaller `coarsest`)",
                    COARSE_EXACT_MAX_M, COARSE_EXACT_MAX_M, dims[depth], dims[depth]);
    }
    {
        hipError_t e = hipGetDevice(&p->device);
        if (e != hipSuccess) {
            delete p;
            return fail(MGX_ERR_HIP, "hipGetDevice: %s (is a GPU visible?)", hipGetErrorString(e));
        }
    }
    noalloc_prepare_device();  // (see mgx_residual_norm_noalloc: its first call may come under stream capture)
    p->coarsest = coarsest;
    p->depth = depth;
    p->use_tail = getenv("MGX_NO_TAIL") ? 0 : 1;
    p->use_mid = getenv("MGX_NO_MID") ? 0 : 1;
    if (const char *e = getenv("MGX_MID_MAX_M")) p->mid_max_m = atoi(e);
    if (const char *e = getenv("MGX_TAIL_SMALL_M")) p->tail_small_m = atoi(e);
    if (const char *e = getenv("MGX_MID_SINGLE_DOWN")) p->mid_single_down_m = atoi(e);
    if (const char *e = getenv("MGX_MID_SINGLE_UP")) p->mid_single_up_m = atoi(e);
    if (const char *e = getenv("MGX_MID_DOWN_NL")) p->mid_nl_down = std::min(std::max(atoi(e), 1), MID_MAX_LEVELS);
    if (const char *e = getenv("MGX_MID_UP_NL")) p->mid_nl_up = std::min(std::max(atoi(e), 1), MID_MAX_LEVELS);
    p->lv.resize(depth + 1);
    // The three arrays of a big level start at different phases of the memory's address
    // interleave: allocations are aligned alike, so without this a launch reads v and f at
    // the same phase at every moment.  Measured at 16385^2 (tools/stagger_sweep.sh): on two
    // of three boxes the level-0 up+down launch went from 1.523-1.529 to 1.503-1.514 ms for
    // any of a dozen offsets between 256 KB and 1.2 MB, on the third every placement,
    // this one and none included, gave 1.504-1.519 ms: it removes a bad case, no more.
    // MGX_STAGGER="bytes,bytes" (f, second iterate buffer) overrides, "0,0" = none.
    size_t shift_f = 384 << 10, shift_v1 = 768 << 10;
    if (const char *e = getenv("MGX_STAGGER")) {
        long a = 0, b = 0;
        if (sscanf(e, "%ld,%ld", &a, &b) == 2 && a >= 0 && b >= 0) {
            shift_f = (size_t)a & ~(size_t)255;
            shift_v1 = (size_t)b & ~(size_t)255;
        }
    }
    double h = dx;
    for (int d = 0; d <= depth; d++) {
        Level &L = p->lv[d];
        L.m = dims[d];
        L.dx = h;
        h = h * 2.0;  // main.cpp:297
        const size_t nb = sizeof(double) * L.n();
        const bool big = nb >= ((size_t)32 << 20);
        rc = plan_alloc(p, (void **)&L.f, nb, big ? shift_f : 0);
        if (!rc) rc = plan_alloc(p, (void **)&L.v[0], nb);
        if (!rc) rc = plan_alloc(p, (void **)&L.v[1], nb, big ? shift_v1 : 0);
        if (rc) { mgx_plan_destroy(p); return rc; }
        // main.cpp:295-299: every level starts as zeros
        hipError_t e = hipMemsetAsync(L.f, 0, nb, p->stream);
        if (e == hipSuccess) e = hipMemsetAsync(L.v[0], 0, nb, p->stream);
        if (e == hipSuccess) e = hipMemsetAsync(L.v[1], 0, nb, p->stream);
        if (e != hipSuccess) { mgx_plan_destroy(p); return fail(MGX_ERR_HIP, "hipMemsetAsync: %s", hipGetErrorString(e)); }
    }
    rc = plan_alloc(p, (void **)&p->r, sizeof(double) * p->lv[0].n());
    if (rc) { mgx_plan_destroy(p); return rc; }
    if (p->coarse_solve == MGX_COARSE_EXACT && dims[depth] != 3) {
        const size_t n = (size_t)dims[depth] - 2;
        std::vector<double> t(n * n + n);
        coarse_tables(dims[depth], t.data());
        rc = plan_alloc(p, (void **)&p->exact_tables, sizeof(double) * t.size());
        if (rc) { mgx_plan_destroy(p); return rc; }
        hipError_t e = hipMemcpyAsync(p->exact_tables, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice,
                                      p->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(p->stream);  // t goes out of scope
        if (e != hipSuccess) { mgx_plan_destroy(p); return fail(MGX_ERR_HIP, "table upload: %s", hipGetErrorString(e)); }
    }
    *out = p;
    return MGX_OK;
}

int mgx_plan_destroy(mgx_plan *p)
{
    DeviceScope on_plan_device(p);
    if (!p) return MGX_OK;
    (void)hipStreamSynchronize(p->stream);
    for (void *a : p->allocs) (void)hipFree(a);
    if (p->slots) (void)hipFree(p->slots);
    for (hipEvent_t e : p->events) (void)hipEventDestroy(e);
    for (hipEvent_t e : p->evpool) (void)hipEventDestroy(e);
    delete p;
    return MGX_OK;
}

int mgx_plan_depth(const mgx_plan *p) { return p ? p->depth : MGX_ERR_ARG; }

int mgx_plan_level_m(const mgx_plan *p, int level)
{
    if (!p || level < 0 || level > p->depth) return MGX_ERR_ARG;
    return p->lv[level].m;
}

size_t mgx_plan_bytes(const mgx_plan *p) { return p ? p->bytes : 0; }

double *mgx_plan_f_dev(mgx_plan *p) { return p ? p->lv[0].f : nullptr; }

double *mgx_plan_v_dev(mgx_plan *p)
{
    if (!p) return nullptr;
    DeviceScope on_plan_device(p);
    if (materialize(p) != MGX_OK) return nullptr;
    return p->lv[0].v[0];
}

int mgx_plan_commit(mgx_plan *p)
{
    DeviceScope on_plan_device(p);
    if (!p) return fail(MGX_ERR_ARG, "mgx_plan_commit: null plan");
    Level &L = p->lv[0];
    {
        const int rc = forget_carry(p);
        if (rc) return rc;
    }
    L.cur = 0;
    // both ping-pong buffers must carry the same Dirichlet boundary values, and so must the
    // third array a call's closing launch may write the iterate to
    HIP_TRY(hipMemcpyAsync(L.v[1], L.v[0], sizeof(double) * L.n(), hipMemcpyDeviceToDevice, p->stream));
    if (p->carry_capable && p->schedule == MGX_SCHEDULE_FUSED) return copy_boundary(p->r, L.v[0], L.m, p->stream);
    return MGX_OK;
}

int mgx_plan_profile(mgx_plan *p, int enable)
{
    if (!p) return fail(MGX_ERR_ARG, "mgx_plan_profile: null plan");
    p->profile = enable == 2 ? 2 : (enable ? 1 : 0);
    for (int k = 0; k < MGX_KERNEL_COUNT; k++) { p->prof_ms[k] = 0; p->prof_n[k] = 0; }
    p->lvl_ms.assign((size_t)p->depth + 1, 0.0);
    p->lvl_n.assign((size_t)p->depth + 1, 0);
    return MGX_OK;
}

int mgx_plan_profile_level(const mgx_plan *p, int level, long *count, double *total_ms)
{
    if (!p || level < 0 || level > p->depth) return fail(MGX_ERR_ARG, "mgx_plan_profile_level: bad argument");
    const bool have = (size_t)level < p->lvl_ms.size();
    if (count) *count = have ? p->lvl_n[(size_t)level] : 0;
    if (total_ms) *total_ms = have ? p->lvl_ms[(size_t)level] : 0.0;
    return MGX_OK;
}

int mgx_plan_profile_get(const mgx_plan *p, int kernel_id, long *count, double *total_ms)
{
    if (!p || kernel_id < 0 || kernel_id >= MGX_KERNEL_COUNT) return fail(MGX_ERR_ARG, "mgx_plan_profile_get: bad argument");
    if (count) *count = p->prof_n[kernel_id];
    if (total_ms) *total_ms = p->prof_ms[kernel_id];
    return MGX_OK;
}

const char *mgx_kernel_name(int kernel_id)
{
    static const char *names[MGX_KERNEL_COUNT] = {"k_jacobi", "k_residual", "k_restrict", "k_refine<1>",
                                                  "k_residual_norm", "k_stream:down", "k_stream:up",
                                                  "k_stream:updown"};
    return (kernel_id >= 0 && kernel_id < MGX_KERNEL_COUNT) ? names[kernel_id] : "";
}

int mgx_plan_set_dev(mgx_plan *p, const double *f_dev, const double *v_dev)
{
    DeviceScope on_plan_device(p);
    if (!p) return fail(MGX_ERR_ARG, "mgx_plan_set_dev: null plan");
    Level &L = p->lv[0];
    const size_t nb = sizeof(double) * L.n();
    {
        const int rc = forget_carry(p, v_dev == nullptr);
        if (rc) return rc;
    }
    if (f_dev) HIP_TRY(hipMemcpyAsync(L.f, f_dev, nb, hipMemcpyDeviceToDevice, p->stream));
    if (v_dev) HIP_TRY(hipMemcpyAsync(L.v[0], v_dev, nb, hipMemcpyDeviceToDevice, p->stream));
    return mgx_plan_commit(p);
}

int mgx_plan_get_dev(mgx_plan *p, double *v_dev)
{
    DeviceScope on_plan_device(p);
    if (!p || !v_dev) return fail(MGX_ERR_ARG, "mgx_plan_get_dev: null argument");
    {
        const int rc = materialize(p);
        if (rc) return rc;
    }
    Level &L = p->lv[0];
    HIP_TRY(hipMemcpyAsync(v_dev, L.v[0], sizeof(double) * L.n(), hipMemcpyDeviceToDevice, p->stream));
    return MGX_OK;
}

int mgx_plan_upload(mgx_plan *p, const double *f_host, const double *v_host)
{
    DeviceScope on_plan_device(p);
    if (!p) return fail(MGX_ERR_ARG, "mgx_plan_upload: null plan");
    Level &L = p->lv[0];
    {
        const int rc = forget_carry(p, v_host == nullptr);
        if (rc) return rc;
    }
    if (f_host) HIP_TRY(host_to_device(L.f, f_host, L.n(), p->stream));
    if (v_host) HIP_TRY(host_to_device(L.v[0], v_host, L.n(), p->stream));
    int rc = mgx_plan_commit(p);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(p->stream));
    return MGX_OK;
}

int mgx_plan_upload_boundary(mgx_plan *p, const double *v_host)
{
    DeviceScope on_plan_device(p);
    if (!p || !v_host) return fail(MGX_ERR_ARG, "mgx_plan_upload_boundary: null argument");
    Level &L = p->lv[0];
    const size_t m = (size_t)L.m, col = sizeof(double) * m;
    hipStream_t s = p->stream;
    (void)forget_carry(p, false);
    HIP_TRY(hipMemsetAsync(L.v[0], 0, sizeof(double) * L.n(), s));
    // columns 0 and m-1 are contiguous; rows 0 and m-1 are one double per column
    HIP_TRY(hipMemcpyAsync(L.v[0], v_host, col, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(L.v[0] + (m - 1) * m, v_host + (m - 1) * m, col, hipMemcpyHostToDevice, s));
    // (a strided copy straight from pageable memory takes ~130 ms per row at 16385: gather on
    // the host, one small upload into the idle residual scratch, scatter with a kernel)
    std::vector<double> rows(2 * m);
    for (size_t j = 0; j < m; j++) {
        rows[j] = v_host[j * m];
        rows[m + j] = v_host[j * m + (m - 1)];
    }
    HIP_TRY(hipMemcpyAsync(p->r, rows.data(), sizeof(double) * 2 * m, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));  // `rows` is about to go out of scope
    HIP_TRY(launch_set_boundary_rows(L.v[0], L.m, p->r, s));
    int rc = mgx_plan_commit(p);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(s));
    return MGX_OK;
}

int mgx_plan_download(mgx_plan *p, double *v_host)
{
    DeviceScope on_plan_device(p);
    if (!p || !v_host) return fail(MGX_ERR_ARG, "mgx_plan_download: null argument");
    {
        const int rc = materialize(p);
        if (rc) return rc;
    }
    Level &L = p->lv[0];
    HIP_TRY(device_to_host(v_host, L.v[0], L.n(), p->stream));
    return MGX_OK;
}

int mgx_plan_cycles(mgx_plan *p, int Nv, int Ndown, int Nup, double omega, int pure_jacobi,
                    double *resnorms, double *times)
{
    DeviceScope on_plan_device(p);
    if (!p) return fail(MGX_ERR_ARG, "mgx_plan_cycles: null plan");
    if (Nv < 0 || Ndown < 0 || Nup < 0) return fail(MGX_ERR_ARG, "mgx_plan_cycles: negative count");
    hipStream_t s = p->stream;
    int rc = ensure_slots(p, Nv + 1);
    if (!rc) rc = ensure_events(p, Nv + 1);
    if (rc) return rc;
    Level &L0 = p->lv[0];
    const SweepCoef c0 = make_coef(L0.dx, omega);
    HIP_TRY(hipMemsetAsync(p->slots, 0, sizeof(unsigned long long) * (size_t)(Nv + 1), s));
    // Carry between calls (see mgx_plan::carry_valid): does this call continue the previous one,
    // does it start from what that call left, and does it leave the same for the next?
    const bool can_carry = p->carry_capable && p->schedule == MGX_SCHEDULE_FUSED && !pure_jacobi && p->depth > 0 &&
                           Ndown >= 1 && Ndown + Nup <= STREAM_MAX_SWEEPS;
    bool use_carry = false, make_carry = false;
    {
        const bool cont = Nv > 0 && can_carry && p->last_valid && p->last_Ndown == Ndown && p->last_Nup == Nup &&
                          p->last_omega == omega;
        if (!cont) {  // anything else needs the iterate itself in v[0]
            rc = materialize(p);
            if (rc) return rc;
        }
        if (Nv > 0) {
            use_carry = cont && (p->lazy || p->carry_valid);
            if (use_carry) {
                if (p->habit < 3) p->habit++;
            } else if ((p->lazy || p->carry_valid) && p->carry_short && p->habit > 0) {
                p->habit--;  // left behind by a call with other parameters
            }
            p->streak = cont ? p->streak + 1 : 0;
            make_carry = can_carry && (p->force_carry || p->streak >= 2 ||
                                       (p->habit >= 2 && Nv <= mgx_plan::HABIT_MAX_CYCLES));
        }
    }
    if (use_carry) {
        // the previous call's closing launch already ran this call's opening down leg: the swept
        // iterate is in v[0] (or, once somebody asked for the end-of-cycle iterate, in r), its
        // coarse right-hand side in place, its norm known
        HIP_TRY(hipMemcpyAsync(p->slots, &p->carry_key, sizeof(unsigned long long), hipMemcpyHostToDevice, s));
        if (p->lazy) {
            L0.cur = 0;
        } else {
            std::swap(L0.v[1], p->r);
            L0.cur = 1;
        }
    }
    if (Nv > 0) {
        p->lazy = false;
        p->carry_valid = false;  // consumed, or left behind by a call with other parameters
        p->last_valid = false;   // until this call has gone through
    }
    // main.cpp:307.  When the first cycle opens with a fused level-0 down leg, that launch
    // takes the norm of the initial residual along (one pass over v and f saved per call).
    const bool norm0_in_down = p->schedule == MGX_SCHEDULE_FUSED && !pure_jacobi && p->depth > 0 && Nv > 0 &&
                               Ndown <= STREAM_MAX_SWEEPS;
    if (norm0_in_down || use_carry) {
        // nothing here
    } else if (p->schedule == MGX_SCHEDULE_FUSED) {  // the streaming kernel's 16-byte loads: 0.79 vs 1.09 ms at 16385^2
        StreamArgs A = level_args(L0, nullptr, omega);
        A.slot = p->slots + 0;
        rc = stream_chain(p, 0, MGX_KERNEL_RESNORM, s, 0, 0, 2, 0, A, L0.v, L0.cur);
        if (rc) return rc;
    } else {
        LAUNCH(p, 0, MGX_KERNEL_RESNORM, launch_residual_norm(L0.v[L0.cur], L0.f, L0.m, 1, L0.m - 1, c0, p->slots + 0, s));
    }
    HIP_TRY(hipEventRecord(p->events[0], s));
    for (int iter = 0; iter < Nv; iter++) {  // main.cpp:309
        unsigned long long *slot = p->slots + iter + 1;
        bool norm_done = false;
        if (p->schedule == MGX_SCHEDULE_FUSED) {
            if (!pure_jacobi && p->depth > 0) {
                FusedCtx cx{Ndown, Nup, omega, false};
                // lean: the previous cycle's level-0 launch already did this cycle's down leg
                const bool lean = cx.lean() && (Ndown + Nup) >= 1;
                rc = MGX_OK;
                const bool opened = iter == 0 && use_carry;  // the down leg came with the previous call
                if ((iter == 0 || !lean) && !opened)
                    rc = fused_down(p, 0, cx, (iter == 0 && norm0_in_down) ? p->slots + 0 : nullptr);
                if (!rc) rc = fused_coarse_legs(p, cx);
                if (!rc && iter == 0 && !opened) rc = normalize_boundary(p);  // later cycles find it done
                const bool closing = iter + 1 == Nv && make_carry;  // leave the next call's down leg behind
                if (!rc) rc = fused_up(p, 0, cx, slot, lean && (iter + 1 < Nv || closing));
                norm_done = true;
            } else if (!pure_jacobi) {
                rc = vcycle_fused(p, Ndown, Nup, omega, slot);
            } else {  // main.cpp:348 with the norm of :351 in the same pass
                StreamArgs A = level_args(L0, nullptr, omega);
                A.slot = slot;
                rc = stream_chain(p, 0, MGX_KERNEL_FUSED_UP, s, 0, 1, 2, 0, A, L0.v, L0.cur);
                norm_done = true;
            }
        } else if (!pure_jacobi) {
            p->relax_cycle = times ? iter : -1;
            rc = vcycle_stepwise(p, Ndown, Nup, omega);
            p->relax_cycle = -1;
        } else {
            rc = level_relax(p, 0, 1, omega);  // main.cpp:348 (not timed there either)
        }
        if (rc) { p->relax_recs.clear(); return rc; }
        // main.cpp:351-353
        if (!norm_done)
            LAUNCH(p, 0, MGX_KERNEL_RESNORM, launch_residual_norm(L0.v[L0.cur], L0.f, L0.m, 1, L0.m - 1, c0, slot, s));
        HIP_TRY(hipEventRecord(p->events[iter + 1], s));
    }
    if (L0.cur != 0) {  // the buffer written last becomes v[0]
        std::swap(L0.v[0], L0.v[1]);
        L0.cur = 0;
    }
    // (after a closing up+down launch v[0] is the swept iterate and v[1] that launch's input:
    // the end-of-cycle iterate mgx_plan_v_dev() promises is made from them when asked for)
    HIP_TRY(hipStreamSynchronize(s));
    if (Nv > 0 && can_carry) {
        p->last_valid = true;
        p->last_Ndown = Ndown;
        p->last_Nup = Nup;
        p->last_omega = omega;
        p->lazy = make_carry;
        p->carry_short = make_carry && Nv <= mgx_plan::HABIT_MAX_CYCLES;
        if (make_carry)
            HIP_TRY(hipMemcpy(&p->carry_key, p->slots + Nv, sizeof(unsigned long long), hipMemcpyDeviceToHost));
    }
    std::vector<double> relax_s;  // stepwise: seconds inside relax() per cycle
    if (times && p->schedule == MGX_SCHEDULE_STEPWISE) {
        relax_s.assign((size_t)Nv, 0.0);
        for (const mgx_plan::RelaxRec &r : p->relax_recs) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, r.a, r.b));
            relax_s[(size_t)r.cycle] += 1e-3 * (double)ms;
        }
    }
    p->relax_recs.clear();
    rc = prof_collect(p);
    if (rc) return rc;
    if (resnorms) {
        std::vector<unsigned long long> keys((size_t)Nv + 1);
        HIP_TRY(hipMemcpy(keys.data(), p->slots, sizeof(unsigned long long) * keys.size(), hipMemcpyDeviceToHost));
        for (int k = 0; k <= Nv; k++) resnorms[k] = norm_decode(keys[k]);
    }
    if (times) {
        times[0] = 0;  // main.cpp:308
        for (int k = 1; k <= Nv; k++) {
            if (p->schedule == MGX_SCHEDULE_STEPWISE) {  // main.cpp:350: time inside relax() only
                times[k] = times[k - 1] + relax_s[(size_t)k - 1];
                continue;
            }
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, p->events[k - 1], p->events[k]));
            times[k] = times[k - 1] + 1e-3 * (double)ms;  // fused: smoothing is not separable, whole-cycle device time
        }
    }
    return MGX_OK;
}

int mgx_plan_cycles_until(mgx_plan *p, int max_cycles, int Ndown, int Nup, double omega, double rel_tol, int batch,
                          double *resnorms, double *times, int *cycles_done, int *reason)
{
    if (!p) return fail(MGX_ERR_ARG, "mgx_plan_cycles_until: null plan");
    if (max_cycles < 0 || Ndown < 0 || Nup < 0 || batch < 1 || !(rel_tol >= 0.0))
        return fail(MGX_ERR_ARG, "mgx_plan_cycles_until: bad argument");
    std::vector<double> rn((size_t)batch + 1), tm((size_t)batch + 1);
    int done = 0, why = MGX_STOP_MAX_CYCLES, since = 0;
    double r0 = 0, best = 0, tbase = 0;
    if (times) times[0] = 0;
    while (true) {
        const int n = std::min(batch, max_cycles - done);
        // (n == 0 only when max_cycles == 0: the call still yields the initial norm)
        p->force_carry = true;  // more batches are likely to follow: leave each one's successor its down leg
        const int rc = mgx_plan_cycles(p, n, Ndown, Nup, omega, 0, rn.data(), tm.data());
        p->force_carry = false;
        if (rc) return rc;
        if (done == 0) {
            r0 = rn[0];
            best = r0;
            if (resnorms) resnorms[0] = r0;
        }
        bool stop = false;
        for (int k = 1; k <= n; k++) {
            if (resnorms) resnorms[done + k] = rn[k];
            if (times) times[done + k] = tbase + tm[k];
            // the first cycle at or below the tolerance, or the third cycle in a row that fails to
            // bring the norm 10 % below the best seen so far: the residual has reached its
            // rounding floor, eps*8/h^2*max|u|, where it hovers.  A cycle that merely converges
            // slowly (a factor of 0.7 with one sweep and a small omega) keeps going.
            if (!stop && rn[k] <= rel_tol * r0) {
                stop = true;
                why = MGX_STOP_TOLERANCE;
            } else if (!stop) {
                if (rn[k] < 0.9 * best) {
                    best = rn[k];
                    since = 0;
                } else if (++since >= 3) {
                    stop = true;
                    why = MGX_STOP_FLOOR;
                }
            }
        }
        tbase += tm[n];
        done += n;
        if (stop || done >= max_cycles) break;
    }
    if (cycles_done) *cycles_done = done;
    if (reason) *reason = why;
    return MGX_OK;
}

int mgx_tune_mode(int enable) { return set_tune_now(enable != 0) ? 1 : 0; }

int mgx_tune_reset(void)
{
    tune_forget();
    return MGX_OK;
}

size_t mgx_tune_report(char *buf, size_t len)
{
    const std::string r = tune_report();
    if (buf && len) {
        const size_t n = std::min(r.size(), len - 1);
        memcpy(buf, r.data(), n);
        buf[n] = 0;
    }
    return r.size() + 1;
}

int mgx_plan_tune(mgx_plan *p, int Ndown, int Nup, double omega)
{
    DeviceScope on_plan_device(p);
    if (!p) return fail(MGX_ERR_ARG, "mgx_plan_tune: null plan");
    if (Ndown < 0 || Nup < 0) return fail(MGX_ERR_ARG, "mgx_plan_tune: negative count");
    if (p->schedule != MGX_SCHEDULE_FUSED || p->depth == 0) return MGX_OK;  // nothing to choose
    Level &L0 = p->lv[0];
    const size_t nb = sizeof(double) * L0.n();
    // three cycles visit every level-0 launch of a run (first down leg, up+down, last up leg)
    // with the measurement forced; the iterate is put back afterwards (p->r is idle in this
    // schedule), coarse levels are rebuilt by every cycle anyway
    // (r is this call's keep buffer: nothing may be carried in it or written to it meanwhile)
    {
        const int rc = forget_carry(p);
        if (rc) return rc;
    }
    const bool capable = p->carry_capable;
    p->carry_capable = false;
    HIP_TRY(hipMemcpyAsync(p->r, L0.v[0], nb, hipMemcpyDeviceToDevice, p->stream));
    const bool was = set_tune_now(true);
    const int prof = p->profile;
    p->profile = 0;
    const int rc = mgx_plan_cycles(p, 3, Ndown, Nup, omega, 0, nullptr, nullptr);
    p->profile = prof;
    set_tune_now(was);
    p->carry_capable = capable;
    // the caller's iterate goes back whether or not the cycles went through (the commit also
    // resets the ping-pong index a failed call may have left on the second buffer)
    std::string why;
    if (rc) why = mgx_last_error();
    HIP_TRY(hipMemcpyAsync(L0.v[0], p->r, nb, hipMemcpyDeviceToDevice, p->stream));
    const int rc2 = mgx_plan_commit(p);
    if (rc) return fail(rc, "%s", why.c_str());
    if (rc2) return rc2;
    HIP_TRY(hipStreamSynchronize(p->stream));
    return MGX_OK;
}

int mgx_plan_set_stream(mgx_plan *p, void *stream)
{
    DeviceScope on_plan_device(p);
    if (!p) return fail(MGX_ERR_ARG, "mgx_plan_set_stream: null plan");
    if ((hipStream_t)stream == p->stream) return MGX_OK;
    // work already enqueued on the old stream must be visible to the new one
    if (p->events.empty()) {
        int rc = ensure_events(p, 1);
        if (rc) return rc;
    }
    HIP_TRY(hipEventRecord(p->events[0], p->stream));
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, p->events[0], 0));
    p->stream = (hipStream_t)stream;
    return MGX_OK;
}

int mgx_plan_fmg(mgx_plan *p, int ncyc, int Ndown, int Nup, double omega, double *resnorms, double *times)
{
    DeviceScope on_plan_device(p);
    if (!p) return fail(MGX_ERR_ARG, "mgx_plan_fmg: null plan");
    if (ncyc < 0 || Ndown < 0 || Nup < 0) return fail(MGX_ERR_ARG, "mgx_plan_fmg: negative count");
    {
        const int rc = forget_carry(p);
        if (rc) return rc;
    }
    hipStream_t s = p->stream;
    const int depth = p->depth;
    // right-hand side and Dirichlet values of every level
    for (int d = 1; d <= depth; d++) {
        Level &L = p->lv[d - 1];
        Level &C = p->lv[d];
        HIP_TRY(launch_restrict(L.f, L.m, C.f, C.m, 1, C.m - 1, 0, 0, s));
        HIP_TRY(hipMemsetAsync(C.v[0], 0, sizeof(double) * C.n(), s));
        HIP_TRY(launch_inject_boundary(L.v[L.cur], L.m, C.v[0], C.m, s));
        HIP_TRY(hipMemcpyAsync(C.v[1], C.v[0], sizeof(double) * C.n(), hipMemcpyDeviceToDevice, s));
        C.cur = 0;
    }
    if (depth > 0) {
        int rc = coarsest_solve(p);
        if (rc) return rc;
    }
    for (int l = depth > 0 ? depth - 1 : 0; l >= 0; l--) {
        Level &L = p->lv[l];
        if (depth > 0) {
            Level &C = p->lv[l + 1];
            // initial guess: the interpolated solution of the level below; boundary values stay
            HIP_TRY(launch_refine(C.v[C.cur], C.m, L.v[L.cur], L.m, 1, L.m - 1, 0, 0, 0, s));
            // from here on level l+1 holds coarse-grid corrections: zero, with a zero boundary
            HIP_TRY(hipMemsetAsync(C.v[0], 0, sizeof(double) * C.n(), s));
            HIP_TRY(hipMemsetAsync(C.v[1], 0, sizeof(double) * C.n(), s));
            C.cur = 0;
        }
        // ncyc cycles of MultiGrid() on the hierarchy that starts at level l: a view of this plan
        mgx_plan view;
        view.device = p->device;
        view.stream = p->stream;
        view.schedule = p->schedule;
        view.verbose = p->verbose;
        view.coarsest = p->coarsest;
        view.depth = depth - l;
        view.coarse_solve = p->coarse_solve;
        view.exact_tables = p->exact_tables;
        view.use_tail = p->use_tail;
        view.use_mid = p->use_mid;
        view.mid_max_m = p->mid_max_m;
        view.tail_small_m = p->tail_small_m;
        view.mid_single_down_m = p->mid_single_down_m;
        view.mid_single_up_m = p->mid_single_up_m;
        view.mid_nl_down = p->mid_nl_down;
        view.mid_nl_up = p->mid_nl_up;
        view.lv.assign(p->lv.begin() + l, p->lv.end());
        view.r = p->r;
        view.carry_capable = false;
        view.slots = p->slots;
        view.nslots = p->nslots;
        view.events.swap(p->events);
        view.evpool.swap(p->evpool);
        const bool top = l == 0;
        const int rc = mgx_plan_cycles(&view, ncyc, Ndown, Nup, omega, 0, top ? resnorms : nullptr,
                                       top ? times : nullptr);
        p->events.swap(view.events);
        p->evpool.swap(view.evpool);
        p->slots = view.slots;
        p->nslots = view.nslots;
        for (int k = l; k <= depth; k++) p->lv[k] = view.lv[k - l];  // buffers may have traded places
        if (rc) return rc;
    }
    return MGX_OK;
}

int mgx_solve2d(const double *f, double *v, int m, int n, double dx,
                int Nv, int Ndown, int Nup, double omega, int coarsest, int pure_jacobi,
                double *resnorms, double *times, const mgx_options *opt)
{
    if (!f || !v) return fail(MGX_ERR_ARG, "mgx_solve2d: null f or v");
    if (m != n) return fail(MGX_ERR_ARG, "mgx_solve2d: Input is not square matrix! (%dx%d)", m, n);  // main.cpp:408-412
    if (opt && opt->struct_size == (int)sizeof(mgx_options) && opt->ngpus > 1 &&
        opt->schedule != MGX_SCHEDULE_STEPWISE) {
        // the grid split into opt->ngpus j-slabs, one per device; a grid too small to split
        // (MGX_ERR_ARG from the partitioned plan) runs on one device like any other
        mgx_mplan *mp = nullptr;
        int rc = mgx_mplan_create(&mp, m, dx, coarsest, opt->ngpus, nullptr, opt);
        if (rc == MGX_OK) {
            rc = mgx_mplan_upload(mp, f, v);
            if (!rc) rc = mgx_mplan_cycles(mp, Nv, Ndown, Nup, omega, pure_jacobi, resnorms, times);
            if (!rc) rc = mgx_mplan_download(mp, v);
            mgx_mplan_destroy(mp);
            return rc;
        }
        if (rc != MGX_ERR_ARG) return rc;
    }
    mgx_plan *p = nullptr;
    int rc = mgx_plan_create(&p, m, dx, coarsest, opt);
    if (rc) return rc;
    if (opt && opt->struct_size == (int)sizeof(mgx_options) && (opt->flags & MGX_FLAG_ZERO_GUESS)) {
        rc = mgx_plan_upload(p, f, nullptr);  // the interior of v is zero by the caller's word: not read
        if (!rc) rc = mgx_plan_upload_boundary(p, v);
    } else if ((size_t)m * (size_t)m * sizeof(double) >= ((size_t)64 << 20)) {
        // A big guess is usually zeros inside a boundary (main.cpp:420-439 passes exactly that).
        // Host threads look while f is on its way up; if it is so, only the boundary follows,
        // otherwise the array (a guess with data ends the look at once).
        std::future<bool> zero;
        try {
            zero = std::async(std::launch::async, host_interior_is_zero, v, m);
        } catch (...) {  // no thread to be had: look after the upload instead of beside it
        }
        rc = mgx_plan_upload(p, f, nullptr);
        const bool z = zero.valid() ? zero.get() : host_interior_is_zero(v, m);
        if (!rc) rc = z ? mgx_plan_upload_boundary(p, v) : mgx_plan_upload(p, nullptr, v);
    } else {
        rc = mgx_plan_upload(p, f, v);
    }
    if (!rc) rc = mgx_plan_cycles(p, Nv, Ndown, Nup, omega, pure_jacobi, resnorms, times);
    if (!rc) rc = mgx_plan_download(p, v);
    mgx_plan_destroy(p);
    return rc;
}

// ---- per-operator entry points ----------------------------------------------

int mgx_relax(double *v, const double *f, int m, double dx, int niter, double omega, void *stream)
{
    if (!v || !f) return fail(MGX_ERR_ARG, "mgx_relax: null pointer");
    int rc = check_grid(m, "mgx_relax");
    if (rc) return rc;
    if (niter <= 0) return MGX_OK;
    hipStream_t s = (hipStream_t)stream;
    const size_t nb = sizeof(double) * (size_t)m * (size_t)m;
    double *tmp = nullptr;
    HIP_TRY(hipMalloc((void **)&tmp, nb));
    hipError_t e = hipMemcpyAsync(tmp, v, nb, hipMemcpyDeviceToDevice, s);  // boundary values into the pong buffer
    const SweepCoef c = make_coef(dx, omega);
    double *buf[2] = {v, tmp};
    int cur = 0;
    for (int it = 0; it < niter && e == hipSuccess; it++) {
        e = launch_jacobi(buf[cur], buf[cur ^ 1], f, m, 1, m - 1, c, s);
        cur ^= 1;
    }
    if (e == hipSuccess && cur == 1) e = hipMemcpyAsync(v, tmp, nb, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail(MGX_ERR_HIP, "mgx_relax: %s", hipGetErrorString(e));
    return MGX_OK;
}

int mgx_residual(const double *v, const double *f, double *r, int m, double dx, void *stream)
{
    if (!v || !f || !r) return fail(MGX_ERR_ARG, "mgx_residual: null pointer");
    int rc = check_grid(m, "mgx_residual");
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const SweepCoef c = make_coef(dx, 0.0);
    HIP_TRY(hipMemsetAsync(r, 0, sizeof(double) * (size_t)m, s));
    HIP_TRY(hipMemsetAsync(r + (size_t)(m - 1) * m, 0, sizeof(double) * (size_t)m, s));
    HIP_TRY(launch_residual(v, f, r, m, 1, m - 1, c, s));
    return MGX_OK;
}

int mgx_restrict(const double *fine, int mf, double *coarse, int mc, void *stream)
{
    if (!fine || !coarse) return fail(MGX_ERR_ARG, "mgx_restrict: null pointer");
    int rc = check_grid(mf, "mgx_restrict");
    if (rc) return rc;
    if (mc != (mf - 1) / 2 + 1 || mc < 2) return fail(MGX_ERR_ARG, "mgx_restrict: mc=%d does not pair with mf=%d", mc, mf);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(coarse, 0, sizeof(double) * (size_t)mc * mc, s));  // main.cpp:211
    HIP_TRY(launch_restrict(fine, mf, coarse, mc, 1, mc - 1, 0, 0, s));
    return MGX_OK;
}

int mgx_refine(const double *coarse, int mc, double *fine, int mf, void *stream)
{
    if (!fine || !coarse) return fail(MGX_ERR_ARG, "mgx_refine: null pointer");
    int rc = check_grid(mf, "mgx_refine");
    if (rc) return rc;
    if (mc != (mf - 1) / 2 + 1) return fail(MGX_ERR_ARG, "mgx_refine: mc=%d does not pair with mf=%d", mc, mf);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(fine, 0, sizeof(double) * (size_t)mf * mf, s));  // main.cpp:229
    HIP_TRY(launch_refine(coarse, mc, fine, mf, 1, mf - 1, 0, 0, 0, s));
    return MGX_OK;
}

int mgx_prolong_add(const double *coarse, int mc, double *v, int mf, void *stream)
{
    if (!v || !coarse) return fail(MGX_ERR_ARG, "mgx_prolong_add: null pointer");
    int rc = check_grid(mf, "mgx_prolong_add");
    if (rc) return rc;
    if (mc != (mf - 1) / 2 + 1) return fail(MGX_ERR_ARG, "mgx_prolong_add: mc=%d does not pair with mf=%d", mc, mf);
    HIP_TRY(launch_refine(coarse, mc, v, mf, 1, mf - 1, 0, 0, 1, (hipStream_t)stream));
    return MGX_OK;
}

int mgx_direct3x3(double *v, const double *f, int m, double dx, void *stream)
{
    if (!v || !f) return fail(MGX_ERR_ARG, "mgx_direct3x3: null pointer");
    if (m != 3) return 1;  // main.cpp:167: message, no change
    HIP_TRY(launch_direct3x3(v, f, dx * dx, (hipStream_t)stream));
    return MGX_OK;
}

int mgx_coarse_exact(double *v, const double *f, int m, double dx, void *stream)
{
    if (!v || !f) return fail(MGX_ERR_ARG, "mgx_coarse_exact: null pointer");
    if (m < 4 || m > COARSE_EXACT_MAX_M)
        return fail(MGX_ERR_ARG, "mgx_coarse_exact: m=%d outside 4..%d", m, COARSE_EXACT_MAX_M);
    const size_t n = (size_t)m - 2;
    std::vector<double> t(n * n + n);
    coarse_tables(m, t.data());
    double *dt = nullptr;
    HIP_TRY(hipMalloc((void **)&dt, sizeof(double) * t.size()));
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemcpyAsync(dt, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = launch_coarse_exact(v, f, m, dx * dx, dt, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(dt);
    if (e != hipSuccess) return fail(MGX_ERR_HIP, "mgx_coarse_exact: %s", hipGetErrorString(e));
    return MGX_OK;
}

// v (in/out) through a temporary pong buffer: copies the boundary over, runs the
// chain, copies the result back if it ended in the temporary.
static int fused_on_copy(int pro, int ns, int res, StreamArgs A, double *v, size_t n, hipStream_t s)
{
    double *tmp = nullptr;
    HIP_TRY(hipMalloc((void **)&tmp, sizeof(double) * n));
    int rc = MGX_OK;
    hipError_t e = hipMemcpyAsync(tmp, v, sizeof(double) * n, hipMemcpyDeviceToDevice, s);
    double *buf[2] = {v, tmp};
    int cur = 0;
    if (e == hipSuccess) rc = stream_chain(nullptr, 0, 0, s, pro, ns, res, 0, A, buf, cur);
    if (e == hipSuccess && rc == MGX_OK && cur == 1) e = hipMemcpyAsync(v, tmp, sizeof(double) * n, hipMemcpyDeviceToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(tmp);
    if (rc) return rc;
    if (e != hipSuccess) return fail(MGX_ERR_HIP, "fused launch: %s", hipGetErrorString(e));
    return MGX_OK;
}

int mgx_fused_down(double *v, const double *f, double *coarse_f, int m, double dx, int nsweeps, double omega, void *stream)
{
    if (!v || !f || !coarse_f) return fail(MGX_ERR_ARG, "mgx_fused_down: null pointer");
    int rc = check_grid(m, "mgx_fused_down");
    if (rc) return rc;
    if (nsweeps < 0) return fail(MGX_ERR_ARG, "mgx_fused_down: negative sweep count");
    hipStream_t s = (hipStream_t)stream;
    Level L, C;
    L.m = m; L.dx = dx; L.f = const_cast<double *>(f);
    C.m = (m - 1) / 2 + 1;
    StreamArgs A = level_args(L, &C, omega);
    A.cout = coarse_f;
    HIP_TRY(hipMemsetAsync(coarse_f, 0, sizeof(double) * C.n(), s));  // main.cpp:211
    return fused_on_copy(0, nsweeps, 1, A, v, L.n(), s);
}

int mgx_fused_up(const double *coarse_v, double *v, const double *f, int m, double dx, int nsweeps, double omega,
                 double *norm_host, void *stream)
{
    if (!v || !f || !coarse_v) return fail(MGX_ERR_ARG, "mgx_fused_up: null pointer");
    int rc = check_grid(m, "mgx_fused_up");
    if (rc) return rc;
    if (nsweeps < 0) return fail(MGX_ERR_ARG, "mgx_fused_up: negative sweep count");
    hipStream_t s = (hipStream_t)stream;
    Level L, C;
    L.m = m; L.dx = dx; L.f = const_cast<double *>(f);
    C.m = (m - 1) / 2 + 1;
    StreamArgs A = level_args(L, &C, omega);
    A.cin = coarse_v;
    unsigned long long *slot = nullptr;
    if (norm_host) {
        HIP_TRY(hipMalloc((void **)&slot, sizeof(*slot)));
        hipError_t e = hipMemsetAsync(slot, 0, sizeof(*slot), s);
        if (e != hipSuccess) { (void)hipFree(slot); return fail(MGX_ERR_HIP, "mgx_fused_up: %s", hipGetErrorString(e)); }
        A.slot = slot;
    }
    rc = fused_on_copy(1, nsweeps, norm_host ? 2 : 0, A, v, L.n(), s);
    if (!rc && norm_host) {
        unsigned long long key = 0;
        hipError_t e = hipMemcpy(&key, slot, sizeof(key), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(MGX_ERR_HIP, "mgx_fused_up: %s", hipGetErrorString(e));
        *norm_host = norm_decode(key);
    }
    if (slot) (void)hipFree(slot);
    return rc;
}

int mgx_slab_launch(const mgx_slab_op *op, void *stream)
{
    if (!op) return fail(MGX_ERR_ARG, "mgx_slab_launch: null op");
    // (a caller built against the struct without the second range passes the smaller size)
    if (op->struct_size != (int)sizeof(mgx_slab_op) && op->struct_size != (int)offsetof(mgx_slab_op, jlo2))
        return fail(MGX_ERR_ARG, "mgx_slab_launch: struct_size %d is neither %d nor %d", op->struct_size,
                    (int)sizeof(mgx_slab_op), (int)offsetof(mgx_slab_op, jlo2));
    StreamOp so;
    so.ns = op->nsweeps;
    so.pa = op->pro_stage;
    so.nm = op->norm_stage;
    so.res = op->restrict_out ? 1 : 0;
    so.zero_in = op->zero_in ? 1 : 0;
    so.store = op->store ? 1 : 0;
    const int ns = so.ns;
    if (ns < 0 || ns > MGX_SLAB_MAX_SWEEPS) return fail(MGX_ERR_ARG, "mgx_slab_launch: nsweeps %d outside 0..%d", ns, MGX_SLAB_MAX_SWEEPS);
    if (so.pa < -1 || so.pa > ns || so.nm < -1 || so.nm > ns)
        return fail(MGX_ERR_ARG, "mgx_slab_launch: pro_stage %d / norm_stage %d outside -1..%d", so.pa, so.nm, ns);
    if (so.store && ns == 0 && so.pa < 0) return fail(MGX_ERR_ARG, "mgx_slab_launch: store with nothing computed");
    if (op->m < 3) return fail(MGX_ERR_ARG, "mgx_slab_launch: m %d < 3", op->m);
    if (op->ld < op->m || op->ncols < 1) return fail(MGX_ERR_ARG, "mgx_slab_launch: bad pitch/ncols");
    if (!op->f || (!so.zero_in && !op->v_in)) return fail(MGX_ERR_ARG, "mgx_slab_launch: null input");
    if (so.store && !op->v_out) return fail(MGX_ERR_ARG, "mgx_slab_launch: null v_out");
    if (so.store && !so.zero_in && op->v_out == op->v_in) return fail(MGX_ERR_ARG, "mgx_slab_launch: v_out must not alias v_in");
    if (!so.store && !so.res && so.nm < 0) return MGX_OK;
    if ((so.pa >= 0 && !op->coarse_v) || (so.res && !op->coarse_f) || (so.nm >= 0 && !op->norm_slot))
        return fail(MGX_ERR_ARG, "mgx_slab_launch: null coarse array or norm slot");
    // the second produced range exists in the struct from version 210 on
    const bool has2 = op->struct_size >= (int)sizeof(mgx_slab_op);
    const int jlo2 = has2 ? op->jlo2 : 0, jhi2 = has2 ? op->jhi2 : 0;
    if (op->jlo < 0 || op->jhi > op->ncols || op->jlo > op->jhi)
        return fail(MGX_ERR_ARG, "mgx_slab_launch: produced range [%d,%d) outside 0..%d", op->jlo, op->jhi, op->ncols);
    if (jhi2 < jlo2 || (jhi2 > jlo2 && (jlo2 < op->jhi || jhi2 > op->ncols)))
        return fail(MGX_ERR_ARG, "mgx_slab_launch: second range [%d,%d) must follow the first [%d,%d) inside 0..%d", jlo2,
                    jhi2, op->jlo, op->jhi, op->ncols);
    if (op->jlo == op->jhi && jlo2 == jhi2) return MGX_OK;
    const int flo = op->jfix_lo, fhi = op->jfix_hi;
    if ((flo >= 0 && flo != 0) || (fhi >= 0 && fhi != op->ncols - 1))
        return fail(MGX_ERR_ARG, "mgx_slab_launch: a Dirichlet column must be the first / last column of the slab");
    // halo the pipeline consumes around the produced columns
    const int need = stream_halo(so);
    for (int rg = 0; rg < 2; rg++) {
        const int jlo = rg ? jlo2 : op->jlo, jhi = rg ? jhi2 : op->jhi;
        if (jlo == jhi) continue;
        if ((flo == 0 && jlo < 1) || (fhi >= 0 && jhi > fhi))
            return fail(MGX_ERR_ARG, "mgx_slab_launch: a Dirichlet column cannot be produced");
        if (flo < 0 && jlo - need < 0)
            return fail(MGX_ERR_ARG, "mgx_slab_launch: %d halo columns needed left of column %d", need, jlo);
        if (fhi < 0 && jhi - 1 + need > op->ncols - 1)
            return fail(MGX_ERR_ARG, "mgx_slab_launch: %d halo columns needed right of column %d", need, jhi - 1);
        if (so.pa >= 0 || so.res) {
            if (op->mc != (op->m - 1) / 2 + 1 || op->ldc < op->mc || op->nccols < 1)
                return fail(MGX_ERR_ARG, "mgx_slab_launch: coarse slab %d rows does not pair with %d fine rows", op->mc, op->m);
            // coarse columns touched: interpolation reads J and J+1 of every column
            // of stage pro_stage that the produced range depends on; restriction
            // writes J of the produced even columns
            int jf_lo = jlo, jf_hi = jhi - 1;
            if (so.pa >= 0) {
                const int reach = need - so.pa;
                jf_lo = jlo - reach < 0 ? 0 : jlo - reach;
                jf_hi = jhi - 1 + reach > op->ncols - 1 ? op->ncols - 1 : jhi - 1 + reach;
            }
            const int Jlo = ((jf_lo + op->gf0) >> 1) - op->gc0;
            const int Jhi = ((jf_hi + op->gf0 + (so.pa >= 0 ? 1 : 0)) >> 1) - op->gc0;
            if (Jlo < 0 || Jhi > op->nccols - 1)
                return fail(MGX_ERR_ARG, "mgx_slab_launch: coarse columns %d..%d outside the coarse slab 0..%d", Jlo, Jhi,
                            op->nccols - 1);
        }
    }
    if (!(op->dx > 0.0)) return fail(MGX_ERR_ARG, "mgx_slab_launch: dx must be positive");
    StreamArgs A;
    A.uin = op->v_in;
    A.uout = op->v_out;
    A.f = op->f;
    A.cin = op->coarse_v;
    A.cout = op->coarse_f;
    A.slot = op->norm_slot;
    A.m = op->m;
    A.ld = op->ld;
    A.ncols = op->ncols;
    A.mc = op->mc;
    A.ldc = op->ldc;
    A.nccols = op->nccols;
    A.jlo = op->jlo;
    A.jhi = op->jhi;
    A.jlo2 = jlo2;
    A.jhi2 = jhi2;
    A.jfix_lo = flo;
    A.jfix_hi = fhi;
    A.gf0 = op->gf0;
    A.gc0 = op->gc0;
    A.c = make_coef(op->dx, op->omega);
    const hipError_t e = launch_stream(so, A, (hipStream_t)stream);
    if (e == hipErrorInvalidValue)
        return fail(MGX_ERR_ARG, "mgx_slab_launch: no kernel for nsweeps=%d pro_stage=%d norm_stage=%d restrict=%d zero_in=%d "
                                 "store=%d (see the leg list in include/mgx.h)", so.ns, so.pa, so.nm, so.res, so.zero_in, so.store);
    HIP_TRY(e);
    return MGX_OK;
}

double mgx_norm_decode(unsigned long long slot) { return norm_decode(slot); }

int mgx_plan_vcycle_async(mgx_plan *p, int Ndown, int Nup, double omega, int zero_guess)
{
    DeviceScope on_plan_device(p);
    if (!p) return fail(MGX_ERR_ARG, "mgx_plan_vcycle_async: null plan");
    if (Ndown < 0 || Nup < 0) return fail(MGX_ERR_ARG, "mgx_plan_vcycle_async: negative count");
    {
        const int rc = forget_carry(p);
        if (rc) return rc;
    }
    if (zero_guess && p->schedule != MGX_SCHEDULE_FUSED) {
        // the stepwise schedule reads the iterate: make it zero in memory
        Level &L = p->lv[0];
        HIP_TRY(hipMemsetAsync(L.v[L.cur], 0, sizeof(double) * L.n(), p->stream));
    }
    int rc = p->schedule == MGX_SCHEDULE_FUSED ? vcycle_fused(p, Ndown, Nup, omega, nullptr, zero_guess != 0)
                                               : vcycle_stepwise(p, Ndown, Nup, omega);
    if (rc) return rc;
    Level &L0 = p->lv[0];
    if (L0.cur != 0) {  // keep the result in the buffer mgx_plan_v_dev() exposes
        HIP_TRY(hipMemcpyAsync(L0.v[0], L0.v[1], sizeof(double) * L0.n(), hipMemcpyDeviceToDevice, p->stream));
        L0.cur = 0;
    }
    return MGX_OK;
}

static int read_slot(unsigned long long *slot, double *out, hipStream_t s)
{
    unsigned long long key = 0;
    HIP_TRY(hipMemcpyAsync(&key, slot, sizeof(key), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    *out = norm_decode(key);
    return MGX_OK;
}

int mgx_norminf(const double *a, size_t len, double *out, void *stream)
{
    if ((!a && len) || !out) return fail(MGX_ERR_ARG, "mgx_norminf: null pointer");
    hipStream_t s = (hipStream_t)stream;
    unsigned long long *slot = nullptr;
    HIP_TRY(hipMalloc((void **)&slot, sizeof(*slot)));
    hipError_t e = hipMemsetAsync(slot, 0, sizeof(*slot), s);
    if (e == hipSuccess && len) e = launch_norminf(a, len, slot, s);
    int rc = (e == hipSuccess) ? read_slot(slot, out, s) : fail(MGX_ERR_HIP, "mgx_norminf: %s", hipGetErrorString(e));
    (void)hipFree(slot);
    return rc;
}

int mgx_residual_norm(const double *v, const double *f, int m, double dx, double *out, void *stream)
{
    if (!v || !f || !out) return fail(MGX_ERR_ARG, "mgx_residual_norm: null pointer");
    int rc = check_grid(m, "mgx_residual_norm");
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    unsigned long long *slot = nullptr;
    HIP_TRY(hipMalloc((void **)&slot, sizeof(*slot)));
    hipError_t e = hipMemsetAsync(slot, 0, sizeof(*slot), s);
    if (e == hipSuccess) e = launch_residual_norm(v, f, m, 1, m - 1, make_coef(dx, 0.0), slot, s);
    rc = (e == hipSuccess) ? read_slot(slot, out, s) : fail(MGX_ERR_HIP, "mgx_residual_norm: %s", hipGetErrorString(e));
    (void)hipFree(slot);
    return rc;
}

} // extern "C"
