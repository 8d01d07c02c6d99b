// mgx_multi.hip -- the V-cycle of MultiGrid() (reference main.cpp:283-357) over several
// j-slabs in ONE process: mgx_mplan_* and the ngpus option of mgx_solve2d().
//
// The reference is a single-threaded, single-device program; nothing here has a counterpart
// in it except the schedule it must reproduce bit for bit.
//
// Decomposition (the same one multigrid_amd/slab.py runs with one process per GPU).  Every
// level is column-major, so a range of columns j is one contiguous block.  The finest `nd`
// levels are DISTRIBUTED: slab g owns columns [b_g, b_{g+1}) of a level and stores them with
// W halo columns per side.  Ownership is nested, b_g(fine) = 2 b_g(coarse) - 1, so the owner
// of fine column 2J owns coarse column J and both level transfers are local apart from the
// halo.  Levels from the first one with at most `tail_m` points per side downwards form the
// TAIL: every slab holds all of it (a single-device plan) and runs it redundantly -- same
// bits everywhere, no scatter on the way up.
//
// One host thread per slab drives that slab's device: it enqueues the slab's launches on
// its compute stream and PULLS what it needs from its neighbours with peer copies on a
// second stream.  Threads meet through two things only:
//   * an atomic counter per slab and purpose ("my data of exchange x is enqueued"), which
//     tells a neighbour that the HIP event it is about to wait for has been recorded;
//   * HIP events recorded by the producer and waited for (hipStreamWaitEvent) by the
//     consumer's stream, which order the device work.  No thread ever waits for a device.
// Communication per cycle with nu1+nu2 <= 4 (the "lean" schedule): ONE neighbour exchange
// (8 columns of the level-0 iterate + a few columns of the level-1 right-hand side per side,
// overlapped with the interior part of the level-0 launch) and ONE gather of the tail's
// right-hand side; every other halo column is recomputed (Jacobi makes recomputation
// bit-identical).  With more sweeps every leg is a chain of launches, an exchange after each.
//
// Several slabs may live on one device (devices[] may repeat an ordinal): the driver is the
// same, peer copies become device-to-device copies.  That is how the partitioned schedule is
// tested on a one-GPU box and how `ngpus` beyond the number of devices is honoured.
#include "../../include/mgx.h"
#include "mgx_internal.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <vector>

using namespace mgx;

namespace {

constexpr int MAX_LEAN = MGX_SLAB_MAX_SWEEPS;  // nu1 + nu2 up to this run the lean legs
constexpr int W0 = 8;                          // halo columns of the level-0 arrays

// ---- halo widths of the lean schedule --------------------------------------------------
// Ev[l] = columns beyond its own that a slab produces of the level-l iterate on the way up
// (l >= 1), Ef[l] = columns beyond its own of the level-l right-hand side a slab holds
// (pulled for l = 1, recomputed for l >= 2).  With NS = Ndown + Nup: the level-0 launch
// reads the level-1 iterate within (NS+2)/2+1 coarse columns of its own; an up leg that
// produces e extra columns needs f within e + NS and the coarser iterate within
// (e + Nup)/2 + 1; a down leg that produces the coarser f on e extra columns needs f within
// 2e + Ndown + 2.
void halo_plan(int nd, int Ndown, int Nup, std::vector<int> &Ev, std::vector<int> &Ef)
{
    const int NS = Ndown + Nup;
    Ev.assign(nd + 1, 0);
    Ef.assign(nd + 1, 0);
    if (nd >= 2) {
        Ev[1] = (NS + 2) / 2 + 1;
        for (int l = 2; l < nd; l++) Ev[l] = (Ev[l - 1] + Nup) / 2 + 1;
        for (int l = nd - 1; l >= 1; l--)
            Ef[l] = std::max(Ev[l] + NS, l + 1 < nd ? 2 * Ef[l + 1] + Ndown + 2 : 0);
    }
}

// what the arrays must hold: the maximum over every lean (Ndown, Nup)
void halo_capacity(int nd, std::vector<int> &Ev, std::vector<int> &Ef)
{
    Ev.assign(nd + 1, 0);
    Ef.assign(nd + 1, 0);
    std::vector<int> ev, ef;
    for (int a = 0; a <= MAX_LEAN; a++)
        for (int b = 0; a + b <= MAX_LEAN; b++) {
            if (a + b == 0) continue;
            halo_plan(nd, a, b, ev, ef);
            for (int l = 0; l <= nd; l++) {
                Ev[l] = std::max(Ev[l], ev[l]);
                Ef[l] = std::max(Ef[l], ef[l]);
            }
        }
}

struct SLevel {
    int m = 0;
    double dx = 0;
    int W = 0;                 // halo columns the arrays hold per side
    int b0 = 0, b1 = 0;        // owned global columns [b0, b1)
    int lo = 0, hi = 0;        // stored global columns [lo, hi)
    int ncols = 0;
    int o0 = 0, o1 = 0;        // owned, local
    int jfix_lo = -1, jfix_hi = -1;
    double *f = nullptr;
    double *v[2] = {nullptr, nullptr};
    int cur = 0;
    size_t bytes() const { return sizeof(double) * (size_t)m * (size_t)ncols; }
};

struct XItem {  // one array whose halo columns are refreshed from the neighbours
    int level;
    bool is_f;
    int width;
};

struct CycleArgs {
    int Nv = 0, Ndown = 0, Nup = 0, pure_jacobi = 0;
    double omega = 0;
    bool tune = false;
    bool fmg = false;
    int top = 0;  // the cycles run on the hierarchy that starts at this distributed level (full multigrid)
};

} // namespace

struct mgx_mplan;

namespace {

struct Slab {
    mgx_mplan *mp = nullptr;
    int rank = 0, dev = 0;
    hipStream_t compute = nullptr, comm = nullptr;
    std::vector<SLevel> lv;
    mgx_plan *tail = nullptr;
    double *tail_f = nullptr, *tail_v = nullptr;
    double *tailsrc[2] = {nullptr, nullptr};  // this slab's columns of the tail's right-hand side, by cycle parity
    unsigned long long *slots = nullptr;
    int nslots = 0;
    hipEvent_t ready[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr}, tready[2] = {nullptr, nullptr};
    std::vector<hipEvent_t> cyc;  // slab 0: one per cycle boundary (Times)
    long xcount = 0;              // exchanges begun
    long tcount = 0;              // tail gathers begun
    std::vector<XItem> pending;
    std::vector<int> Ev, Ef;  // halo widths of the lean schedule, indexed from the view's top level
    int top = 0;              // top level of the hierarchy the current call cycles on
    size_t bytes = 0;
    int rc = MGX_OK;
    std::string err;
    // per-launch timing of the level-0 launches (mgx_mplan_profile)
    struct Rec { int kid; hipEvent_t a, b; long pts; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> evpool;
    size_t evused = 0;
};

} // namespace

struct mgx_mplan {
    int m = 0, coarsest = 2, depth = 0, nd = 0, R = 1;
    double dx = 0;
    int tail_m = 0;
    int overlap = 1;
    std::vector<int> dims;
    std::vector<double> hs;
    std::vector<std::vector<int>> bounds;  // bounds[l][g], l = 0..nd
    std::vector<std::unique_ptr<Slab>> slabs;
    std::unique_ptr<std::atomic<long>[]> posted, pulled, tposted;
    std::atomic<int> failed{0};
    // Set when a call failed part-way: the slabs' hand-shake counters, ping-pong indices and
    // streams are then out of step with each other, so every entry point but destroy refuses.
    int poisoned = 0;
    std::string poison_msg;
    int profile = 0;
    double prof_ms[MGX_KERNEL_COUNT] = {0};
    long prof_n[MGX_KERNEL_COUNT] = {0};
    double prof_pts[MGX_KERNEL_COUNT] = {0};
    long nexchanges = 0, ngathers = 0;
    double host_seconds = 0;     // wall time of the last mgx_mplan_cycles() call
    double enqueue_seconds = 0;  // of which slab 0's thread spent enqueuing (hand-shakes included), before it waited for its device
};

namespace {

int slab_fail(Slab &S, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    S.rc = code;
    S.err = buf;
    S.mp->failed.store(1, std::memory_order_release);
    return code;
}

#define S_HIP(S, expr)                                                                                   \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            return slab_fail(S, MGX_ERR_HIP, "slab %d: %s failed: %s (%s:%d)", (S).rank, #expr,          \
                             hipGetErrorString(e_), __FILE__, __LINE__);                                 \
    } while (0)

#define S_TRY(expr)                   \
    do {                              \
        const int rc_ = (expr);       \
        if (rc_ != MGX_OK) return rc_; \
    } while (0)

// wait (on the host) until a neighbour's counter reaches x; gives up when any slab failed,
// or after MGX_MPLAN_TIMEOUT seconds (default 120): the slabs then disagree about the number
// of hand-shakes of the schedule, which is a bug here and must not hang the caller
bool spin_until(mgx_mplan *mp, const std::atomic<long> &c, long x)
{
    static const double limit = [] {
        const char *e = getenv("MGX_MPLAN_TIMEOUT");
        const double v = e ? atof(e) : 120.0;
        return v > 0 ? v : 120.0;
    }();
    long n = 0;
    timespec t0 = {0, 0};
    while (c.load(std::memory_order_acquire) < x) {
        if (mp->failed.load(std::memory_order_acquire)) return false;
        if (++n > 64) std::this_thread::yield();
        if ((n & 0xfff) == 0) {
            timespec t;
            clock_gettime(CLOCK_MONOTONIC, &t);
            if (t0.tv_sec == 0) t0 = t;
            if ((double)(t.tv_sec - t0.tv_sec) > limit) {
                mp->failed.store(2, std::memory_order_release);
                return false;
            }
        }
    }
    return true;
}

hipEvent_t pool_event(Slab &S)
{
    if (S.evused == S.evpool.size()) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        S.evpool.push_back(e);
    }
    return S.evpool[S.evused++];
}

struct CoarseShape {
    int mc = 0, nccols = 0, gc0 = 0;
};

struct OpSpec {
    int nsweeps = 0, pro_stage = -1, norm_stage = -1, restrict_out = 0, zero_in = 0, store = 1;
    const double *v_in = nullptr;
    double *v_out = nullptr;
    const double *coarse_v = nullptr;
    double *coarse_f = nullptr;
    unsigned long long *norm_slot = nullptr;
    CoarseShape cs;
    int jlo = 0, jhi = 0;
    int jlo2 = 0, jhi2 = 0;  // a second column range produced by the same launch (none if equal)
    double omega = 0;
};

int launch(Slab &S, int level, int kid, const OpSpec &o)
{
    const SLevel &L = S.lv[level];
    mgx_slab_op op;
    memset(&op, 0, sizeof(op));
    op.struct_size = (int)sizeof(op);
    op.nsweeps = o.nsweeps;
    op.pro_stage = o.pro_stage;
    op.norm_stage = o.norm_stage;
    op.restrict_out = o.restrict_out;
    op.zero_in = o.zero_in;
    op.store = o.store;
    op.v_in = o.v_in;
    op.v_out = o.v_out;
    op.f = L.f;
    op.coarse_v = o.coarse_v;
    op.coarse_f = o.coarse_f;
    op.norm_slot = o.norm_slot;
    op.m = L.m;
    op.ld = L.m;
    op.ncols = L.ncols;
    op.mc = o.cs.mc;
    op.ldc = o.cs.mc;
    op.nccols = o.cs.nccols;
    op.jlo = o.jlo;
    op.jhi = o.jhi;
    op.jlo2 = o.jlo2;
    op.jhi2 = o.jhi2;
    op.jfix_lo = L.jfix_lo;
    op.jfix_hi = L.jfix_hi;
    op.gf0 = L.lo;
    op.gc0 = o.cs.gc0;
    op.dx = L.dx;
    op.omega = o.omega;
    const bool prof = S.mp->profile && level == 0 && S.top == 0 && kid >= 0;
    hipEvent_t a = nullptr, b = nullptr;
    if (prof && (a = pool_event(S))) S_HIP(S, hipEventRecord(a, S.compute));
    const int rc = mgx_slab_launch(&op, S.compute);
    if (rc != MGX_OK) return slab_fail(S, rc, "slab %d level %d: %s", S.rank, level, mgx_last_error());
    if (prof && a && (b = pool_event(S))) {
        S_HIP(S, hipEventRecord(b, S.compute));
        S.recs.push_back({kid, a, b, (long)(o.jhi - o.jlo + o.jhi2 - o.jlo2) * L.m});
    }
    return MGX_OK;
}

// local [jlo, jhi) of the columns to produce: the slab's own columns widened by `extra` on
// each side (or, with coarse_extra = e >= 0, exactly the fine columns whose restriction gives
// the slab's own coarse columns widened by e), clipped to the interior columns 1..m-2
void col_range(const SLevel &L, int extra, int coarse_extra, int &jlo, int &jhi)
{
    int glo, ghi;
    if (coarse_extra >= 0) {
        glo = L.b0 + 1 - 2 * coarse_extra;
        ghi = L.b1 + 2 * coarse_extra;
        if (coarse_extra == 0) glo = L.b0;
    } else {
        glo = L.b0 - extra;
        ghi = L.b1 + extra;
    }
    glo = std::max(glo, 1);
    ghi = std::min(ghi, L.m - 1);
    jlo = glo - L.lo;
    jhi = ghi - L.lo;
}

// ---- neighbour exchange ------------------------------------------------------------------

// Before a launch overwrites columns a neighbour pulled in the last exchange: make the
// compute stream wait until both neighbours' pulls of that exchange have finished.
int fence_pulls(Slab &S)
{
    mgx_mplan *mp = S.mp;
    const long x = S.xcount;
    if (x == 0) return MGX_OK;
    for (int n : {S.rank - 1, S.rank + 1}) {
        if (n < 0 || n >= mp->R) continue;
        if (!spin_until(mp, mp->pulled[n], x)) return MGX_ERR_STATE;
        S_HIP(S, hipStreamWaitEvent(S.compute, mp->slabs[n]->done[x & 1], 0));
    }
    return MGX_OK;
}

// Everything `items` names has been enqueued on the compute stream: tell the neighbours.
int exchange_begin(Slab &S, const std::vector<XItem> &items)
{
    mgx_mplan *mp = S.mp;
    const long x = ++S.xcount;
    S.pending = items;
    if (S.rank == 0) mp->nexchanges++;
    // test hook: slab MGX_TEST_FAIL_SLAB gives up at the second exchange of a call's life, the way
    // a HIP error would, so that the other slabs' reaction and the plan's refusal can be tested
    if (const char *e = getenv("MGX_TEST_FAIL_SLAB"))
        if (*e && atoi(e) == S.rank && x >= 2)
            return slab_fail(S, MGX_ERR_STATE, "slab %d: failure injected by MGX_TEST_FAIL_SLAB", S.rank);
    S_HIP(S, hipEventRecord(S.ready[x & 1], S.compute));
    mp->posted[S.rank].store(x, std::memory_order_release);
    return MGX_OK;
}

// Pull the neighbours' columns on the comm stream; the compute stream continues after them.
int exchange_end(Slab &S)
{
    mgx_mplan *mp = S.mp;
    const long x = S.xcount;
    // the pulls overwrite halo columns the launches enqueued before `ready` may still read
    S_HIP(S, hipStreamWaitEvent(S.comm, S.ready[x & 1], 0));
    for (int side = 0; side < 2; side++) {
        const int n = side == 0 ? S.rank - 1 : S.rank + 1;
        if (n < 0 || n >= mp->R) continue;
        if (!spin_until(mp, mp->posted[n], x)) return MGX_ERR_STATE;
        Slab &N = *mp->slabs[n];
        S_HIP(S, hipStreamWaitEvent(S.comm, N.ready[x & 1], 0));
        for (const XItem &it : S.pending) {
            SLevel &L = S.lv[it.level];
            SLevel &Q = N.lv[it.level];
            double *dst = it.is_f ? L.f : L.v[L.cur];
            const double *src = it.is_f ? Q.f : Q.v[L.cur];  // every slab flips its buffers alike
            const size_t bytes = sizeof(double) * (size_t)it.width * (size_t)L.m;
            size_t doff, soff;
            if (side == 0) {  // my columns [o0-w, o0) <- the lower neighbour's last own columns
                doff = (size_t)(L.o0 - it.width) * L.m;
                soff = (size_t)(Q.o1 - it.width) * Q.m;
            } else {          // my columns [o1, o1+w) <- the upper neighbour's first own columns
                doff = (size_t)L.o1 * L.m;
                soff = (size_t)Q.o0 * Q.m;
            }
            S_HIP(S, hipMemcpyPeerAsync(dst + doff, S.dev, src + soff, N.dev, bytes, S.comm));
        }
    }
    S_HIP(S, hipEventRecord(S.done[x & 1], S.comm));
    mp->pulled[S.rank].store(x, std::memory_order_release);
    S_HIP(S, hipStreamWaitEvent(S.compute, S.done[x & 1], 0));
    S.pending.clear();
    return MGX_OK;
}

int halo_exchange(Slab &S, const std::vector<XItem> &items)
{
    S_TRY(exchange_begin(S, items));
    return exchange_end(S);
}

// ---- the tail ------------------------------------------------------------------------------

// Gather an array of the tail's top level: every slab's own columns, from that slab's staging
// array of this gather's parity, into `dst` (this slab's copy of the whole level).
int gather_tail(Slab &S, double *dst)
{
    mgx_mplan *mp = S.mp;
    const int nd = mp->nd, mt = mp->dims[nd];
    if (mp->R == 1) return MGX_OK;  // the single slab writes the tail's arrays directly
    const long t = ++S.tcount;
    if (S.rank == 0) mp->ngathers++;
    S_HIP(S, hipEventRecord(S.tready[t & 1], S.compute));
    mp->tposted[S.rank].store(t, std::memory_order_release);
    const std::vector<int> &tb = mp->bounds[nd];
    for (int q = 0; q < mp->R; q++) {
        const int k = (S.rank + q) % mp->R;  // own columns first, then round the ring
        Slab &P = *mp->slabs[k];
        const size_t off = (size_t)tb[k] * mt, bytes = sizeof(double) * (size_t)(tb[k + 1] - tb[k]) * mt;
        if (k != S.rank) {
            if (!spin_until(mp, mp->tposted[k], t)) return MGX_ERR_STATE;
            S_HIP(S, hipStreamWaitEvent(S.compute, P.tready[t & 1], 0));
        }
        S_HIP(S, hipMemcpyPeerAsync(dst + off, S.dev, P.tailsrc[t & 1] + off, P.dev, bytes, S.compute));
    }
    return MGX_OK;
}

// Gather the tail's right-hand side and run one cycle of the tail from a zero iterate.
int run_tail(Slab &S, const CycleArgs &a)
{
    S_TRY(gather_tail(S, S.tail_f));
    if (mgx_plan_vcycle_async(S.tail, a.Ndown, a.Nup, a.omega, 1) != MGX_OK)
        return slab_fail(S, MGX_ERR_STATE, "slab %d tail: %s", S.rank, mgx_last_error());
    return MGX_OK;
}

// where this slab puts its columns of a tail array for the NEXT gather: the staging array of
// that gather's parity, or (a single slab) the tail's own array `direct`
double *tail_staging(Slab &S, double *direct)
{
    return S.mp->R > 1 ? S.tailsrc[(S.tcount + 1) & 1] : direct;
}
double *tail_target(Slab &S) { return tail_staging(S, S.tail_f); }

void coarse_of(Slab &S, int l, SLevel *&C, CoarseShape &cs)
{
    mgx_mplan *mp = S.mp;
    if (l + 1 < mp->nd) {
        C = &S.lv[l + 1];
        cs.mc = C->m;
        cs.nccols = C->ncols;
        cs.gc0 = C->lo;
    } else {
        C = nullptr;
        cs.mc = mp->dims[mp->nd];
        cs.nccols = mp->dims[mp->nd];
        cs.gc0 = 0;
    }
}

// [pro] -> ns sweeps -> [res: 1 restrict, 2 norm] on the slab's own columns as launches of at
// most MGX_SLAB_MAX_SWEEPS sweeps; every launch that stores the iterate flips the level's
// ping-pong buffers, and between two launches of a chain the halo is exchanged.
int chain(Slab &S, int level, int kid, int pro, int ns, int res, int zero_in, double omega, const CoarseShape &cs,
          const double *coarse_v, double *coarse_f, unsigned long long *slot)
{
    SLevel &L = S.lv[level];
    int remaining = ns;
    bool first = true;
    int jlo, jhi;
    col_range(L, 0, -1, jlo, jhi);
    for (;;) {
        const int k = std::min(remaining, MGX_SLAB_MAX_SWEEPS);
        remaining -= k;
        const bool last = remaining == 0;
        const int pr = first ? pro : 0, rs = last ? res : 0, zi = first ? zero_in : 0;
        if (k > 0 || pr || rs) {
            OpSpec o;
            o.nsweeps = k;
            o.pro_stage = pr ? 0 : -1;
            o.norm_stage = rs == 2 ? k : -1;
            o.restrict_out = rs == 1 ? 1 : 0;
            o.zero_in = zi;
            o.store = (k > 0 || pr) ? 1 : 0;
            o.v_in = L.v[L.cur];
            o.v_out = L.v[L.cur ^ 1];
            o.cs = cs;
            o.coarse_v = pr ? coarse_v : nullptr;
            o.coarse_f = rs == 1 ? coarse_f : nullptr;
            o.norm_slot = rs == 2 ? slot : nullptr;
            o.jlo = jlo;
            o.jhi = jhi;
            o.omega = omega;
            S_TRY(fence_pulls(S));
            S_TRY(launch(S, level, kid, o));
            if (o.store) {
                L.cur ^= 1;
                if (!last) S_TRY(halo_exchange(S, {XItem{level, false, L.W}}));
            }
        }
        first = false;
        if (remaining <= 0) break;
    }
    return MGX_OK;
}

int normalize_boundary(Slab &S)
{
    SLevel &L = S.lv[S.top];
    // it rewrites rows 0 and m-1 of every stored column, also of the columns the neighbours
    // are pulling in the exchange posted last: after their pulls
    S_TRY(fence_pulls(S));
    for (int k = 0; k < 2; k++)
        S_HIP(S, launch_slab_boundary_plus_zero(L.v[k], L.m, L.ncols, L.jfix_lo, L.jfix_hi, S.compute));
    return MGX_OK;
}

// ---- the general schedule: a chain per leg, an exchange after every launch --------------------
int cycle_general(Slab &S, const CycleArgs &a, unsigned long long *slot, bool first)
{
    mgx_mplan *mp = S.mp;
    const int nd = mp->nd, top = a.top;
    for (int l = top; l < nd; l++) {  // main.cpp:317-328
        SLevel &L = S.lv[l];
        SLevel *C;
        CoarseShape cs;
        coarse_of(S, l, C, cs);
        double *cf = C ? C->f : tail_target(S);
        std::vector<XItem> items;
        if (l > top && a.Ndown == 0) {  // the iterate is zero here: f -> coarse f, nothing stored
            OpSpec o;
            o.restrict_out = 1;
            o.zero_in = 1;
            o.store = 0;
            o.cs = cs;
            o.coarse_f = cf;
            o.omega = a.omega;
            col_range(L, 0, -1, o.jlo, o.jhi);
            S_TRY(fence_pulls(S));
            S_TRY(launch(S, l, MGX_KERNEL_FUSED_DOWN, o));
        } else {
            S_TRY(chain(S, l, MGX_KERNEL_FUSED_DOWN, 0, a.Ndown, 1, l > top ? 1 : 0, a.omega, cs, nullptr, cf, nullptr));
            items.push_back(XItem{l, false, L.W});
        }
        if (C) items.push_back(XItem{l + 1, true, C->W});
        if (!items.empty()) S_TRY(halo_exchange(S, items));
    }
    S_TRY(run_tail(S, a));
    if (first) S_TRY(normalize_boundary(S));
    for (int l = nd - 1; l >= top; l--) {  // main.cpp:335-346
        SLevel &L = S.lv[l];
        SLevel *C;
        CoarseShape cs;
        coarse_of(S, l, C, cs);
        const double *cv = C ? C->v[C->cur] : S.tail_v;
        const int zero_in = (l > top && a.Ndown == 0) ? 1 : 0;
        const int res = (l == top && slot) ? 2 : 0;
        S_TRY(chain(S, l, MGX_KERNEL_FUSED_UP, 1, a.Nup, res, zero_in, a.omega, cs, cv, nullptr, slot));
        S_TRY(halo_exchange(S, {XItem{l, false, L.W}}));
    }
    return MGX_OK;
}

// ---- the lean schedule: one exchange and one gather per cycle ----------------------------------
std::vector<XItem> lean_items(Slab &S, bool with_v)
{
    std::vector<XItem> items;
    if (with_v) items.push_back(XItem{S.top, false, W0});
    if (S.mp->nd - S.top >= 2) items.push_back(XItem{S.top + 1, true, S.Ef[1]});
    return items;
}

// A level-0 launch in up to three parts, so that the exchange that follows it overlaps most
// of it: the columns the neighbours are waiting for first, then the exchange is posted,
// then the interior columns, then the neighbours' columns are pulled.
int launch_level0(Slab &S, int kid, OpSpec o, const std::vector<XItem> &items, bool flip)
{
    mgx_mplan *mp = S.mp;
    const int top = S.top;
    SLevel &L = S.lv[top];
    int jlo, jhi;
    col_range(L, 0, -1, jlo, jhi);
    const int ef1 = mp->nd - top >= 2 ? S.Ef[1] : 0;
    int B = std::max(2 * ef1, W0);
    B += B & 1;
    const bool lo_nb = L.lo > 0, hi_nb = L.hi < L.m;
    int ilo = jlo, ihi = jhi;
    S_TRY(fence_pulls(S));
    bool split = false;
    if (mp->overlap && (lo_nb || hi_nb) && (jhi - jlo) >= 4 * B) {
        split = true;
        // the columns next to both neighbours go out in ONE launch of two column ranges
        if (lo_nb) ilo = jlo + B;
        if (hi_nb) ihi = jhi - B;
        if (lo_nb && hi_nb) {
            o.jlo = jlo;
            o.jhi = ilo;
            o.jlo2 = ihi;
            o.jhi2 = jhi;
        } else {
            o.jlo = lo_nb ? jlo : ihi;
            o.jhi = lo_nb ? ilo : jhi;
        }
        S_TRY(launch(S, top, -1, o));
        o.jlo2 = o.jhi2 = 0;
    }
    o.jlo = ilo;
    o.jhi = ihi;
    // (whether a slab splits its launch depends on its own width, so the number of exchanges
    // must not depend on the split: nothing to exchange means no exchange on either path)
    if (split && !items.empty()) {
        if (flip) L.cur ^= 1;  // from here on v[cur] is the buffer being written
        S_TRY(exchange_begin(S, items));
        S_TRY(launch(S, top, kid, o));
        return exchange_end(S);
    }
    S_TRY(launch(S, top, kid, o));
    if (flip) L.cur ^= 1;
    if (!items.empty()) S_TRY(halo_exchange(S, items));
    return MGX_OK;
}

int lean_down0(Slab &S, const CycleArgs &a)
{
    SLevel &L = S.lv[a.top];
    SLevel *C;
    CoarseShape cs;
    coarse_of(S, a.top, C, cs);
    OpSpec o;
    o.nsweeps = a.Ndown;
    o.restrict_out = 1;
    o.store = a.Ndown > 0 ? 1 : 0;
    o.omega = a.omega;
    o.v_in = L.v[L.cur];
    o.v_out = L.v[L.cur ^ 1];
    o.cs = cs;
    o.coarse_f = C ? C->f : tail_target(S);
    return launch_level0(S, MGX_KERNEL_FUSED_DOWN, o, lean_items(S, o.store != 0), o.store != 0);
}

// levels 1..nd-1 down, the tail, levels nd-1..1 up: no communication but the tail's gather
int lean_coarse(Slab &S, const CycleArgs &a)
{
    mgx_mplan *mp = S.mp;
    const int nd = mp->nd, top = a.top;
    for (int l = top + 1; l < nd; l++) {
        SLevel &L = S.lv[l];
        SLevel *C;
        CoarseShape cs;
        coarse_of(S, l, C, cs);
        OpSpec o;
        o.nsweeps = a.Ndown;
        o.restrict_out = 1;
        o.zero_in = 1;
        o.store = 0;
        o.omega = a.omega;
        o.cs = cs;
        o.coarse_f = C ? C->f : tail_target(S);
        col_range(L, 0, l + 1 < nd ? S.Ef[l + 1 - top] : 0, o.jlo, o.jhi);
        S_TRY(launch(S, l, MGX_KERNEL_FUSED_DOWN, o));
    }
    S_TRY(run_tail(S, a));
    for (int l = nd - 1; l > top; l--) {
        SLevel &L = S.lv[l];
        SLevel *C;
        CoarseShape cs;
        coarse_of(S, l, C, cs);
        OpSpec o;
        o.nsweeps = a.Ndown + a.Nup;
        o.pro_stage = a.Ndown;
        o.zero_in = 1;
        o.store = 1;
        o.omega = a.omega;
        o.v_out = L.v[0];
        o.cs = cs;
        o.coarse_v = C ? C->v[C->cur] : S.tail_v;
        col_range(L, S.Ev[l - top], -1, o.jlo, o.jhi);
        L.cur = 0;
        S_TRY(launch(S, l, MGX_KERNEL_FUSED_UP, o));
    }
    return MGX_OK;
}

int lean_up0(Slab &S, const CycleArgs &a, unsigned long long *slot, bool next_down)
{
    SLevel &L = S.lv[a.top];
    SLevel *C;
    CoarseShape cs;
    coarse_of(S, a.top, C, cs);
    OpSpec o;
    o.pro_stage = 0;
    o.norm_stage = a.Nup;
    o.store = 1;
    o.omega = a.omega;
    o.v_in = L.v[L.cur];
    o.v_out = L.v[L.cur ^ 1];
    o.cs = cs;
    o.coarse_v = C ? C->v[C->cur] : S.tail_v;
    o.norm_slot = slot;
    if (next_down) {
        o.nsweeps = a.Nup + a.Ndown;
        o.restrict_out = 1;
        o.coarse_f = C ? C->f : tail_target(S);
        return launch_level0(S, MGX_KERNEL_FUSED_UPDOWN, o, lean_items(S, true), true);
    }
    o.nsweeps = a.Nup;
    return launch_level0(S, MGX_KERNEL_FUSED_UP, o, {XItem{a.top, false, W0}}, true);
}

// Nv cycles on one slab (main.cpp:309-354): what each slab's thread runs.
int slab_cycles(Slab &S, const CycleArgs &a)
{
    mgx_mplan *mp = S.mp;
    if (hipSetDevice(S.dev) != hipSuccess) return slab_fail(S, MGX_ERR_HIP, "slab %d: hipSetDevice(%d)", S.rank, S.dev);
    S.top = a.top;
    const bool was_tune = set_tune_now(a.tune);
    struct Restore {
        bool was;
        ~Restore() { set_tune_now(was); }
    } restore{was_tune};
    const int Nv = a.Nv;
    if (Nv + 1 > S.nslots) {
        if (S.slots) S_HIP(S, hipFree(S.slots));
        S.slots = nullptr;
        S_HIP(S, hipMalloc((void **)&S.slots, sizeof(unsigned long long) * (size_t)(Nv + 1)));
        S.nslots = Nv + 1;
    }
    S_HIP(S, hipMemsetAsync(S.slots, 0, sizeof(unsigned long long) * (size_t)(Nv + 1), S.compute));
    if (S.rank == 0)
        while ((int)S.cyc.size() < Nv + 1) {
            hipEvent_t e;
            S_HIP(S, hipEventCreate(&e));
            S.cyc.push_back(e);
        }
    const bool lean = !a.pure_jacobi && a.Ndown + a.Nup >= 1 && a.Ndown + a.Nup <= MAX_LEAN;
    if (lean) {
        halo_plan(mp->nd - a.top, a.Ndown, a.Nup, S.Ev, S.Ef);
        for (int l = a.top + 1; l < mp->nd; l++)
            if (std::max(S.Ef[l - a.top], S.Ev[l - a.top]) > S.lv[l].W)
                return slab_fail(S, MGX_ERR_STATE, "level %d halo %d exceeds the allocated %d", l,
                                 std::max(S.Ef[l - a.top], S.Ev[l - a.top]), S.lv[l].W);
    }
    const CoarseShape none;
    // main.cpp:307: the norm of the initial residual (halo columns of v must be current)
    S_TRY(halo_exchange(S, {XItem{a.top, false, W0}}));
    S_TRY(chain(S, a.top, MGX_KERNEL_RESNORM, 0, 0, 2, 0, a.omega, none, nullptr, nullptr, S.slots + 0));
    if (S.rank == 0) S_HIP(S, hipEventRecord(S.cyc[0], S.compute));
    for (int it = 0; it < Nv; it++) {
        unsigned long long *slot = S.slots + it + 1;
        if (a.pure_jacobi) {  // main.cpp:348 and :351 in one pass
            S_TRY(chain(S, a.top, MGX_KERNEL_FUSED_UP, 0, 1, 2, 0, a.omega, none, nullptr, nullptr, slot));
            S_TRY(halo_exchange(S, {XItem{a.top, false, W0}}));
        } else if (lean) {
            if (it == 0) S_TRY(lean_down0(S, a));  // later cycles: the previous level-0 launch did the down leg
            S_TRY(lean_coarse(S, a));
            if (it == 0) S_TRY(normalize_boundary(S));
            S_TRY(lean_up0(S, a, slot, it + 1 < Nv));
        } else {
            S_TRY(cycle_general(S, a, slot, it == 0));
        }
        if (S.rank == 0) S_HIP(S, hipEventRecord(S.cyc[it + 1], S.compute));
    }
    return MGX_OK;
}

// EXTENSION (the reference has no full-multigrid driver): the composition mgx_plan_fmg runs
// on one device, over the slabs.  The right-hand side is restricted to every level
// (coarsen, main.cpp:201-221), the Dirichlet values of v are injected, every slab's tail runs
// its own full multigrid (identical bits), and on the way up each distributed level starts
// from refine() of the level below and runs a.Nv cycles on the hierarchy below it.  Only the
// boundary values of v are used.
int slab_fmg(Slab &S, const CycleArgs &a)
{
    mgx_mplan *mp = S.mp;
    if (hipSetDevice(S.dev) != hipSuccess) return slab_fail(S, MGX_ERR_HIP, "slab %d: hipSetDevice(%d)", S.rank, S.dev);
    const int nd = mp->nd, R = mp->R, g = S.rank, mt = mp->dims[nd];
    const std::vector<int> &tb = mp->bounds[nd];
    S.top = 0;
    for (int l = 0; l < nd; l++) {
        SLevel &L = S.lv[l];
        SLevel *C;
        CoarseShape cs;
        coarse_of(S, l, C, cs);
        // the residual of a zero iterate is the right-hand side itself: restriction of f
        OpSpec o;
        o.restrict_out = 1;
        o.zero_in = 1;
        o.store = 0;
        o.omega = a.omega;
        o.cs = cs;
        const double *src = L.v[L.cur];
        col_range(L, 0, -1, o.jlo, o.jhi);
        if (!C) {
            // The tail's iterate (its Dirichlet values) is gathered first, its right-hand side
            // second: a staging array is written only after the gather that follows the one it
            // was last read in (a peer may still be reading the array of the latest gather).
            double *vstage = tail_staging(S, S.tail_v);
            const size_t tbytes = sizeof(double) * (size_t)mt * mt;
            S_HIP(S, hipMemsetAsync(S.tail_v, 0, tbytes, S.compute));
            if (R > 1)
                S_HIP(S, hipMemsetAsync(vstage + (size_t)tb[g] * mt, 0, sizeof(double) * (size_t)(tb[g + 1] - tb[g]) * mt,
                                        S.compute));
            S_HIP(S, launch_slab_inject_boundary(src, L.m, L.ncols, 2 * tb[g] - L.lo, vstage, mt, tb[g], tb[g + 1] - tb[g],
                                                 g == 0, g == R - 1, S.compute));
            S_TRY(gather_tail(S, S.tail_v));
            o.coarse_f = tail_target(S);
            S_TRY(launch(S, l, -1, o));
            continue;
        }
        o.coarse_f = C->f;
        S_TRY(fence_pulls(S));
        S_TRY(launch(S, l, -1, o));
        S_TRY(halo_exchange(S, {XItem{l + 1, true, C->W}}));
        S_TRY(fence_pulls(S));
        S_HIP(S, hipMemsetAsync(C->v[0], 0, C->bytes(), S.compute));
        S_HIP(S, launch_slab_inject_boundary(src, L.m, L.ncols, 2 * C->b0 - L.lo, C->v[0], C->m, C->o0, C->b1 - C->b0,
                                             C->lo == 0, C->hi == C->m, S.compute));
        C->cur = 0;
        S_TRY(halo_exchange(S, {XItem{l + 1, false, C->W}}));
        S_HIP(S, hipMemcpyAsync(C->v[1], C->v[0], C->bytes(), hipMemcpyDeviceToDevice, S.compute));
    }
    S_TRY(gather_tail(S, S.tail_f));
    if (mgx_plan_commit(S.tail) != MGX_OK ||
        mgx_plan_fmg(S.tail, a.Nv, a.Ndown, a.Nup, a.omega, nullptr, nullptr) != MGX_OK)
        return slab_fail(S, MGX_ERR_STATE, "slab %d tail full multigrid: %s", S.rank, mgx_last_error());
    S.tail_v = mgx_plan_v_dev(S.tail);  // the tail's ping-pong buffers may have traded places
    for (int l = nd - 1; l >= 0; l--) {
        SLevel &L = S.lv[l];
        SLevel *C;
        CoarseShape cs;
        coarse_of(S, l, C, cs);
        // initial guess: the interpolated solution of the level below (boundary values stay)
        OpSpec o;
        o.pro_stage = 0;
        o.zero_in = 1;
        o.store = 1;
        o.omega = a.omega;
        o.v_out = L.v[L.cur];
        o.cs = cs;
        o.coarse_v = C ? C->v[C->cur] : S.tail_v;
        col_range(L, 0, -1, o.jlo, o.jhi);
        S_TRY(fence_pulls(S));
        S_TRY(launch(S, l, -1, o));
        // from here on the level below holds coarse-grid corrections: zero, zero boundary
        if (C) {
            S_HIP(S, hipMemsetAsync(C->v[0], 0, C->bytes(), S.compute));
            S_HIP(S, hipMemsetAsync(C->v[1], 0, C->bytes(), S.compute));
            C->cur = 0;
        } else {
            S_HIP(S, hipMemsetAsync(S.tail_v, 0, sizeof(double) * (size_t)mt * mt, S.compute));
            if (mgx_plan_commit(S.tail) != MGX_OK)
                return slab_fail(S, MGX_ERR_STATE, "slab %d tail reset: %s", S.rank, mgx_last_error());
        }
        CycleArgs sub = a;
        sub.top = l;
        S_TRY(slab_cycles(S, sub));
    }
    return MGX_OK;
}

void destroy_slab(Slab &S)
{
    if (hipSetDevice(S.dev) != hipSuccess) return;
    if (S.compute) (void)hipStreamSynchronize(S.compute);
    if (S.comm) (void)hipStreamSynchronize(S.comm);
    if (S.tail) mgx_plan_destroy(S.tail);
    for (SLevel &L : S.lv) {
        if (L.f) (void)hipFree(L.f);
        if (L.v[0]) (void)hipFree(L.v[0]);
        if (L.v[1]) (void)hipFree(L.v[1]);
    }
    for (int k = 0; k < 2; k++) {
        if (S.tailsrc[k] && S.tailsrc[k] != S.tail_f) (void)hipFree(S.tailsrc[k]);
        if (S.ready[k]) (void)hipEventDestroy(S.ready[k]);
        if (S.done[k]) (void)hipEventDestroy(S.done[k]);
        if (S.tready[k]) (void)hipEventDestroy(S.tready[k]);
    }
    if (S.slots) (void)hipFree(S.slots);
    for (hipEvent_t e : S.cyc) (void)hipEventDestroy(e);
    for (hipEvent_t e : S.evpool) (void)hipEventDestroy(e);
    if (S.compute) (void)hipStreamDestroy(S.compute);
    if (S.comm) (void)hipStreamDestroy(S.comm);
}

int slab_alloc(Slab &S, void **p, size_t nbytes)
{
    S_HIP(S, hipMalloc(p, nbytes));
    S_HIP(S, hipMemsetAsync(*p, 0, nbytes, S.compute));
    S.bytes += nbytes;
    return MGX_OK;
}

// device-side set-up of one slab (on its device)
int slab_create(Slab &S, const mgx_options *opt)
{
    mgx_mplan *mp = S.mp;
    const int nd = mp->nd, R = mp->R, g = S.rank;
    S_HIP(S, hipSetDevice(S.dev));
    S_HIP(S, hipStreamCreateWithFlags(&S.compute, hipStreamNonBlocking));
    S_HIP(S, hipStreamCreateWithFlags(&S.comm, hipStreamNonBlocking));
    for (int k = 0; k < 2; k++) {
        S_HIP(S, hipEventCreateWithFlags(&S.ready[k], hipEventDisableTiming));
        S_HIP(S, hipEventCreateWithFlags(&S.done[k], hipEventDisableTiming));
        S_HIP(S, hipEventCreateWithFlags(&S.tready[k], hipEventDisableTiming));
    }
    // halo widths the arrays must hold: for the whole hierarchy and for every hierarchy that
    // starts at one of its levels (full multigrid cycles on those too)
    std::vector<int> Evm(nd + 1, 0), Efm(nd + 1, 0);
    for (int t = 0; t < nd; t++) {
        std::vector<int> ev, ef;
        halo_capacity(nd - t, ev, ef);
        for (int j = 0; j <= nd - t; j++) {
            Evm[t + j] = std::max(Evm[t + j], ev[j]);
            Efm[t + j] = std::max(Efm[t + j], ef[j]);
        }
    }
    S.lv.resize(nd);
    for (int l = 0; l < nd; l++) {
        SLevel &L = S.lv[l];
        L.m = mp->dims[l];
        L.dx = mp->hs[l];
        L.W = l == 0 ? W0 : std::max(W0, std::max(Efm[l], Evm[l]));
        L.b0 = mp->bounds[l][g];
        L.b1 = mp->bounds[l][g + 1];
        L.lo = std::max(0, L.b0 - L.W);
        L.hi = std::min(L.m, L.b1 + L.W);
        L.ncols = L.hi - L.lo;
        L.o0 = L.b0 - L.lo;
        L.o1 = L.b1 - L.lo;
        L.jfix_lo = L.lo == 0 ? 0 : -1;
        L.jfix_hi = L.hi == L.m ? L.ncols - 1 : -1;
        if (L.b1 - L.b0 < L.W && R > 1)
            return slab_fail(S, MGX_ERR_ARG, "slab %d owns %d < %d columns of level %d", g, L.b1 - L.b0, L.W, l);
        S_TRY(slab_alloc(S, (void **)&L.f, L.bytes()));
        S_TRY(slab_alloc(S, (void **)&L.v[0], L.bytes()));
        S_TRY(slab_alloc(S, (void **)&L.v[1], L.bytes()));
    }
    mgx_options o;
    memset(&o, 0, sizeof(o));
    o.struct_size = (int)sizeof(o);
    o.device = S.dev;
    o.stream = S.compute;
    o.schedule = MGX_SCHEDULE_FUSED;
    o.verbose = (opt && g == 0) ? opt->verbose : 0;
    o.coarse_solve = opt ? opt->coarse_solve : MGX_COARSE_REFERENCE;
    if (mgx_plan_create(&S.tail, mp->dims[nd], mp->hs[nd], mp->coarsest, &o) != MGX_OK)
        return slab_fail(S, MGX_ERR_STATE, "slab %d tail plan: %s", g, mgx_last_error());
    S.bytes += mgx_plan_bytes(S.tail);
    S.tail_f = mgx_plan_f_dev(S.tail);
    S.tail_v = mgx_plan_v_dev(S.tail);
    if (R > 1) {
        const size_t nb = sizeof(double) * (size_t)mp->dims[nd] * mp->dims[nd];
        S_TRY(slab_alloc(S, (void **)&S.tailsrc[0], nb));
        S_TRY(slab_alloc(S, (void **)&S.tailsrc[1], nb));
    } else {
        S.tailsrc[0] = S.tailsrc[1] = S.tail_f;
    }
    S_HIP(S, hipStreamSynchronize(S.compute));
    return MGX_OK;
}

// Which levels are split and who owns what: levels 0..nd-1 are distributed, level nd is the
// top of the tail (a plan of its own, whose finest level must be odd: an even coarsest level
// cannot be the whole tail).  tail_m <= 0 picks the default: every slab repeats the tail, so
// its cost does not shrink with the number of slabs while a distributed level's does -- the
// more slabs, the smaller the tail (2049 on one slab, 1025 on several).  bounds[l][g] is the
// first global column slab g owns of level l (bounds[l][R] = size of the level).
int partition(const std::vector<int> &dims, int R, int tail_m, int &tail_m_used, int &nd_out,
              std::vector<std::vector<int>> &bounds)
{
    const int depth = (int)dims.size() - 1;
    if (tail_m <= 0) tail_m = R == 1 ? 2049 : 1025;
    tail_m_used = tail_m;
    const int max_nd = depth - ((dims.back() % 2) == 0 ? 1 : 0);
    int nd = 0;
    while (nd < max_nd && dims[nd] > tail_m) {
        // would every slab still own a halo's worth of columns of every level, for every
        // sub-hierarchy, if one more level were distributed?
        std::vector<int> Ev(nd + 2, 0), Ef(nd + 2, 0);
        for (int s = 0; s <= nd; s++) {
            std::vector<int> ev, ef;
            halo_capacity(nd + 1 - s, ev, ef);
            for (int j = 0; j <= nd + 1 - s; j++) {
                Ev[s + j] = std::max(Ev[s + j], ev[j]);
                Ef[s + j] = std::max(Ef[s + j], ef[j]);
            }
        }
        const int base = (dims[nd + 1] - 2) / R;
        bool fits = base >= 1;
        for (int l = 0; l <= nd; l++) {
            const long own = (long)base * (1L << (nd + 1 - l)) - 1;
            const int need = l == 0 ? W0 : std::max(W0, std::max(Ef[l], Ev[l]));
            fits = fits && own >= need;
        }
        if (R > 1 && !fits) break;
        nd++;
    }
    if (nd == 0)
        return api_fail(MGX_ERR_ARG, "a %dx%d grid is not split over %d slab(s) with tail_m=%d: solve it with "
                                     "mgx_plan_* on one device", dims[0], dims[0], R, tail_m);
    nd_out = nd;
    const int mt = dims[nd];
    bounds.assign(nd + 1, std::vector<int>(R + 1, 0));
    for (int g = 1; g < R; g++) bounds[nd][g] = 1 + (int)(((long)g * (mt - 2)) / R);
    bounds[nd][R] = mt;
    for (int l = nd - 1; l >= 0; l--) {
        for (int g = 1; g < R; g++) bounds[l][g] = 2 * bounds[l + 1][g] - 1;
        bounds[l][R] = dims[l];
    }
    return MGX_OK;
}

// the sizes of the levels (main.cpp:286,293) or an error code
int level_sizes(int m, int coarsest, std::vector<int> &dims)
{
    if (m < 3) return api_fail(MGX_ERR_ARG, "grid size %d is below 3", m);
    if ((m % 2) == 0) return api_fail(MGX_ERR_ARG, "Input has even rows and columns! (m=%d)", m);
    if (coarsest < 1) return api_fail(MGX_ERR_ARG, "coarsest must be >= 1 (got %d)", coarsest);
    const int depth = mgx_depth(m, coarsest);
    if (depth < 0) return api_fail(MGX_ERR_HIERARCHY, "depth %d < 0 for m=%d coarsest=%d", depth, m, coarsest);
    dims.assign(1, m);
    for (int d = 1; d <= depth; d++) {
        dims.push_back((dims.back() - 1) / 2 + 1);  // main.cpp:293
        if (dims[d] < 3 || ((dims[d] % 2) == 0 && d < depth))
            return api_fail(MGX_ERR_HIERARCHY, "level %d would be %dx%d (needs odd >= 3): the reference indexes out "
                                               "of range there", d, dims[d], dims[d]);
    }
    return MGX_OK;
}

int poisoned_fail(const mgx_mplan *mp, const char *who)
{
    return api_fail(MGX_ERR_STATE, "%s: an earlier call on this partitioned plan failed (%s); its slabs are out of step, "
                                   "destroy the plan", who, mp->poison_msg.c_str());
}

// Run fn on every slab, one thread per slab; the calling thread's current device is left as
// it was.  `sticky`: a failure leaves the slabs out of step with each other (counters,
// ping-pong indices, work still queued), so the plan is marked and refuses further calls.
template <class F>
int on_every_slab(mgx_mplan *mp, F fn, bool sticky = true)
{
    std::vector<std::thread> th;
    int spawned = 1;
    try {
        for (int g = 1; g < mp->R; g++) {
            th.emplace_back([mp, g, &fn] { (void)fn(*mp->slabs[g]); });
            spawned++;
        }
    } catch (...) {
        // a thread could not be created: the slabs that did start must not wait for the others
        mp->failed.store(1, std::memory_order_release);
    }
    int prev = -1;
    const bool have_prev = hipGetDevice(&prev) == hipSuccess;
    (void)fn(*mp->slabs[0]);
    for (std::thread &t : th) t.join();
    if (have_prev) (void)hipSetDevice(prev);
    int rc = MGX_OK;
    std::string msg;
    if (spawned < mp->R) {
        rc = MGX_ERR_NOMEM;
        msg = "a slab's host thread could not be created";
    }
    for (int g = 0; g < mp->R && rc == MGX_OK; g++)
        if (mp->slabs[g]->rc != MGX_OK) {
            rc = mp->slabs[g]->rc;
            msg = mp->slabs[g]->err;
        }
    if (rc == MGX_OK && mp->failed.load()) {
        rc = MGX_ERR_STATE;
        msg = mp->failed.load() == 2 ? "a slab's hand-shake with its neighbour timed out (MGX_MPLAN_TIMEOUT)"
                                     : "a slab thread gave up";
    }
    for (int k = 0; k < mp->R; k++) {
        mp->slabs[k]->rc = MGX_OK;
        mp->slabs[k]->err.clear();
    }
    mp->failed.store(0);
    if (rc == MGX_OK) return MGX_OK;
    if (sticky) {
        // let what the slabs did enqueue drain, so that nothing of this call is still running
        // when the caller reacts (the plan stays unusable: the slabs disagree on where they are)
        for (auto &S : mp->slabs)
            if (hipSetDevice(S->dev) == hipSuccess) {
                if (S->compute) (void)hipStreamSynchronize(S->compute);
                if (S->comm) (void)hipStreamSynchronize(S->comm);
            }
        if (have_prev) (void)hipSetDevice(prev);
        mp->poisoned = 1;
        mp->poison_msg = msg;
    }
    return api_fail(rc, "%s", msg.c_str());
}

} // namespace

extern "C" {

int mgx_mplan_partition(int m, int coarsest, int nslabs, int tail_m, int *distributed_levels, int *bounds, int bounds_len)
{
    if (nslabs < 1 || nslabs > 64) return api_fail(MGX_ERR_ARG, "mgx_mplan_partition: nslabs %d outside 1..64", nslabs);
    std::vector<int> dims;
    int rc = level_sizes(m, coarsest, dims);
    if (rc != MGX_OK) return rc;
    int used = 0, nd = 0;
    std::vector<std::vector<int>> b;
    rc = partition(dims, nslabs, tail_m, used, nd, b);
    if (rc != MGX_OK) return rc;
    if (distributed_levels) *distributed_levels = nd;
    if (bounds) {
        if (bounds_len < (nd + 1) * (nslabs + 1))
            return api_fail(MGX_ERR_ARG, "mgx_mplan_partition: bounds needs %d entries", (nd + 1) * (nslabs + 1));
        for (int l = 0; l <= nd; l++)
            for (int g = 0; g <= nslabs; g++) bounds[l * (nslabs + 1) + g] = b[l][g];
    }
    return MGX_OK;
}

int mgx_halo_plan(int distributed_levels, int Ndown, int Nup, int *Ev, int *Ef)
{
    if (distributed_levels < 1 || Ndown < 0 || Nup < 0 || Ndown + Nup < 1 || Ndown + Nup > MAX_LEAN || !Ev || !Ef)
        return api_fail(MGX_ERR_ARG, "mgx_halo_plan: bad argument");
    std::vector<int> ev, ef;
    halo_plan(distributed_levels, Ndown, Nup, ev, ef);
    for (int l = 0; l <= distributed_levels; l++) {
        Ev[l] = ev[l];
        Ef[l] = ef[l];
    }
    return MGX_OK;
}

int mgx_mplan_create(mgx_mplan **out, int m, double dx, int coarsest, int nslabs, const int *devices,
                     const mgx_options *opt)
{
    if (!out) return api_fail(MGX_ERR_ARG, "mgx_mplan_create: null output pointer");
    *out = nullptr;
    if (m < 3) return api_fail(MGX_ERR_ARG, "mgx_mplan_create: grid size %d is below 3", m);
    if ((m % 2) == 0) return api_fail(MGX_ERR_ARG, "mgx_mplan_create: Input has even rows and columns! (m=%d)", m);
    if (coarsest < 1) return api_fail(MGX_ERR_ARG, "mgx_mplan_create: coarsest must be >= 1 (got %d)", coarsest);
    if (!(dx > 0.0)) return api_fail(MGX_ERR_ARG, "mgx_mplan_create: dx must be positive");
    if (nslabs < 1 || nslabs > 64) return api_fail(MGX_ERR_ARG, "mgx_mplan_create: nslabs %d outside 1..64", nslabs);
    if (opt && opt->struct_size != (int)sizeof(mgx_options))
        return api_fail(MGX_ERR_ARG, "mgx_mplan_create: options.struct_size %d != %d", opt->struct_size,
                        (int)sizeof(mgx_options));
    if (opt && opt->schedule == MGX_SCHEDULE_STEPWISE)
        return api_fail(MGX_ERR_ARG, "mgx_mplan_create: the partitioned driver runs the fused schedule only");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return api_fail(MGX_ERR_HIP, "mgx_mplan_create: no HIP device is visible");
    const int depth = mgx_depth(m, coarsest);
    if (depth < 0) return api_fail(MGX_ERR_HIERARCHY, "mgx_mplan_create: depth %d < 0 for m=%d coarsest=%d", depth, m, coarsest);
    std::unique_ptr<mgx_mplan> mp(new (std::nothrow) mgx_mplan());
    if (!mp) return api_fail(MGX_ERR_NOMEM, "mgx_mplan_create: out of host memory");
    const int R = nslabs;
    mp->m = m;
    mp->dx = dx;
    mp->coarsest = coarsest;
    mp->depth = depth;
    mp->R = R;
    mp->dims.push_back(m);
    mp->hs.push_back(dx);
    for (int d = 1; d <= depth; d++) {
        mp->dims.push_back((mp->dims.back() - 1) / 2 + 1);  // main.cpp:293
        mp->hs.push_back(mp->hs.back() * 2.0);              // main.cpp:297
        if (mp->dims[d] < 3 || ((mp->dims[d] % 2) == 0 && d < depth))
            return api_fail(MGX_ERR_HIERARCHY, "mgx_mplan_create: level %d would be %dx%d (needs odd >= 3): the "
                                               "reference indexes out of range there", d, mp->dims[d], mp->dims[d]);
    }
    int tail_m = opt ? opt->tail_m : 0;
    if (tail_m <= 0) {
        const char *e = getenv("MGX_TAIL_M");
        tail_m = e ? atoi(e) : 0;
    }
    {
        const char *e = getenv("MGX_SLAB_OVERLAP");
        mp->overlap = (e && atoi(e) == 0) ? 0 : 1;
    }
    {
        const int rc = partition(mp->dims, R, tail_m, mp->tail_m, mp->nd, mp->bounds);
        if (rc != MGX_OK) return rc;
    }
    mp->posted.reset(new std::atomic<long>[R]);
    mp->pulled.reset(new std::atomic<long>[R]);
    mp->tposted.reset(new std::atomic<long>[R]);
    for (int g = 0; g < R; g++) {
        mp->posted[g].store(0);
        mp->pulled[g].store(0);
        mp->tposted[g].store(0);
        std::unique_ptr<Slab> S(new Slab());
        S->mp = mp.get();
        S->rank = g;
        S->dev = devices ? devices[g] : g % ndev;
        if (S->dev < 0 || S->dev >= ndev)
            return api_fail(MGX_ERR_ARG, "mgx_mplan_create: device %d of slab %d is not one of the %d visible", S->dev, g, ndev);
        mp->slabs.push_back(std::move(S));
    }
    // peer access between the devices in use (a pull from a device without it is staged by the runtime)
    int prev = -1;
    (void)hipGetDevice(&prev);
    for (int a = 0; a < R; a++)
        for (int b = 0; b < R; b++) {
            const int da = mp->slabs[a]->dev, db = mp->slabs[b]->dev;
            if (da == db) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, da, db) == hipSuccess && can && hipSetDevice(da) == hipSuccess) {
                const hipError_t e = hipDeviceEnablePeerAccess(db, 0);
                if (e != hipSuccess) (void)hipGetLastError();  // already enabled is fine
            }
        }
    if (prev >= 0) (void)hipSetDevice(prev);
    const int rc = on_every_slab(mp.get(), [opt](Slab &S) { return slab_create(S, opt); });
    if (rc != MGX_OK) {
        mgx_mplan_destroy(mp.release());
        return rc;
    }
    *out = mp.release();
    return MGX_OK;
}

int mgx_mplan_destroy(mgx_mplan *mp)
{
    if (!mp) return MGX_OK;
    int prev = -1;
    (void)hipGetDevice(&prev);
    // every slab's streams first: a peer copy queued on one slab's stream reads another slab's
    // arrays, so nothing may be freed while any stream still holds work
    for (auto &S : mp->slabs)
        if (hipSetDevice(S->dev) == hipSuccess) {
            if (S->compute) (void)hipStreamSynchronize(S->compute);
            if (S->comm) (void)hipStreamSynchronize(S->comm);
        }
    for (auto &S : mp->slabs) destroy_slab(*S);
    if (prev >= 0) (void)hipSetDevice(prev);
    delete mp;
    return MGX_OK;
}

int mgx_mplan_nslabs(const mgx_mplan *mp) { return mp ? mp->R : MGX_ERR_ARG; }
int mgx_mplan_depth(const mgx_mplan *mp) { return mp ? mp->depth : MGX_ERR_ARG; }
int mgx_mplan_distributed_levels(const mgx_mplan *mp) { return mp ? mp->nd : MGX_ERR_ARG; }
int mgx_mplan_tail_m(const mgx_mplan *mp) { return mp ? mp->dims[mp->nd] : MGX_ERR_ARG; }

size_t mgx_mplan_bytes(const mgx_mplan *mp, int slab)
{
    if (!mp || slab < 0 || slab >= mp->R) return 0;
    return mp->slabs[slab]->bytes;
}

int mgx_mplan_slab(mgx_mplan *mp, int slab, mgx_slab_info *info)
{
    if (!mp || !info || slab < 0 || slab >= mp->R) return api_fail(MGX_ERR_ARG, "mgx_mplan_slab: bad argument");
    if (info->struct_size != (int)sizeof(mgx_slab_info))
        return api_fail(MGX_ERR_ARG, "mgx_mplan_slab: struct_size %d != %d", info->struct_size, (int)sizeof(mgx_slab_info));
    Slab &S = *mp->slabs[slab];
    SLevel &L = S.lv[0];
    info->device = S.dev;
    info->m = L.m;
    info->col_first = L.lo;
    info->ncols = L.ncols;
    info->own_first = L.b0;
    info->own_count = L.b1 - L.b0;
    info->f_dev = L.f;
    info->v_dev = L.v[L.cur];
    info->stream = S.compute;
    return MGX_OK;
}

int mgx_mplan_commit(mgx_mplan *mp)
{
    if (!mp) return api_fail(MGX_ERR_ARG, "mgx_mplan_commit: null plan");
    if (mp->poisoned) return poisoned_fail(mp, "mgx_mplan_commit");
    return on_every_slab(mp, [](Slab &S) {
        S_HIP(S, hipSetDevice(S.dev));
        SLevel &L = S.lv[0];
        // both ping-pong buffers carry the same Dirichlet values
        S_HIP(S, hipMemcpyAsync(L.v[L.cur ^ 1], L.v[L.cur], L.bytes(), hipMemcpyDeviceToDevice, S.compute));
        S_HIP(S, hipStreamSynchronize(S.compute));
        return (int)MGX_OK;
    });
}

int mgx_mplan_upload(mgx_mplan *mp, const double *f_host, const double *v_host)
{
    if (!mp) return api_fail(MGX_ERR_ARG, "mgx_mplan_upload: null plan");
    if (mp->poisoned) return poisoned_fail(mp, "mgx_mplan_upload");
    return on_every_slab(mp, [f_host, v_host](Slab &S) {
        S_HIP(S, hipSetDevice(S.dev));
        SLevel &L = S.lv[0];
        const size_t off = (size_t)L.lo * L.m;  // a slab's columns, halo included, are one contiguous block
        if (f_host) S_HIP(S, hipMemcpyAsync(L.f, f_host + off, L.bytes(), hipMemcpyHostToDevice, S.compute));
        if (v_host) {
            S_HIP(S, hipMemcpyAsync(L.v[L.cur], v_host + off, L.bytes(), hipMemcpyHostToDevice, S.compute));
            S_HIP(S, hipMemcpyAsync(L.v[L.cur ^ 1], L.v[L.cur], L.bytes(), hipMemcpyDeviceToDevice, S.compute));
        }
        S_HIP(S, hipStreamSynchronize(S.compute));
        return (int)MGX_OK;
    });
}

int mgx_mplan_download(mgx_mplan *mp, double *v_host)
{
    if (!mp || !v_host) return api_fail(MGX_ERR_ARG, "mgx_mplan_download: null argument");
    if (mp->poisoned) return poisoned_fail(mp, "mgx_mplan_download");
    return on_every_slab(mp, [v_host](Slab &S) {
        S_HIP(S, hipSetDevice(S.dev));
        SLevel &L = S.lv[0];
        // owned columns; the two physical boundary columns belong to the first and last slab
        const size_t n = sizeof(double) * (size_t)(L.b1 - L.b0) * L.m;
        S_HIP(S, hipMemcpyAsync(v_host + (size_t)L.b0 * L.m, L.v[L.cur] + (size_t)L.o0 * L.m, n, hipMemcpyDeviceToHost,
                                S.compute));
        S_HIP(S, hipStreamSynchronize(S.compute));
        return (int)MGX_OK;
    });
}

static int mplan_run(mgx_mplan *mp, const CycleArgs &a, double *resnorms, double *times)
{
    const int Nv = a.Nv;
    timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    const int rc = on_every_slab(mp, [&a, mp, t0](Slab &S) {
        const int rc_ = a.fmg ? slab_fmg(S, a) : slab_cycles(S, a);
        if (rc_ != MGX_OK) return rc_;
        if (S.rank == 0) {
            timespec te;
            clock_gettime(CLOCK_MONOTONIC, &te);
            mp->enqueue_seconds = (double)(te.tv_sec - t0.tv_sec) + 1e-9 * (double)(te.tv_nsec - t0.tv_nsec);
        }
        S_HIP(S, hipStreamSynchronize(S.compute));
        S_HIP(S, hipStreamSynchronize(S.comm));
        return (int)MGX_OK;
    });
    clock_gettime(CLOCK_MONOTONIC, &t1);
    mp->host_seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    if (rc != MGX_OK) return rc;
    int prev = -1;
    (void)hipGetDevice(&prev);
    int ret = MGX_OK;
    std::vector<unsigned long long> keys((size_t)Nv + 1, 0ull), part((size_t)Nv + 1);
    for (auto &Sp : mp->slabs) {
        Slab &S = *Sp;
        if (hipSetDevice(S.dev) != hipSuccess ||
            hipMemcpy(part.data(), S.slots, sizeof(unsigned long long) * part.size(), hipMemcpyDeviceToHost) != hipSuccess) {
            ret = api_fail(MGX_ERR_HIP, "mgx_mplan_cycles: reading the norms of slab %d failed", S.rank);
            break;
        }
        for (int k = 0; k <= Nv; k++) keys[k] = std::max(keys[k], part[k]);  // the keys order like the values
        for (const Slab::Rec &r : S.recs) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess && S.rank == 0) {
                mp->prof_ms[r.kid] += (double)ms;
                mp->prof_n[r.kid] += 1;
                mp->prof_pts[r.kid] += (double)r.pts;
            }
        }
        S.recs.clear();
        S.evused = 0;
    }
    if (ret == MGX_OK && resnorms)
        for (int k = 0; k <= Nv; k++) resnorms[k] = norm_decode(keys[k]);
    if (ret == MGX_OK && times) {
        Slab &S = *mp->slabs[0];
        (void)hipSetDevice(S.dev);
        times[0] = 0;  // main.cpp:308
        for (int k = 1; k <= Nv; k++) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, S.cyc[k - 1], S.cyc[k]) != hipSuccess) ms = 0.f;
            times[k] = times[k - 1] + 1e-3 * (double)ms;  // whole-cycle device time of slab 0
        }
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    return ret;
}

int mgx_mplan_cycles(mgx_mplan *mp, int Nv, int Ndown, int Nup, double omega, int pure_jacobi, double *resnorms,
                     double *times)
{
    if (!mp) return api_fail(MGX_ERR_ARG, "mgx_mplan_cycles: null plan");
    if (mp->poisoned) return poisoned_fail(mp, "mgx_mplan_cycles");
    if (Nv < 0 || Ndown < 0 || Nup < 0) return api_fail(MGX_ERR_ARG, "mgx_mplan_cycles: negative count");
    CycleArgs a;
    a.Nv = Nv;
    a.Ndown = Ndown;
    a.Nup = Nup;
    a.omega = omega;
    a.pure_jacobi = pure_jacobi;
    return mplan_run(mp, a, resnorms, times);
}

int mgx_mplan_fmg(mgx_mplan *mp, int ncyc, int Ndown, int Nup, double omega, double *resnorms, double *times)
{
    if (!mp) return api_fail(MGX_ERR_ARG, "mgx_mplan_fmg: null plan");
    if (mp->poisoned) return poisoned_fail(mp, "mgx_mplan_fmg");
    if (ncyc < 0 || Ndown < 0 || Nup < 0) return api_fail(MGX_ERR_ARG, "mgx_mplan_fmg: negative count");
    CycleArgs a;
    a.Nv = ncyc;
    a.Ndown = Ndown;
    a.Nup = Nup;
    a.omega = omega;
    a.fmg = true;
    return mplan_run(mp, a, resnorms, times);
}

int mgx_mplan_tune(mgx_mplan *mp, int Ndown, int Nup, double omega)
{
    if (!mp) return api_fail(MGX_ERR_ARG, "mgx_mplan_tune: null plan");
    if (mp->poisoned) return poisoned_fail(mp, "mgx_mplan_tune");
    if (Ndown < 0 || Nup < 0) return api_fail(MGX_ERR_ARG, "mgx_mplan_tune: negative count");
    // keep the level-0 iterate, run three cycles with the measurement forced, put it back
    std::vector<double *> keep((size_t)mp->R, nullptr);
    auto free_keep = [&] {
        int prev = -1;
        (void)hipGetDevice(&prev);
        for (int g = 0; g < mp->R; g++)
            if (keep[g] && hipSetDevice(mp->slabs[g]->dev) == hipSuccess) {
                (void)hipFree(keep[g]);
                keep[g] = nullptr;
            }
        if (prev >= 0) (void)hipSetDevice(prev);
    };
    // (a failed allocation leaves the slabs in step: nothing was enqueued but copies out of v)
    int rc = on_every_slab(mp, [&keep](Slab &S) {
        S_HIP(S, hipSetDevice(S.dev));
        SLevel &L = S.lv[0];
        S_HIP(S, hipMalloc((void **)&keep[S.rank], L.bytes()));
        S_HIP(S, hipMemcpyAsync(keep[S.rank], L.v[L.cur], L.bytes(), hipMemcpyDeviceToDevice, S.compute));
        S_HIP(S, hipStreamSynchronize(S.compute));
        return (int)MGX_OK;
    }, false);
    if (rc != MGX_OK) {
        free_keep();
        return rc;
    }
    CycleArgs a;
    a.Nv = 3;
    a.Ndown = Ndown;
    a.Nup = Nup;
    a.omega = omega;
    a.tune = true;
    const int prof = mp->profile;
    mp->profile = 0;
    rc = mplan_run(mp, a, nullptr, nullptr);
    mp->profile = prof;
    if (rc != MGX_OK) {  // the plan is poisoned; nothing to put back into
        free_keep();
        return rc;
    }
    rc = on_every_slab(mp, [&keep](Slab &S) {
        S_HIP(S, hipSetDevice(S.dev));
        SLevel &L = S.lv[0];
        S_HIP(S, hipMemcpyAsync(L.v[0], keep[S.rank], L.bytes(), hipMemcpyDeviceToDevice, S.compute));
        S_HIP(S, hipMemcpyAsync(L.v[1], keep[S.rank], L.bytes(), hipMemcpyDeviceToDevice, S.compute));
        L.cur = 0;
        S_HIP(S, hipStreamSynchronize(S.compute));
        return (int)MGX_OK;
    });
    free_keep();
    return rc;
}

int mgx_mplan_profile(mgx_mplan *mp, int enable)
{
    if (!mp) return api_fail(MGX_ERR_ARG, "mgx_mplan_profile: null plan");
    mp->profile = enable ? 1 : 0;
    for (int k = 0; k < MGX_KERNEL_COUNT; k++) {
        mp->prof_ms[k] = 0;
        mp->prof_n[k] = 0;
        mp->prof_pts[k] = 0;
    }
    mp->nexchanges = 0;
    mp->ngathers = 0;
    return MGX_OK;
}

int mgx_mplan_profile_get(const mgx_mplan *mp, int kernel_id, long *count, double *total_ms, double *points)
{
    if (!mp || kernel_id < 0 || kernel_id >= MGX_KERNEL_COUNT)
        return api_fail(MGX_ERR_ARG, "mgx_mplan_profile_get: bad argument");
    if (count) *count = mp->prof_n[kernel_id];
    if (total_ms) *total_ms = mp->prof_ms[kernel_id];
    if (points) *points = mp->prof_pts[kernel_id];
    return MGX_OK;
}

int mgx_mplan_stats(const mgx_mplan *mp, long *exchanges, long *gathers, double *host_seconds,
                    double *enqueue_seconds)
{
    if (!mp) return api_fail(MGX_ERR_ARG, "mgx_mplan_stats: null plan");
    if (exchanges) *exchanges = mp->nexchanges;
    if (gathers) *gathers = mp->ngathers;
    if (host_seconds) *host_seconds = mp->host_seconds;
    if (enqueue_seconds) *enqueue_seconds = mp->enqueue_seconds;
    return MGX_OK;
}

} // extern "C"
