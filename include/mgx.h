/*
 * mgx.h -- C ABI of libmgx, the MI355X (gfx950) implementation of the 2D
 * Poisson geometric-multigrid V-cycle of RoyJames/MultiGrid (main.cpp).
 *
 * Conventions (all entry points)
 *   - Arrays are fp64, COLUMN-MAJOR with leading dimension m, exactly the
 *     storage DTMutableDoubleArray::Pointer() exposes
 *     (reference DTSource/DTDoubleArray.h:38-49,125-128,184): A(i,j) = a[i + j*m].
 *   - Grids are square and odd-sized (reference main.cpp:408-417); h = dx
 *     (dy is carried by the reference but never used, main.cpp:180,263).
 *   - Return value 0 = success, negative = MGX_ERR_*; mgx_last_error() gives
 *     the message of the calling thread's last failure.  The library never
 *     calls exit() and never prints unless options.verbose is set.
 *   - "dev" pointers are HIP device pointers on the current device; "host"
 *     pointers are ordinary host memory.  No torch types cross this boundary.
 *   - Arithmetic is strict IEEE fp64 in the reference's left-to-right
 *     evaluation order with no FMA contraction, so results are bit-identical
 *     to the reference built without fast-math.
 */
#ifndef MGX_H
#define MGX_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGX_VERSION 210

#define MGX_OK 0
#define MGX_ERR_ARG (-1)      /* bad size / null pointer / non-square / even size   */
#define MGX_ERR_HIERARCHY (-2) /* depth < 0 or an even-sized level above the coarsest  */
#define MGX_ERR_HIP (-3)      /* a HIP runtime call failed                           */
#define MGX_ERR_NOMEM (-4)
#define MGX_ERR_STATE (-5)    /* plan used before data was set, etc.                 */

/* Solver options.  Zero-initialise, set struct_size = sizeof(mgx_options),
 * then change what you need; NULL means all defaults. */
typedef struct mgx_options {
    int struct_size;
    int device;       /* HIP device ordinal, -1 = current device (default)                     */
    void *stream;     /* hipStream_t to run on, NULL = the null stream                          */
    int schedule;     /* 0 = default (fused), 1 = one kernel per reference operator (stepwise),
                         2 = fused streaming kernels                                            */
    int verbose;      /* 1 = print the reference's "Direct solver not defined" message etc.     */
    int coarse_solve; /* MGX_COARSE_REFERENCE (0, default) or MGX_COARSE_EXACT, see below           */
    int ngpus;        /* mgx_solve2d: 0 or 1 = one device (default); N > 1 = the grid is split into
                         N j-slabs, slab k on device k modulo the number of visible devices
                         (mgx_mplan_*, below).  A grid too small to split, or the stepwise
                         schedule (which exists on one device only), runs on one device.        */
    int tail_m;       /* partitioned runs: largest level (points per side) that every slab
                         holds whole instead of a share of; 0 = default (1025, MGX_TAIL_M)      */
    int flags;        /* MGX_FLAG_* bits, see below                                             */
    int reserved[4];
} mgx_options;

/* options.flags
 *   MGX_FLAG_ZERO_GUESS  mgx_solve2d: the caller states that every INTERIOR entry of v is
 *                        +0.0 (what main.cpp:420-421 passes: u = 0, then the boundary is
 *                        filled).  Only the boundary of v is then read from the host array,
 *                        which saves one of the three host<->device transfers of the call.
 *                        The result is the same bits; a wrong statement gives the solution
 *                        for the zero guess.  Without the flag a big guess (>= 64 MB) is
 *                        looked at on the host while f uploads, and one whose interior is
 *                        all +0.0 takes the same short cut; the flag saves the look. */
#define MGX_FLAG_ZERO_GUESS 1

#define MGX_SCHEDULE_DEFAULT 0
#define MGX_SCHEDULE_STEPWISE 1
#define MGX_SCHEDULE_FUSED 2

/* What happens on the coarsest grid (main.cpp:331).
 *   MGX_COARSE_REFERENCE  the reference's direct_solve (main.cpp:153-169): the closed form
 *                         for a 3x3 grid, and NOTHING for any other size (the reference
 *                         prints "Direct solver not defined" and carries on).
 *   MGX_COARSE_EXACT      EXTENSION, no counterpart in the reference's MultiGrid(): a
 *                         coarsest grid other than 3x3 (up to 65x65, odd or even) is solved
 *                         exactly -- the 5-point system of the reference's unused
 *                         getSparseSol() (main.cpp:79-120) -- by a sine-transform solver on
 *                         the GPU.  Results agree with a Cholesky solve to rounding
 *                         (about 1e-13 relative), not bit for bit.  A 3x3 grid still takes
 *                         the closed form. */
#define MGX_COARSE_REFERENCE 0
#define MGX_COARSE_EXACT 1

int mgx_version(void);
const char *mgx_last_error(void);

/* Number of coarsenings the reference performs for an m x m fine grid:
 * int(log2(1.0f*(m-1)/coarsest) + 0.5f) in float arithmetic
 * (reference main.cpp:286).  Level d has (m_{d-1}-1)/2+1 points per side
 * (main.cpp:293). */
int mgx_depth(int m, int coarsest);

/* ------------------------------------------------------------------------
 * Whole-solver entry point: replaces
 *   MGOutputs MultiGrid(gridtype &prob, int Nv, int Ndown, int Nup,
 *                       double omega, int coarsest, bool pureJacobi)
 * (reference main.cpp:283-357, called at main.cpp:445).
 *   f        host, m*n : right-hand side  (prob.f.DoubleData().Pointer())
 *   v        host, m*n : initial guess with the Dirichlet values already on
 *                        the boundary; overwritten with the solution in place,
 *                        as the reference does through shared storage
 *                        (main.cpp:288, DTSource/DTMesh2D.cpp:553-559)
 *   dx                 : prob.v.Grid().dx()
 *   resnorms host, Nv+1: infinity norm of the fine residual before the first
 *                        cycle and after each cycle (main.cpp:307,351-353)
 *   times    host, Nv+1: times[0]=0 (main.cpp:308); times[k] = times[k-1] +
 *                        seconds of cycle k.  The reference sums wall time
 *                        around its relax() calls only (main.cpp:320-322,
 *                        342-344,350).  MGX_SCHEDULE_STEPWISE reports the same
 *                        quantity as device time: HIP events around the
 *                        launches of every relax() of the cycle (0 for
 *                        pure_jacobi, as in the reference).  The fused
 *                        schedules (the default) cannot separate smoothing
 *                        from the transfers it is fused with and report
 *                        whole-cycle device time instead; with the level-0
 *                        up+down fusion the event that closes cycle k also
 *                        covers the level-0 down leg of cycle k+1.
 * Stopping rule: exactly Nv cycles, no tolerance test (main.cpp:309).
 * ------------------------------------------------------------------------ */
int mgx_solve2d(const double *f, double *v, int m, int n, double dx,
                int Nv, int Ndown, int Nup, double omega, int coarsest, int pure_jacobi,
                double *resnorms, double *times, const mgx_options *opt);

/* ------------------------------------------------------------------------
 * Device-resident plan: the level hierarchy of MultiGrid() (main.cpp:286-300)
 * allocated once in HBM, for callers that keep f and v on the GPU or run many
 * solves of one size.  mgx_solve2d() is create + upload + cycles + download.
 * ------------------------------------------------------------------------ */
typedef struct mgx_plan mgx_plan;

int mgx_plan_create(mgx_plan **plan, int m, double dx, int coarsest, const mgx_options *opt);
int mgx_plan_destroy(mgx_plan *plan);
int mgx_plan_depth(const mgx_plan *plan);
int mgx_plan_level_m(const mgx_plan *plan, int level);
size_t mgx_plan_bytes(const mgx_plan *plan); /* HBM held by the plan */

/* Fine-level device buffers (m*m doubles each) the caller may fill directly,
 * e.g. from a kernel of its own.  After writing f or v through these pointers call
 * mgx_plan_commit(): the plan refreshes its other copies of the boundary values
 * and forgets what it may have computed ahead from the old data (see
 * mgx_plan_cycles).  mgx_plan_v_dev() names the buffer that holds the current
 * iterate: the plan's iterate buffers trade places inside mgx_plan_cycles()
 * (no copy of the grid), so ask again after it returns.  After a series of short calls the
 * iterate may not be in memory yet (see mgx_plan_cycles): mgx_plan_v_dev() then enqueues the one
 * launch that produces it on the plan's stream before it answers (NULL if that launch fails).  mgx_plan_f_dev() never
 * changes, and mgx_plan_vcycle_async() leaves mgx_plan_v_dev() as it is. */
double *mgx_plan_f_dev(mgx_plan *plan);
double *mgx_plan_v_dev(mgx_plan *plan);
int mgx_plan_commit(mgx_plan *plan);

/* Device <-> plan copies on the plan's stream (asynchronous): f_dev / v_dev are
 * the caller's own m*m device arrays; either may be NULL to leave that field
 * as it is.  set_dev includes the commit. */
int mgx_plan_set_dev(mgx_plan *plan, const double *f_dev, const double *v_dev);
int mgx_plan_get_dev(mgx_plan *plan, double *v_dev);

/* Host <-> plan copies (synchronous with respect to the plan's stream).  The arrays may be
 * ordinary pageable memory (what DTMutableDoubleArray holds, DTDoubleArray.cpp:39): big ones
 * move through pinned staging buffers on several host threads (MGX_IO_THREADS, default 8),
 * which about doubles the rate of a plain hipMemcpy.  upload_boundary sets the iterate to
 * zero in the interior and to v_host on the four boundary lines (all it reads of v_host). */
int mgx_plan_upload(mgx_plan *plan, const double *f_host, const double *v_host);
int mgx_plan_upload_boundary(mgx_plan *plan, const double *v_host);
int mgx_plan_download(mgx_plan *plan, double *v_host);

/* Nv V(Ndown,Nup) cycles on the resident data; the V-cycle loop of
 * main.cpp:309-354.  resnorms/times: host, Nv+1 entries, may be NULL.
 * Coarse-level v is zero again after every cycle (main.cpp:345), so repeated calls
 * continue from the current fine v exactly as one longer call would.  The
 * per-cycle residual norm is always computed on the device, as the reference
 * does; NULL only skips the copy to the host.
 * A caller that stops on a tolerance runs one cycle (or a few) per call.  Such calls cost
 * what one long call costs: a call's last launch may also run the NEXT call's opening sweeps
 * and restriction, and a next call that continues it (same Ndown >= 1, Nup, omega with
 * Ndown + Nup <= 4, no mgx_plan_commit / set / upload in between) starts from there: one
 * level-0 launch per cycle instead of two.  The end-of-cycle iterate is then produced only
 * when somebody asks for it (mgx_plan_v_dev / get_dev / download: one more launch).  A call
 * ends that way once two calls in a row have continued their predecessor, and a call of at
 * most four cycles also when earlier short calls on the same plan were continued that way (the
 * plan keeps count, so the series on a plan's second right-hand side pays nothing extra; a
 * caller who stops calling that way pays the unused launch twice at most).  Same bits either way. */
int mgx_plan_cycles(mgx_plan *plan, int Nv, int Ndown, int Nup, double omega, int pure_jacobi,
                    double *resnorms, double *times);

/* EXTENSION for harnesses that want "V-cycles to a tolerance": the reference's stopping rule
 * is a fixed number of cycles (main.cpp:309), so this is a loop AROUND mgx_plan_cycles(), not
 * a change of it.  Cycles run in batches of `batch`; after each batch the history is read and
 * the loop ends when some cycle's norm is at most rel_tol times the initial one
 * (MGX_STOP_TOLERANCE), when three cycles in a row fail to bring the norm 10 % below the
 * smallest seen so far (MGX_STOP_FLOOR: in fp64 the residual f - Lap_h(u) cannot fall below about
 * eps * 8/h^2 * max|u|, which at 22529^2 is 4e-8 of a unit right-hand side, and hovers
 * there; a cycle that merely converges slowly is not a floor), or after
 * max_cycles (MGX_STOP_MAX_CYCLES).  The cycles of the last batch are all run: *cycles_done is a
 * multiple of `batch` (or max_cycles).  resnorms / times: max_cycles+1 entries, filled up to
 * *cycles_done; the same bits as one mgx_plan_cycles() call of *cycles_done cycles. */
#define MGX_STOP_MAX_CYCLES 0
#define MGX_STOP_TOLERANCE 1
#define MGX_STOP_FLOOR 2
int mgx_plan_cycles_until(mgx_plan *plan, int max_cycles, int Ndown, int Nup, double omega, double rel_tol, int batch,
                          double *resnorms, double *times, int *cycles_done, int *reason);

/* The stream the plan's work is enqueued on from now on (options.stream at creation).  Work
 * already enqueued on the previous stream is ordered before anything that follows. */
int mgx_plan_set_stream(mgx_plan *plan, void *stream);

/* EXTENSION (the reference has no full-multigrid driver): nested iteration with the
 * reference's own operators.  The right-hand side is restricted to every level
 * (coarsen, main.cpp:201-221), the Dirichlet values of v are injected (coarse boundary
 * point k <- fine boundary point 2k), the coarsest grid is solved (options.coarse_solve),
 * and on the way up every level starts from refine() (main.cpp:223-253) of the level
 * below and runs ncyc V(Ndown,Nup) cycles of MultiGrid() on the hierarchy below it.  The
 * interior of v on entry is ignored.  resnorms/times (ncyc+1 each, or NULL): the history of
 * the finest level's cycles.  Synchronous like mgx_plan_cycles(); bit-identical to the same
 * composition of the reference's operators on the CPU. */
int mgx_plan_fmg(mgx_plan *plan, int ncyc, int Ndown, int Nup, double omega, double *resnorms, double *times);

/* Per-launch timing of the FINE-LEVEL kernels, for roofline reporting: when
 * enabled, every level-0 kernel launch inside mgx_plan_cycles() is bracketed by
 * two HIP events on the plan's stream; the totals accumulate per kernel id
 * until the next mgx_plan_profile() call.  No counterpart in the reference
 * (which times relax() with gettimeofday, DTSource/DTTimer.h:13-19). */
#define MGX_KERNEL_JACOBI 0
#define MGX_KERNEL_RESIDUAL 1
#define MGX_KERNEL_RESTRICT 2
#define MGX_KERNEL_PROLONG_ADD 3
#define MGX_KERNEL_RESNORM 4
#define MGX_KERNEL_FUSED_DOWN 5
#define MGX_KERNEL_FUSED_UP 6
#define MGX_KERNEL_FUSED_UPDOWN 7 /* level 0: up leg of a cycle + down leg of the next */
#define MGX_KERNEL_COUNT 8
int mgx_plan_profile(mgx_plan *plan, int enable);
int mgx_plan_profile_get(const mgx_plan *plan, int kernel_id, long *count, double *total_ms);
/* enable = 2 brackets the launches of EVERY level (a few microseconds of event records per
 * small launch: for a breakdown, not for a timed run); mgx_plan_profile_level() then gives, per
 * level, the launches that START there and their device time -- a launch that covers a group
 * of small levels, or the LDS-resident bottom of the cycle, is booked on its finest level. */
int mgx_plan_profile_level(const mgx_plan *plan, int level, long *count, double *total_ms);
const char *mgx_kernel_name(int kernel_id);

/* ------------------------------------------------------------------------
 * Per-operator entry points (device pointers, run on `stream`, asynchronous
 * unless a host output is written).  They exist so that each kernel can be
 * parity-tested against the reference operator it replaces.
 * ------------------------------------------------------------------------ */

/* relax(gridtype&, int Niter, double omega), main.cpp:171-199: Niter damped
 * Jacobi sweeps on v (in/out), boundary untouched. */
int mgx_relax(double *v_dev, const double *f_dev, int m, double dx, int niter, double omega, void *stream);

/* residual(const gridtype&), main.cpp:255-281: r = f - Lap_h(v) on the
 * interior, 0 on the boundary; r_dev receives all m*m entries. */
int mgx_residual(const double *v_dev, const double *f_dev, double *r_dev, int m, double dx, void *stream);

/* coarsen(fine, coarse), main.cpp:201-221, including the corner-term quirk at
 * :218 (fine(2i+1,2j-1) twice, fine(2i-1,2j+1) never).  mc = (mf-1)/2+1. */
int mgx_restrict(const double *fine_dev, int mf, double *coarse_dev, int mc, void *stream);

/* refine(coarse, fine), main.cpp:223-253: bilinear interpolation, boundary 0;
 * fine_dev receives all mf*mf entries. */
int mgx_refine(const double *coarse_dev, int mc, double *fine_dev, int mf, void *stream);

/* refine() followed by `cur += refined` (main.cpp:339-341 with
 * DTSource/DTArrayTemplates.h:157-167) without the temporary. */
int mgx_prolong_add(const double *coarse_dev, int mc, double *v_dev, int mf, void *stream);

/* direct_solve(gridtype&), main.cpp:153-169.  Returns 1 (not an error) and
 * leaves v unchanged when m != 3, as the reference does after printing. */
int mgx_direct3x3(double *v_dev, const double *f_dev, int m, double dx, void *stream);

/* EXTENSION (MGX_COARSE_EXACT as a single operator): the interior of the m x m grid v_dev
 * (4 <= m <= 65) becomes the exact solution of 4 v(i,j) - (four neighbours) = -dx^2 f(i,j)
 * for the boundary values v_dev holds: the system of getSparseSol(), main.cpp:79-110.
 * Synchronous (builds and frees its tables). */
int mgx_coarse_exact(double *v_dev, const double *f_dev, int m, double dx, void *stream);

/* calcNorm(array), main.cpp:147-150 = max(Maximum, -Minimum) with
 * DTSource/DTDoubleArray.cpp:883-899,923-939 semantics (NaNs ignored).
 * Synchronises the stream and writes the scalar to *out_host. */
int mgx_norminf(const double *a_dev, size_t len, double *out_host, void *stream);

/* calcNorm(residual(p)) without materialising the residual
 * (main.cpp:307,351). */
int mgx_residual_norm(const double *v_dev, const double *f_dev, int m, double dx, double *out_host, void *stream);

/* calcNorm(residual(p)) like mgx_residual_norm(), for callers that cannot allocate.  This
 * entry point has no counterpart in the reference; it exists for code under HIP graph
 * capture, tight serving loops and harnesses that forbid hipMalloc/hipFree: it allocates
 * nothing, synchronises nothing and leaves the result in DEVICE memory (*norm_dev, one
 * double), stream-ordered on `stream`.  Its partial results live in a fixed 64 KiB device
 * array inside the library (one per GPU, held until the process ends), which bounds the grid
 * it accepts: m <= 65537, larger sizes are rejected with MGX_ERR_ARG.  All calls on one GPU
 * share that array: calls that overlap on one GPU (different streams or threads) race, and so
 * do replays of a captured graph on two streams.  The first use on a device looks up the
 * array's address (hipGetSymbolAddress, which may load the library's device code): that must
 * not happen under stream capture.  mgx_plan_create() and mgx_mplan_create() do the look-up
 * for the device they run on; a caller that creates no plan makes one call outside capture
 * first. */
int mgx_residual_norm_noalloc(const double *v_dev, const double *f_dev, int m, double dx, double *norm_dev,
                              void *stream);

/* ---- fused legs (what the default schedule runs on every level) ------------ */

/* Down leg of one level in one pass over HBM: relax(nsweeps) + residual +
 * coarsen, main.cpp:321-327.  v_dev in/out (m*m); coarse_f_dev (mc*mc,
 * mc=(m-1)/2+1) receives coarsen(residual(v)), boundary 0. */
int mgx_fused_down(double *v_dev, const double *f_dev, double *coarse_f_dev, int m, double dx, int nsweeps,
                   double omega, void *stream);

/* Up leg of one level in one pass: refine + `cur += refined` + relax(nsweeps),
 * main.cpp:337-344, and, if norm_host is not NULL, calcNorm(residual) of the
 * result (main.cpp:351) from the same pass.  Synchronises the stream. */
int mgx_fused_up(const double *coarse_v_dev, double *v_dev, const double *f_dev, int m, double dx, int nsweeps,
                 double omega, double *norm_host, void *stream);

/* ---- j-slab operator: the unit a multi-GPU driver launches ------------------
 *
 * The reference is single-process; this entry point has no counterpart there.
 * It runs one fused leg built from the operators of main.cpp:171-281 on a SLAB:
 * an m-row column-major array holding `ncols` consecutive columns of a level,
 * i.e. the columns one GPU owns plus the halo columns copied from its
 * neighbours.  The leg is a pipeline over the stages of the iterate (stage 0 =
 * v_in, stage s = after s relax sweeps):
 *   nsweeps       number of damped-Jacobi sweeps (<= MGX_SLAB_MAX_SWEEPS)
 *   pro_stage     >= 0: P(coarse_v) (refine) is added to that stage; -1: none
 *   norm_stage    >= 0: max |residual| of that stage over the produced columns
 *                 is merged (atomic max) into *norm_slot, encoded as
 *                 mgx_norm_decode() reads it; -1: none
 *   restrict_out  1: coarsen(residual(stage nsweeps)) of the produced columns
 *                 with even global index is written to coarse_f
 *   zero_in       1: v_in is all zeros and is not read
 *   store         1: rows 1..m-2 of the produced columns of stage nsweeps are
 *                 written to v_out (never a Dirichlet column)
 * The legs that exist (others are rejected with MGX_ERR_ARG):
 *   down            pro -1, norm -1, restrict 1            (any zero_in/store that makes sense)
 *   sweeps          pro -1, norm -1, restrict 0, store 1, nsweeps >= 1
 *   (sweeps +) norm pro -1, norm = nsweeps, restrict 0
 *   up              pro 0, norm -1 or nsweeps, restrict 0, store 1
 *   up + down       pro 0, norm 0..nsweeps, restrict 1, store 1, zero_in 0, nsweeps >= 1
 *   recomputing up  pro 1..nsweeps, norm -1, restrict 0, store 1, zero_in 1
 * Only columns [jlo, jhi) are produced; the columns around them are read as
 * halo.  A slab that contains a physical boundary column (global j = 0 or
 * j = M-1) names its local index in jfix_lo / jfix_hi (else -1); Dirichlet
 * columns are never updated and end the halo requirement on that side.
 * Without a fixed column on a side, the caller must supply, beyond the produced
 * range, nsweeps (+2 with restrict_out; +1 with norm_stage == nsweeps) valid
 * columns of v_in and f, and the coarse columns they interpolate from; the call
 * is rejected otherwise.  gf0 / gc0 are the GLOBAL column indices of local
 * column 0 of the fine / coarse slab: global coarse column J pairs with global
 * fine column 2J.  v_in and v_out must not overlap.  Asynchronous on `stream`. */
#define MGX_SLAB_MAX_SWEEPS 4
typedef struct mgx_slab_op {
    int struct_size;
    int nsweeps, pro_stage, norm_stage, restrict_out, zero_in, store;
    const double *v_in;
    double *v_out;
    const double *f;
    const double *coarse_v;
    double *coarse_f;
    unsigned long long *norm_slot;
    int m, ld, ncols;       /* fine slab: rows, pitch (>= m), columns present   */
    int mc, ldc, nccols;    /* coarse slab (pro_stage >= 0 or restrict_out)     */
    int jlo, jhi;           /* fine local columns to produce                    */
    int jfix_lo, jfix_hi;   /* local index of Dirichlet columns, or -1          */
    int gf0, gc0;           /* global column index of local column 0            */
    double dx, omega;
    int jlo2, jhi2;         /* a second range of fine local columns to produce in the same launch,
                               jlo2 >= jhi (the columns next to both neighbours go out in one
                               launch ahead of the interior); jlo2 == jhi2: none.  struct_size may
                               also be offsetof(mgx_slab_op, jlo2): the struct without this pair. */
} mgx_slab_op;
int mgx_slab_launch(const mgx_slab_op *op, void *stream);

/* Decode a norm slot written through norm_stage (0 decodes to -infinity: nothing
 * merged yet).  Slots combine across GPUs by an integer MAX. */
double mgx_norm_decode(unsigned long long slot);

/* One V(Ndown,Nup) cycle of the plan, enqueued on its stream with no host
 * synchronisation and no residual norm (a multi-GPU driver runs the coarse
 * tail of the hierarchy through this).  zero_guess = 1: the fine iterate is
 * taken to be all zeros, as a coarse level is when the down sweep reaches it
 * (main.cpp:295-299,345); its Dirichlet boundary must then be zero too. */
int mgx_plan_vcycle_async(mgx_plan *plan, int Ndown, int Nup, double omega, int zero_guess);

/* ------------------------------------------------------------------------
 * Partitioned plan: the same V-cycle over N j-slabs driven from ONE process, one host
 * thread, one compute stream and one copy stream per slab.  The reference is a
 * single-device program (main.cpp); this has no counterpart there except the results it
 * must reproduce: Sol and ResNorms are bit-identical to mgx_plan_* for any N.
 *   devices[nslabs]   HIP ordinal of each slab's device, NULL = slab k on device k modulo
 *                     the number of visible devices.  An ordinal may repeat: several slabs
 *                     then share a GPU (how the partitioned schedule runs on fewer GPUs).
 * Levels above options.tail_m are split (slab g owns a contiguous range of columns, the
 * owner of fine column 2J owns coarse column J); the levels below are held and computed by
 * every slab.  Per cycle with Ndown + Nup <= 4: one neighbour exchange (peer copies of
 * 8 columns of the level-0 iterate + a few of the level-1 right-hand side, overlapped with
 * the interior of the level-0 launch) and one gather of the tail's right-hand side.
 * Only the fused schedule exists here; options.schedule = MGX_SCHEDULE_STEPWISE is
 * rejected.  A grid that cannot be split (fewer than a halo's worth of columns per slab
 * above tail_m) is rejected with MGX_ERR_ARG: use mgx_plan_*.  After a failed call the plan
 * must be destroyed.
 * ------------------------------------------------------------------------ */
typedef struct mgx_mplan mgx_mplan;
int mgx_mplan_create(mgx_mplan **plan, int m, double dx, int coarsest, int nslabs, const int *devices,
                     const mgx_options *opt);
int mgx_mplan_destroy(mgx_mplan *plan);
/* The decomposition mgx_mplan_create() would choose, computed on the host (no GPU needed):
 * *distributed_levels = nd, and bounds[l*(nslabs+1) + g] = first global column slab g owns of
 * level l, for l = 0..nd (level nd is the top of the tail) and g = 0..nslabs (entry nslabs is
 * the size of the level).  bounds may be NULL; otherwise bounds_len >= (nd+1)*(nslabs+1).
 * tail_m = 0 takes the default.  mgx_halo_plan: the halo widths of the lean schedule
 * (Ndown+Nup in 1..4) for nd distributed levels, Ev[0..nd] / Ef[0..nd]: columns beyond its
 * own that a slab computes of the level-l iterate / holds of the level-l right-hand side. */
int mgx_mplan_partition(int m, int coarsest, int nslabs, int tail_m, int *distributed_levels, int *bounds,
                        int bounds_len);
int mgx_halo_plan(int distributed_levels, int Ndown, int Nup, int *Ev, int *Ef);
/* Geometry of the grouped small-level launches (csrc/mgx_mid.hip), computed on the host (no
 * GPU needed; no counterpart in the reference: it has one loop nest per operator and level).
 * A launch takes nl consecutive levels, the finest with m0 rows; a workgroup owns a tile of
 * the output level (down: level nl, up: level 0) and recomputes what the tile depends on.
 * mgx_mid_shape: the tile, tiles per axis, threads and LDS bytes a launch would use (ok = 0:
 * no such launch; the plan then runs the levels one by one).  mgx_mid_spans: along one axis,
 * for tile t of `tile` points, inclusive index ranges lo[]/hi[] --
 *   down: [k], k = 0..nl: points of level k whose right-hand side the workgroup holds;
 *         [nl+1+k]: points of level k it writes (these partition the level);
 *   up:   [k], k = 0..nl-1: points of level k it computes; [nl]: points of level nl it reads
 *         (boundary included); [nl+1]: points of level 0 it writes.
 * lo, hi: at least 2*(nl+1) entries each. */
typedef struct mgx_mid_info {
    int ok, threads, points_per_thread, tile_i, tile_j, ntiles_i, ntiles_j;
    size_t lds_bytes;
} mgx_mid_info;
int mgx_mid_shape(int down, int nl, int m0, int Ndown, int Nup, mgx_mid_info *out);
int mgx_mid_spans(int down, int nl, int m0, int Ndown, int Nup, int tile, int t, int *lo, int *hi);
int mgx_mplan_nslabs(const mgx_mplan *plan);
int mgx_mplan_depth(const mgx_mplan *plan);
int mgx_mplan_distributed_levels(const mgx_mplan *plan); /* levels split over the slabs          */
int mgx_mplan_tail_m(const mgx_mplan *plan);             /* size of the first level held whole   */
size_t mgx_mplan_bytes(const mgx_mplan *plan, int slab); /* device memory of one slab            */
/* Host <-> slabs: f, v are whole m*m column-major host arrays; each slab copies its own
 * columns (a contiguous block).  Synchronous. */
int mgx_mplan_upload(mgx_mplan *plan, const double *f_host, const double *v_host);
int mgx_mplan_download(mgx_mplan *plan, double *v_host);
/* Device access to one slab's fine-level arrays, for callers that produce f on the GPU:
 * the slab stores global columns [col_first, col_first + ncols) (its own columns
 * [own_first, own_first + own_count) plus halo) as an m-row column-major block.  Fill f_dev
 * and v_dev for ALL stored columns (halo included, or at least the own columns: the halo of
 * v is refreshed at the start of mgx_mplan_cycles, the halo of f must be right), on `device`,
 * then call mgx_mplan_commit().  v_dev names the live iterate; ask again after cycles. */
typedef struct mgx_slab_info {
    int struct_size;
    int device, m, col_first, ncols, own_first, own_count;
    double *f_dev, *v_dev;
    void *stream; /* the slab's compute stream */
} mgx_slab_info;
int mgx_mplan_slab(mgx_mplan *plan, int slab, mgx_slab_info *info);
int mgx_mplan_commit(mgx_mplan *plan);
/* Nv cycles: the loop of main.cpp:309-354 over all slabs.  resnorms/times as in
 * mgx_plan_cycles (times: whole-cycle device time of slab 0).  Synchronous. */
int mgx_mplan_cycles(mgx_mplan *plan, int Nv, int Ndown, int Nup, double omega, int pure_jacobi, double *resnorms,
                     double *times);
/* EXTENSION: full multigrid over the slabs, the composition mgx_plan_fmg() runs on one
 * device (same arguments, same bits): right-hand side restricted to every level, Dirichlet
 * values injected, every slab's tail runs its own mgx_plan_fmg, each distributed level then
 * starts from refine() of the level below and runs ncyc V(Ndown,Nup) cycles on the hierarchy
 * below it.  Only the boundary values of v are read.  resnorms/times: the finest level's
 * history (ncyc+1 entries each, or NULL). */
int mgx_mplan_fmg(mgx_mplan *plan, int ncyc, int Ndown, int Nup, double omega, double *resnorms, double *times);
/* Measure the launch shapes of every slab's big launches now (see mgx_plan_tune). */
int mgx_mplan_tune(mgx_mplan *plan, int Ndown, int Nup, double omega);
/* Per-launch timing of slab 0's level-0 launches (kernel ids as mgx_plan_profile; the
 * edge-column launches that precede an exchange are not counted); points = grid points the
 * timed launches produced. */
int mgx_mplan_profile(mgx_mplan *plan, int enable);
int mgx_mplan_profile_get(const mgx_mplan *plan, int kernel_id, long *count, double *total_ms, double *points);
/* Since mgx_mplan_profile(): neighbour exchanges and tail gathers begun; of the last
 * mgx_mplan_cycles() call: its wall seconds on the host, and how much of that slab 0's thread
 * needed to enqueue its work (hand-shakes with the neighbours' threads included) before it
 * started waiting for its device -- the host cost of the driver. */
int mgx_mplan_stats(const mgx_mplan *plan, long *exchanges, long *gathers, double *host_seconds,
                    double *enqueue_seconds);

/* Launch-shape tuning.  A fused launch over at least 8 M grid points that streams both the
 * iterate and the right-hand side can run in two or three workgroup shapes that compute
 * the same bits at different speeds; which one is fastest is measured once per geometry and
 * process.  No counterpart in the reference.
 *   mgx_plan_tune()   measures now, for every level-0 launch a run of V(Ndown,Nup) cycles on
 *                     this plan issues: three cycles with each candidate shape run on the spot
 *                     (blocking), then the fine iterate is put back as it was.  Call it after
 *                     the data is set and before the cycles that matter; later calls of
 *                     mgx_plan_cycles() never block for tuning.
 *   mgx_tune_mode(1)  the same forced measurement for whatever the calling thread launches
 *                     (mgx_slab_launch, mgx_plan_vcycle_async) until mgx_tune_mode(0); returns
 *                     the previous setting.  A launch repeats safely: it never writes what it
 *                     reads and merges norms by a max.
 * Without either, the first ten launches of such a geometry time the candidates in place
 * (three launches each between events on the launch stream; the third of each group waits
 * for the stream).  Environment: MGX_TUNE=0 turns measuring off (a fixed rule picks the
 * shape), MGX_TUNE_VERBOSE=1 reports each decision on stderr. */
int mgx_plan_tune(mgx_plan *plan, int Ndown, int Nup, double omega);
int mgx_tune_mode(int enable);
/* Forget every measured shape of this process: the next launches measure again (tests use
 * it to send every candidate shape through a parity check).  Not to be called while another
 * thread launches. */
int mgx_tune_reset(void);
/* The shapes measured so far in this process as text, one line per geometry (kernel instance,
 * device, rows, columns produced, chosen shape); copies at most len-1 characters and a NUL into
 * buf (which may be NULL) and returns the size the whole text needs, NUL included. */
size_t mgx_tune_report(char *buf, size_t len);

#ifdef __cplusplus
}
#endif
#endif /* MGX_H */
