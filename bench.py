#!/usr/bin/env python3
"""Benchmark of the multigrid V-cycle hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--size M]

One "step" is one V(2,2) cycle (omega 0.6, coarsest 2) of the 2D Poisson solver
on an M x M fp64 grid, synthetic random right-hand side generated on the GPU,
zero initial guess, everything resident in HBM before the timed region.  The
metric is fine-grid DOF/s per V-cycle = M^2 * K / seconds.  For N > 1 the same
M x M problem is split into j-slabs (strong scaling), one rank per GPU: started
as `python bench.py --gpus N` the script starts its own N ranks as child
processes (before it imports torch or touches a GPU); under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it takes
the ranks it is given.

Rank 0 prints ONE JSON line.  `roofline` describes the kernel with the largest
share of the timed region (HIP events around every fine-level launch, on the
stream the kernels run on); `cpu_baseline` is the CPU oracle (a plain-C port of
the reference algorithm, 1 thread) timed on this box.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak

# Algorithmic (compulsory) HBM bytes per fine-level point and launch:
# fp64 streams with perfect neighbour reuse.
#   Jacobi sweep      : read u, f; write u'                     = 24
#   residual          : read u, f; write r                      = 24
#   restriction       : read r; write 1/4 coarse point          = 10
#   prolong + correct : read 1/4 coarse point; read + write u   = 18
#   residual norm     : read u, f                               = 16
#   fused down  (nu1 sweeps + residual + restriction)           = 24*nu1 + 18
#   fused up    (prolong + correct + nu2 sweeps)                = 18 + 24*nu2
def algorithmic_bytes_per_point(kernel, nd, nu):
    return {
        "k_jacobi": 24.0,
        "k_residual": 24.0,
        "k_restrict": 10.0,
        "k_refine<1>": 18.0,
        "k_residual_norm": 16.0,
        "k_stream:down": 24.0 * nd + 18.0,
        "k_stream:up": 18.0 + 24.0 * nu + 16.0,  # the level-0 up launch also yields the residual norm
        "k_stream:norm": 16.0,
        # level 0: the up leg of one cycle and the down leg of the next in one launch
        "k_stream:updown": (18.0 + 24.0 * nu + 16.0) + (24.0 * nd + 18.0),
    }[kernel]


# What one fused launch must move at the very least (every array touched once): the
# figure its real efficiency is judged by.  The table above prices the same work as the
# separate passes of the reference, so a fused launch exceeds "100 %" of it.
def fused_bytes_per_point(kernel, nd, nu):
    return {
        "k_stream:down": 8.0 + 8.0 + (8.0 if nd > 0 else 0.0) + 2.0,  # read u, f; write u'; write 1/4 coarse f
        "k_stream:up": 2.0 + 8.0 + 8.0 + 8.0,                         # read 1/4 coarse v, u, f; write u'
        "k_stream:norm": 16.0,
        "k_stream:updown": 2.0 + 8.0 + 8.0 + 8.0 + 2.0,
    }.get(kernel)


# HBM bytes per launch from the PMC counters (rocprofv3 --pmc, separate passes, gfx950
# corrections applied) for the default workload, measured in a SEPARATE profiled run of
# this same command and kept with the round's profiles.  Reported as `traffic_profiled`
# (never as `traffic`, which is reserved for something this run measured itself).  The figure
# is read from the profile summary itself, so the two cannot drift apart.
PMC_TRAFFIC_FILES = ["profiles/r03_final_pmc_traffic.txt"]


def profiled_traffic(kernel_rocprof_name):
    """(bytes per launch, file) of the dominant launch from the newest committed PMC summary
    (profiles/pmc_summary.py output of `bench.py` with its default workload): the line of this
    kernel that is marked as the launch shape the timed region used; (None, None) if there is none."""
    import re
    short = kernel_rocprof_name.replace("mgx::(anonymous namespace)::", "")
    for rel in PMC_TRAFFIC_FILES:
        path = os.path.join(ROOT, rel)
        if not os.path.exists(path):
            continue
        for ln in open(path):
            m = re.search(r"corrected=([0-9.]+) GB", ln)
            if m and ln.startswith(short) and "timed-region shape" in ln:
                return float(m.group(1)) * 1e9, rel
    return None, None


def rocprof_kernel_name(kernel, nd, nu):
    """The name the fine-level launch carries in a rocprofv3 kernel trace
    (template arguments NS, PA, NM, RES, ZERO_IN, STORE of k_stream)."""
    ns = "mgx::(anonymous namespace)::k_stream<%d, %d, %d, %d, false, %s>"
    if kernel == "k_stream:down":
        return ns % (nd, -1, -1, 1, "true" if nd > 0 else "false")
    if kernel == "k_stream:up":
        return ns % (nu, 0, nu, 0, "true")
    if kernel == "k_stream:updown":
        return ns % (nu + nd, 0, nu, 1, "true")
    if kernel == "k_stream:norm":
        return ns % (0, -1, 0, 0, "false")
    return "mgx::" + kernel


def cycle_bytes(M, depth, nd, nu):
    """B = sum_{l<depth} n_l*(24(nd+nu)+36) + 16*n_0, the per-cycle algorithmic traffic."""
    tot = 0.0
    m = M
    for _ in range(depth):
        tot += float(m) * m * (24.0 * (nd + nu) + 36.0)
        m = (m - 1) // 2 + 1
    return tot + 16.0 * M * M


def cycle_compulsory_bytes(M, depth, nd, nu):
    """What the lean schedule must move per cycle at the least: 28 B per point of every level
    with legs of its own (level 0: read v, f, 1/4 coarse v; write v, 1/4 coarse f.  Levels
    >= 1: down leg reads f, writes 1/4 coarse f; up leg reads f, 1/4 coarse v, writes v)."""
    tot = 0.0
    m = M
    for _ in range(depth):
        tot += 28.0 * float(m) * m
        m = (m - 1) // 2 + 1
    return tot


def cpu_baseline(M, nd, nu, omega, omp=False, coarsest=2, exact=False):
    """Time the CPU oracle on V-cycles of the same workload: the plain restatement on one
    thread (the reference is single-threaded), or its OpenMP build on the host cores this
    process may use."""
    import numpy as np
    from oracle import oracle as O
    from multigrid_amd import rhs
    L = O.lib_omp() if omp else O.lib()
    cores = L.mgo_threads()
    f = rhs.random_np(M)
    v = np.zeros((M, M), order="F")
    ncyc = 4 if omp else 1
    if omp:
        O.multigrid(f, v, 1.0 / (M - 1), 1, nd, nu, omega, coarsest, omp=True, coarse_exact=exact)  # first touch
    t0 = time.perf_counter()
    O.multigrid(f, v, 1.0 / (M - 1), ncyc, nd, nu, omega, coarsest, omp=omp, coarse_exact=exact)
    dt = time.perf_counter() - t0
    return {"value": M * M * ncyc / dt, "unit": "DOF/s", "cores": cores, "kind": "port",
            "sample": "%d V(%d,%d) cycle(s) at %dx%d incl. the residual-norm passes, %.1f s"
                      % (ncyc, nd, nu, M, M, dt)}


def cpu_baseline_reference(M, nd, nu, omega, coarsest=2):
    """The reference's OWN MultiGrid() (main.cpp:283-357 compiled from the reference checkout
    where that checkout exists; the built libraries travel with the tree) timed on one core,
    one V-cycle at M x M, in two builds: oracle/_ref/librefops_release.so with the flags the
    reference's Release configuration ships (-O3 -Ofast -DNDEBUG, CMakeLists.txt:17) -- the stated
    baseline -- and oracle/_ref/librefops.so (-O2, no fast-math), the build the parity tests pin
    the oracle with.  None if neither library was built."""
    import numpy as np
    from oracle import oracle as O
    from multigrid_amd import rhs
    # the reference allocates ~7 arrays of the fine size on top of f and v: stay within the host
    avail = os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
    while M > 1025 and 10.0 * 8.0 * M * M > 0.5 * avail:
        M = (M - 1) // 2 + 1
    out = None
    for release in (True, False):
        if (O.ref_ops_release() if release else O.ref_ops()) is None:
            continue
        R = O.Ref(release=release)
        f = rhs.random_np(M)
        v = np.zeros((M, M), order="F")
        t0 = time.perf_counter()
        _, _, tim = R.multigrid(f, v, 1.0 / (M - 1), 1, nd, nu, omega, coarsest)
        dt = time.perf_counter() - t0
        rec = {"value": M * M / dt, "unit": "DOF/s", "cores": 1, "kind": "reference",
               "build": "-O3 -Ofast -DNDEBUG (the reference's Release flags)" if release else "-O2, no fast-math (the parity build)",
               "sample": "1 V(%d,%d) cycle at %dx%d by the reference's MultiGrid() itself, %.1f s of which %.1f s inside "
                         "relax() by its own Times" % (nd, nu, M, M, dt, float(tim[-1]))}
        if out is None:
            out = rec
        else:
            out["parity_build"] = rec
    return out


def comm_counters(solver):
    """(neighbour exchanges, tail gathers) a solver has begun so far."""
    if hasattr(solver, "comm"):
        return solver.comm.nexchanges, solver.comm.ngathers
    if hasattr(solver, "stats"):
        st = solver.stats()
        return st[0], st[1]
    return 0, 0


def launch_ranks(n, out_fd):
    """Start `n` ranks of this script as child processes (one per GPU, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment, rendezvous on 127.0.0.1), pass rank 0's JSON
    line on to `out_fd` and return the worst exit status.  The caller has initialised
    neither torch nor the GPU, and nothing is exec'ed over a process that has."""
    import signal
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    limit = float(os.environ.get("MGX_BENCH_TIMEOUT", "1500"))
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"WORLD_SIZE": str(n), "RANK": str(r), "LOCAL_RANK": str(r), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "MGX_BENCH_LAUNCHER": "self"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else None))
    # rank 0's stdout is read while it runs (a line longer than the pipe's buffer must not block it)
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    t0 = time.time()
    worst, failed_at = 0, None
    live = set(range(n))
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0:
                worst = worst or (rc if rc > 0 else 128 - rc)
                failed_at = failed_at or time.time()
                sys.stderr.write("bench.py: rank %d ended with status %d\n" % (r, rc))
        overdue = time.time() - t0 > limit
        # a rank that failed leaves the others waiting in a collective: end them
        if live and (overdue or (failed_at and time.time() - failed_at > 10)):
            for r in live:
                procs[r].send_signal(signal.SIGTERM)
            time.sleep(5)
            for r in live:
                if procs[r].poll() is None:
                    procs[r].kill()
            worst = worst or 124
            if overdue:
                sys.stderr.write("bench.py: ranks still running after %.0f s were ended\n" % limit)
            break
        time.sleep(0.05)
    for p in procs:
        p.wait()
    reader.join(timeout=30)
    text = b"".join(chunks)
    lines = [l for l in text.decode(errors="replace").splitlines() if l.startswith("{")]
    if lines:
        os.write(out_fd, (lines[-1] + "\n").encode())
    elif worst == 0:
        worst = 1
        sys.stderr.write("bench.py: rank 0 printed no JSON line\n")
    return worst


def main():
    # Libraries write to stdout too (RCCL prints a version banner when a communicator comes
    # up): everything goes to stderr, except the one JSON line at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="V-cycles timed (the reference's default Nv is 100)")
    ap.add_argument("--warmup", type=int, default=8,
                    help="untimed cycles first (eight let the launch-shape measurement finish before the timed region)")
    ap.add_argument("--size", type=int, default=16385, help="grid points per side (2^k+1)")
    ap.add_argument("--nd", type=int, default=2)
    ap.add_argument("--nu", type=int, default=2)
    ap.add_argument("--omega", type=float, default=0.6)
    ap.add_argument("--schedule", type=int, default=0, help="0 default, 1 stepwise, 2 fused")
    ap.add_argument("--slab", action="store_true", help="run the j-slab driver even on one GPU")
    ap.add_argument("--mplan", type=int, default=0,
                    help="single process: the partitioned plan in C++ (mgx_mplan_*) over this many slabs, slab k on "
                         "device k modulo the visible devices (a rehearsal on one GPU: every slab's copy of the "
                         "tail then runs on that GPU)")
    ap.add_argument("--tail-m", type=int, default=0,
                    help="slab driver: largest replicated (tail) level size; 0 = 2049 on one rank, 1025 on more")
    ap.add_argument("--coarsest", type=int, default=2, help="the reference's -c (threshold for the direct solver)")
    ap.add_argument("--exact-coarse", action="store_true",
                    help="extension: solve a coarsest grid other than 3x3 exactly (needed for --coarsest != 2)")
    ap.add_argument("--weak", action="store_true",
                    help="weak scaling: ~16385^2 points per GPU (16385, 22529, 32769, 45057 per side on 1/2/4/8 "
                         "GPUs; the sizes 11*2^k+1 use --coarsest 11 --exact-coarse)")
    ap.add_argument("--cpu-baseline", choices=["auto", "off"], default="auto")
    ap.add_argument("--cpu-size", type=int, default=0, help="grid size for the CPU baseline (0 = --size)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and args.mplan == 0:
        # `python bench.py --gpus N` with no launcher around it: this process becomes the
        # launcher.  It touches neither torch nor the GPU; the ranks are its children.
        sys.exit(launch_ranks(args.gpus, real_stdout))

    # the HIP runtime reads this when it comes up (dmabuf IPC, what RCCL between processes needs here)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world
    # one GPU per rank; MGX_DIST_BACKEND=gloo rehearses the multi-rank path on a box
    # with fewer GPUs than ranks (ranks then share devices and stage halos on the host)
    backend = os.environ.get("MGX_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev)
    grouped = world > 1 or (os.environ.get("MGX_COMM_FORCE") == "1" and "MASTER_ADDR" in os.environ)
    if grouped:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)
        # bring the communicator up with a collective before the first point-to-point batch
        warm = torch.zeros(1, dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(warm)
        if backend == "nccl":
            torch.cuda.synchronize()

    import multigrid_amd as mg
    from multigrid_amd import rhs

    if args.weak:
        sizes = {1: (16385, 2), 2: (22529, 11), 4: (32769, 2), 8: (45057, 11)}
        if world not in sizes:
            sys.exit("bench.py: --weak is defined for 1, 2, 4 or 8 GPUs")
        args.size, args.coarsest = sizes[world]
        args.exact_coarse = args.exact_coarse or args.coarsest != 2
    M, K, W = args.size, args.steps, args.warmup
    dx = 1.0 / (M - 1)
    coarse_solve = mg.COARSE_EXACT if args.exact_coarse else mg.COARSE_REFERENCE

    if world == 1 and args.mplan > 0:
        solver = mg.MultiPlan(M, dx, args.coarsest, nslabs=args.mplan, tail_m=args.tail_m, coarse_solve=coarse_solve)
        solver.set_random_rhs()
        depth = solver.depth
        schedule_name = "one process, %d j-slabs on %d device(s), %d distributed levels, replicated tail from %d^2" % (
            args.mplan, min(args.mplan, ndev), solver.ndist, solver.tail_m)
    elif world == 1 and not args.slab:
        plan = mg.Plan(M, dx, args.coarsest, schedule=args.schedule, coarse_solve=coarse_solve)
        plan.set(rhs.random_torch(M), mg.colmajor(M, M))
        solver = plan
        depth = plan.depth
        schedule_name = "single-gpu"
    else:
        from multigrid_amd import slab
        solver = slab.SlabSolver(M, dx, args.coarsest, schedule=args.schedule, tail_m=args.tail_m,
                                 coarse_solve=coarse_solve)
        solver.set_random_rhs()
        depth = solver.depth
        schedule_name = "j-slabs x%d, %d distributed levels, replicated tail from %d^2" % (
            world, solver.ndist, solver.dims[solver.ndist])

    def barrier():
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    # launch shapes are measured before anything is timed (and before the warm-up, so that
    # the warm-up cycles already run the shapes the timed region uses)
    solver.tune(args.nd, args.nu, args.omega)
    if W > 0:
        solver.cycles(W, args.nd, args.nu, args.omega)
    solver.profile(True)
    comm_before = comm_counters(solver)
    barrier()
    t0 = time.perf_counter()
    out = solver.cycles(K, args.nd, args.nu, args.omega)
    barrier()
    elapsed = time.perf_counter() - t0
    if grouped:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    prof = solver.profile_totals()
    solver.profile(False)

    # what each rank ran: its level-0 launch times, the launch shapes its tuner chose, the
    # exchanges and gathers it began in the timed region
    from multigrid_amd import _lib as mglib
    comm_now = comm_counters(solver)
    info = {"rank": rank, "device": dev,
            "level0_avg_ms": {k: round(ms / c, 5) for k, (c, ms) in prof.items() if c},
            "exchanges_per_cycle": (comm_now[0] - comm_before[0]) / float(K),
            "gathers_per_cycle": (comm_now[1] - comm_before[1]) / float(K),
            "launch_shapes": mglib.tune_report()}
    infos = [info]
    if grouped:
        infos = [None] * world
        dist.all_gather_object(infos, info)

    # where the cycle's time goes, level by level (single plan): a few more cycles with every
    # launch bracketed by events, outside the timed region
    levels = None
    if world == 1 and not args.slab and args.mplan == 0 and args.schedule != 1 and hasattr(solver, "profile_levels"):
        nlev = 5
        solver.profile(2)
        solver.cycles(nlev, args.nd, args.nu, args.omega)
        rows = solver.profile_levels()
        solver.profile(False)
        levels = []
        for (d, m_d, n, ms) in rows:
            comp = 28.0 * float(m_d) * m_d if d < 4 else None
            us = ms * 1e3 / nlev
            rec = {"level": d, "m": m_d, "launches_per_cycle": n / float(nlev), "us_per_cycle": round(us, 2),
                   "compulsory_bytes": comp}
            if comp and d == 0:
                # every level-0 launch moves the level once (up+down 28 B/point; the call's opening
                # down and closing up launch 26): judged per launch, not per cycle of this short run
                rec["us_per_launch"] = round(ms * 1e3 / n, 2)
                rec["frac"] = round(comp / (ms * 1e-3 / n) / 1e9 / HBM_PEAK_GBS, 4)
            elif comp:
                rec["frac"] = round(comp / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)  # down + up leg together: 10 + 18 B/point
            else:
                rec["frac"] = None
            levels.append(rec)

    if rank == 0:
        n0 = float(M) * M
        value = n0 * K / elapsed
        B = cycle_bytes(M, depth, args.nd, args.nu)
        roof = None
        if prof:
            cands = {k: v for k, v in prof.items() if k != "k_stream:edge"} or prof
            kname, (cnt, tot_ms) = max(cands.items(), key=lambda kv: kv[1][1])
            avg_s = tot_ms * 1e-3 / cnt
            # on N GPUs each launch covers this rank's slab (less the edge columns that are
            # launched first so that the halo exchange overlaps the rest)
            pts = n0 / world
            if hasattr(solver, "profile_points") and solver.profile_points().get(kname):
                pts = solver.profile_points()[kname] / cnt
            # `achieved`/`frac`: the bytes this ONE launch must move at the least (every array
            # it touches once) over its measured duration -- a physical position under the
            # HBM roof, at most 1.  The same launch priced as the separate passes of the
            # reference's operators it fuses (24 B per sweep, 18 per transfer, 16 per norm)
            # is kept as algorithmic_equiv_*: that ratio exceeds 1 because fusion removes passes.
            bpp = algorithmic_bytes_per_point(kname, args.nd, args.nu)
            fbpp = fused_bytes_per_point(kname, args.nd, args.nu) or bpp
            achieved = fbpp * pts / avg_s / 1e9
            whole_grid = world == 1 and not args.slab and args.mplan == 0  # the profiled launch is the single-plan one
            prof_traffic, prof_src = (profiled_traffic(rocprof_kernel_name(kname, args.nd, args.nu))
                                      if whole_grid and (M, args.nd, args.nu) == (16385, 2, 2) else (None, None))
            roof = {"bound": "hbm", "kernel": rocprof_kernel_name(kname, args.nd, args.nu),
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": None,
                    "bytes_per_point": fbpp, "points_per_launch": pts, "launches": cnt, "avg_launch_ms": avg_s * 1e3,
                    "algorithmic_equiv_bytes_per_point": bpp,
                    "algorithmic_equiv_GBps": bpp * pts / avg_s / 1e9,
                    "algorithmic_equiv_ratio_to_peak": bpp * pts / avg_s / 1e9 / HBM_PEAK_GBS,
                    "traffic_profiled": {"bytes_per_launch": prof_traffic, "source": prof_src,
                                         "note": "PMC counters of a separate rocprofv3 run of this command"}
                    if prof_traffic else None,
                    "share_of_timed_region": tot_ms * 1e-3 / elapsed,
                    "all_fine_level_kernels_ms": {k: [c, round(ms, 4)] for k, (c, ms) in prof.items()}}
        line = {
            "metric": "fine-grid DOF/s per V-cycle (fp64)",
            "value": value, "unit": "DOF/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed * 1e3 / K, "higher_is_better": True,
            "scaling": "weak" if args.weak else "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "2D Poisson 5-point %dx%d fp64 V(%d,%d), omega %.2f, coarsest %d%s, random RHS, "
                                   "zero guess, %d levels" % (M, M, args.nd, args.nu, args.omega, args.coarsest,
                                                              " (exact coarsest solve)" if args.exact_coarse else "",
                                                              depth + 1),
                       "M": M, "decomposition": schedule_name, "schedule": args.schedule},
            "cycle_algorithmic_equiv_GB": B / 1e9,
            "cycle_algorithmic_equiv_GBps": B * K / elapsed / 1e9,
            "cycle_compulsory_GB": cycle_compulsory_bytes(M, depth, args.nd, args.nu) / 1e9,
            "cycle_frac_of_hbm_peak": cycle_compulsory_bytes(M, depth, args.nd, args.nu) * K / elapsed / 1e9
                                      / (HBM_PEAK_GBS * world),
            "resnorm_first_last": [float(out.ResidualNorms[0]), float(out.ResidualNorms[-1])],
            "roofline": roof,
        }
        if levels:
            line["levels"] = levels
            line["levels_note"] = ("a separate run of 5 cycles with every launch between events (adds a few us per "
                                   "small launch); a launch over several small levels is booked on its finest level; "
                                   "level 0: 6 launches in these 5 cycles (the call's opening down and closing up launch), frac per launch")
        if grouped or args.slab or args.mplan:
            main = [i["level0_avg_ms"].get(kname) for i in infos if i and i["level0_avg_ms"].get(kname)] if roof else []
            line["comm"] = {
                "backend": dist.get_backend() if grouped else ("peer copies in one process" if args.mplan else "none"),
                "rccl_world": dist.get_world_size() if grouped else 1,
                "launcher": os.environ.get("MGX_BENCH_LAUNCHER", "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ
                                           else "environment"),
                "exchanges_per_cycle": infos[0]["exchanges_per_cycle"], "gathers_per_cycle": infos[0]["gathers_per_cycle"],
                "level0_launch_ms_min_max_over_ranks": [min(main), max(main)] if main else None,
                "per_rank": infos}
        if world == 1 and args.cpu_baseline == "auto":
            try:
                line["cpu_baseline"] = cpu_baseline(args.cpu_size or M, args.nd, args.nu, args.omega, False,
                                                    args.coarsest, args.exact_coarse)
                line["cpu_baseline_all_cores"] = cpu_baseline(args.cpu_size or M, args.nd, args.nu, args.omega, True,
                                                              args.coarsest, args.exact_coarse)
                if not args.exact_coarse:
                    ref = cpu_baseline_reference(args.cpu_size or M, args.nd, args.nu, args.omega, args.coarsest)
                    if ref:
                        line["cpu_baseline_reference"] = ref
            except Exception as e:  # the oracle is a checker; its absence must not hide the GPU number
                line["cpu_baseline"] = {"value": None, "unit": "DOF/s", "cores": 1, "kind": "port",
                                        "sample": "failed: %r" % (e,)}
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())

    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
