"""Multi-GPU V-cycle: j-slab decomposition, one process per GPU.

The reference is single-process (main.cpp); this module is the multi-device
driver of the same V-cycle (main.cpp:283-357) and produces the same bits.

Decomposition.  Every level is column-major, so a contiguous range of columns j
is one contiguous block of memory.  The finest `ndist` levels are DISTRIBUTED:
rank g owns the columns [b_g, b_{g+1}) of a level and stores them together with
halo columns copied from (or recomputed for) each neighbour.  Ownership is
nested, b_g(fine) = 2*b_g(coarse) - 1, so the rank that owns fine column 2J
owns coarse column J and both level transfers are local apart from the halo.
Levels from the first one with at most `tail_m` points per side downwards form
the TAIL: its right-hand side is all-gathered and every rank runs the tail
redundantly on its own GPU (identical bits everywhere), which removes the
scatter on the way up.

Communication per cycle with the usual few sweeps (Ndown + Nup <= 4, "lean"):
ONE neighbour exchange -- the level-0 iterate (8 columns per side) and the
level-1 right-hand side (a wider halo) in one batch -- and ONE all-gather at
the tail.  The coarser distributed levels need no exchange at all: a coarse
level's data is a deterministic function of the level-1 right-hand side, so
each rank recomputes the few halo columns it needs (the halo widths below);
Jacobi makes the recomputed values bit-identical to the neighbour's.  With
more sweeps every leg is a chain of launches with an exchange after each.
Exchanges are point-to-point send/recv of whole columns through
torch.distributed: RCCL over xGMI with backend "nccl", or gloo for tests.

The arithmetic runs through a backend object.  The product backend is
HipBackend (libmgx's mgx_slab_launch / mgx_plan_vcycle_async); tests inject a
CPU backend to exercise this driver under gloo without a GPU.
"""
import ctypes
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .solver import MGOutputs


class SlabOp:
    """Arguments of one fused slab launch (mirrors mgx_slab_op in include/mgx.h)."""
    __slots__ = ("nsweeps", "pro_stage", "norm_stage", "restrict_out", "zero_in", "store", "v_in", "v_out", "f",
                 "coarse_v", "coarse_f", "norm_slot", "m", "ncols", "mc", "nccols", "jlo", "jhi", "jfix_lo", "jfix_hi",
                 "gf0", "gc0", "dx", "omega", "jlo2", "jhi2")

    def __init__(self, **kw):
        for k in self.__slots__:
            setattr(self, k, kw.get(k))


class HipBackend:
    """Slab arithmetic on the GPU through libmgx.  Arrays are torch float64 CUDA
    tensors of shape (ncols, m): row k of the tensor is column k of the slab."""
    name = "hip"

    def __init__(self, schedule=_lib.SCHEDULE_DEFAULT, coarse_solve=_lib.COARSE_REFERENCE):
        if not torch.cuda.is_available():
            raise RuntimeError("multigrid_amd.slab: HipBackend needs a GPU; there is no CPU path")
        self.L = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.schedule = schedule
        self.coarse_solve = coarse_solve  # what the replicated tail does on the coarsest grid

    def zeros(self, ncols, m, phase=0):
        """phase 1 / 2: a big array starts 384 / 768 KB into its allocation, so that the arrays a
        launch streams side by side sit at different phases of the memory's address interleave
        (allocations are aligned alike; what mgx_plan_create does for its levels)."""
        shift = (phase * 384 * 1024) // 8 if ncols * m * 8 >= (32 << 20) else 0
        if not shift:
            return torch.zeros((ncols, m), dtype=torch.float64, device=self.device)
        return torch.zeros(ncols * m + shift, dtype=torch.float64, device=self.device)[shift:].view(ncols, m)

    def slots(self, n):
        return torch.zeros(n, dtype=torch.int64, device=self.device)

    def from_numpy(self, a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def to_numpy(self, t):
        return t.cpu().numpy()

    def comm_tensor(self, t):
        return t

    def index_array(self, rows):
        return torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(self.device)

    def take_rows(self, src, rows, out):
        torch.index_select(src, 0, rows, out=out)

    def launch(self, op):
        o = _lib.mgx_slab_op()
        o.struct_size = ctypes.sizeof(_lib.mgx_slab_op)
        o.nsweeps, o.pro_stage, o.norm_stage = int(op.nsweeps), int(op.pro_stage), int(op.norm_stage)
        o.restrict_out, o.zero_in, o.store = int(op.restrict_out), int(op.zero_in), int(op.store)
        o.v_in = op.v_in.data_ptr() if op.v_in is not None else None
        o.v_out = op.v_out.data_ptr() if op.v_out is not None else None
        o.f = op.f.data_ptr()
        o.coarse_v = op.coarse_v.data_ptr() if op.coarse_v is not None else None
        o.coarse_f = op.coarse_f.data_ptr() if op.coarse_f is not None else None
        o.norm_slot = op.norm_slot
        o.m, o.ld, o.ncols = op.m, op.m, op.ncols
        o.mc, o.ldc, o.nccols = op.mc or 0, op.mc or 0, op.nccols or 0
        o.jlo, o.jhi, o.jfix_lo, o.jfix_hi = op.jlo, op.jhi, op.jfix_lo, op.jfix_hi
        o.jlo2, o.jhi2 = op.jlo2 or 0, op.jhi2 or 0
        o.gf0, o.gc0 = op.gf0, op.gc0 or 0
        o.dx, o.omega = op.dx, op.omega
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(self.L.mgx_slab_launch(ctypes.byref(o), stream))

    def slot_ptr(self, slots, k):
        return slots.data_ptr() + 8 * k

    def decode_slots(self, slots):
        keys = slots.cpu().numpy().astype(np.uint64)
        return np.array([self.L.mgx_norm_decode(int(k)) for k in keys])

    def make_tail(self, m, dx, coarsest):
        return _HipTail(self, m, dx, coarsest)

    def synchronize(self):
        torch.cuda.current_stream().synchronize()


class _DevicePointer:
    """Adapter that lets torch wrap device memory owned by libmgx (a plan's level-0 arrays)
    as a tensor without copying."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def _device_view(ptr, m, device):
    if not ptr:
        raise RuntimeError("libmgx returned a null device pointer")
    return torch.as_tensor(_DevicePointer(ptr, (m, m)), device=device)


class _HipTail:
    """The replicated coarse tail: a single-GPU plan of the tail's finest level."""

    def __init__(self, be, m, dx, coarsest):
        self.be = be
        self.m = m
        self.h = ctypes.c_void_p()
        opt = _lib.make_options(device=torch.cuda.current_device(), stream=torch.cuda.current_stream().cuda_stream,
                                schedule=be.schedule, coarse_solve=be.coarse_solve)
        _lib.check(be.L.mgx_plan_create(ctypes.byref(self.h), m, dx, coarsest, ctypes.byref(opt)))
        # the plan's own fine-level buffers as tensors (no copies in or out):
        # row k of the tensor is column k of the level
        self.f = _device_view(be.L.mgx_plan_f_dev(self.h), m, be.device)
        self.v = _device_view(be.L.mgx_plan_v_dev(self.h), m, be.device)

    def _on_current_stream(self):
        # the all-gather and the index copy that fill self.f, and the up leg that reads
        # self.v, run on torch's current stream: so must the plan
        _lib.check(self.be.L.mgx_plan_set_stream(self.h, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def run(self, Ndown, Nup, omega):
        """One V(Ndown,Nup) cycle from a zero guess for the right-hand side in self.f;
        returns the iterate (self.v).  Asynchronous on the current stream."""
        self._on_current_stream()
        _lib.check(self.be.L.mgx_plan_vcycle_async(self.h, Ndown, Nup, omega, 1))
        return self.v

    def fmg(self, ncyc, Ndown, Nup, omega):
        """Full multigrid on the tail (mgx_plan_fmg) for the right-hand side in self.f and the
        boundary values in self.v; returns the result (the plan's iterate buffer may have
        traded places with its twin, so the view is taken again)."""
        L = self.be.L
        self._on_current_stream()
        _lib.check(L.mgx_plan_commit(self.h))
        _lib.check(L.mgx_plan_fmg(self.h, ncyc, Ndown, Nup, omega, None, None))
        self.v = _device_view(L.mgx_plan_v_dev(self.h), self.m, self.be.device)
        return self.v

    def reset(self):
        """Zero iterate with zero boundary values: what a cycle's coarse-grid correction starts from."""
        self._on_current_stream()
        self.v.zero_()
        _lib.check(self.be.L.mgx_plan_commit(self.h))

    def close(self):
        if self.h:
            self.be.L.mgx_plan_destroy(self.h)
            self.h = ctypes.c_void_p()


class _Comm:
    """Neighbour exchange and collectives over torch.distributed (or a single rank)."""

    def __init__(self, group=None):
        self.group = group
        self.on = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.on else 0
        self.size = dist.get_world_size(group) if self.on else 1
        self.backend = dist.get_backend(group) if self.on else "none"
        # MGX_COMM_FORCE=1 issues the collectives even in a group of one rank (a way to
        # exercise the RCCL calls on a box with a single GPU)
        self.solo = self.size == 1 and not (self.on and os.environ.get("MGX_COMM_FORCE", "0") == "1")
        self.nexchanges = 0
        self.ngathers = 0

    def _global(self, r):
        return dist.get_global_rank(self.group, r) if self.group is not None else r

    def _stage(self, t):
        # gloo moves host memory: stage device tensors through the host
        return t.cpu() if (self.backend == "gloo" and t.is_cuda) else t

    def exchange(self, items):
        """items: list of (tensor, send_lo, recv_lo, send_hi, recv_hi) where each entry but
        the tensor is a (start, stop) row range or None.  *_lo talk to rank-1, *_hi to rank+1."""
        self.exchange_finish(self.exchange_start(items))

    def exchange_start(self, items):
        """Post the sends and receives of exchange(items); they run beside whatever is
        launched next on the compute stream.  Returns a handle for exchange_finish()."""
        self.nexchanges += 1
        if self.size == 1:
            return None
        ops, copies = [], []
        for idx, (t, s_lo, r_lo, s_hi, r_hi) in enumerate(items):
            for (rng, peer, is_send, side) in ((r_lo, self.rank - 1, False, 0), (r_hi, self.rank + 1, False, 1),
                                               (s_lo, self.rank - 1, True, 1), (s_hi, self.rank + 1, True, 0)):
                if rng is None or peer < 0 or peer >= self.size:
                    continue
                view = t[rng[0]:rng[1]]
                # a message is tagged by (item, which side of the RECEIVER it fills)
                tag = 2 * idx + side
                if is_send:
                    buf = self._stage(view)
                    ops.append((dist.isend, buf, peer, tag))
                else:
                    buf = self._stage(view) if not (self.backend == "gloo" and view.is_cuda) else torch.empty(
                        view.shape, dtype=view.dtype)
                    if buf is not view:
                        copies.append((view, buf))
                    ops.append((dist.irecv, buf, peer, tag))
        if not ops:
            return None
        if self.backend == "nccl":
            reqs = dist.batch_isend_irecv([dist.P2POp(fn, buf, self._global(peer), self.group) for (fn, buf, peer, _)
                                           in ops])
        else:
            reqs = [fn(buf, self._global(peer), group=self.group, tag=tag) for (fn, buf, peer, tag) in ops]
        return reqs, copies

    def exchange_finish(self, handle):
        """Make the compute stream (and, for gloo, the host) wait for a posted exchange."""
        if handle is None:
            return
        reqs, copies = handle
        for r in reqs:
            r.wait()
        for view, buf in copies:
            view.copy_(buf)

    def allgather(self, t, out=None):
        """Concatenation over ranks of equally shaped tensors along dim 0 (into `out` if given)."""
        self.ngathers += 1
        if self.solo:
            if out is not None:
                out.copy_(t)
                return out
            return t
        src = self._stage(t).contiguous()
        staged = out is None or out.device != src.device
        buf = torch.empty((self.size * src.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype,
                          device=src.device) if staged else out
        if self.backend == "nccl":
            dist.all_gather_into_tensor(buf, src, group=self.group)
        else:
            dist.all_gather(list(buf.chunk(self.size, 0)), src, group=self.group)
        if out is None:
            return buf.to(t.device) if buf.device != t.device else buf
        if staged:
            out.copy_(buf)
        return out

    def allreduce_max_(self, t):
        if self.solo:
            return t
        buf = self._stage(t)
        dist.all_reduce(buf, op=dist.ReduceOp.MAX, group=self.group)
        if buf is not t:
            t.copy_(buf)
        return t


class _Level:
    pass


MAX_LEAN = _lib.SLAB_MAX_SWEEPS  # Ndown + Nup up to this run the lean legs


def halo_plan(nd, Ndown, Nup):
    """Halo widths of the lean schedule for `nd` distributed levels.

    Returns (Ev, Ef): Ev[l] = columns beyond its own that a rank produces of the level-l
    iterate on the way up (l >= 1), Ef[l] = columns beyond its own of the level-l right-hand
    side a rank holds (received for l = 1, recomputed for l >= 2).  Derivation, with
    NS = Ndown + Nup: the level-0 launch reads the level-1 iterate within (NS+2)/2+1 coarse
    columns of its own; an up leg that produces e extra columns needs f within e + NS and
    the coarser iterate within (e + Nup)/2 + 1; a down leg that produces the coarser f on e
    extra columns needs f within 2e + Ndown + 2."""
    NS = Ndown + Nup
    Ev = [0] * (nd + 1)
    Ef = [0] * (nd + 1)
    if nd >= 2:
        Ev[1] = (NS + 2) // 2 + 1
        for l in range(2, nd):
            Ev[l] = (Ev[l - 1] + Nup) // 2 + 1
        for l in range(nd - 1, 0, -1):
            Ef[l] = max(Ev[l] + NS, 2 * Ef[l + 1] + Ndown + 2 if l + 1 < nd else 0)
    return Ev, Ef


def halo_capacity(nd):
    """Elementwise maximum of halo_plan over every lean (Ndown, Nup): what the arrays must hold."""
    Ev = [0] * (nd + 1)
    Ef = [0] * (nd + 1)
    for a in range(MAX_LEAN + 1):
        for b in range(MAX_LEAN + 1 - a):
            if a + b == 0:
                continue
            ev, ef = halo_plan(nd, a, b)
            Ev = [max(x, y) for x, y in zip(Ev, ev)]
            Ef = [max(x, y) for x, y in zip(Ef, ef)]
    return Ev, Ef


def view_capacity(nd):
    """halo_capacity for the hierarchy of nd distributed levels AND for every sub-hierarchy
    that starts at one of its levels (full multigrid runs cycles on those too)."""
    Ev = [0] * (nd + 1)
    Ef = [0] * (nd + 1)
    for s in range(nd):
        ev, ef = halo_capacity(nd - s)
        for j in range(nd - s + 1):
            Ev[s + j] = max(Ev[s + j], ev[j])
            Ef[s + j] = max(Ef[s + j], ef[j])
    return Ev, Ef


class SlabSolver:
    """Nv V(Ndown,Nup) cycles of the reference's MultiGrid() on an M x M grid split into j-slabs.

    Call from every rank of the process group with identical arguments.  With a
    single rank (no process group) it degenerates to one slab plus the tail."""

    W0 = 8  # halo columns of the level-0 arrays (the level-0 launch reads Ndown+Nup+2 <= 6)

    def __init__(self, M, dx, coarsest=2, schedule=_lib.SCHEDULE_DEFAULT, backend=None, group=None, tail_m=None,
                 halo=8, coarse_solve=_lib.COARSE_REFERENCE):
        self.be = backend if backend is not None else HipBackend(schedule, coarse_solve)
        self.comm = _Comm(group)
        if not tail_m:
            # Every rank repeats the tail, so its cost does not shrink with the number of
            # ranks while a distributed level's does: the more ranks, the smaller the tail.
            tail_m = int(os.environ.get("MGX_TAIL_M", "0")) or (2049 if self.comm.size == 1 else 1025)
        self.M, self.dx, self.coarsest = M, dx, coarsest
        if M < 3 or M % 2 == 0:
            raise ValueError("Error: Input has even rows and columns!")  # main.cpp:415
        L = _lib.load()
        self.depth = L.mgx_depth(M, coarsest)  # main.cpp:286
        if self.depth < 0:
            raise ValueError("depth %d < 0" % self.depth)
        dims = [M]
        hs = [dx]
        for _ in range(self.depth):
            dims.append((dims[-1] - 1) // 2 + 1)  # main.cpp:293
            hs.append(hs[-1] * 2.0)               # main.cpp:297
            if dims[-1] < 3 or (dims[-1] % 2 == 0 and len(dims) - 1 < self.depth):
                raise ValueError("level of size %d: the reference indexes out of range there" % dims[-1])
        self.dims, self.hs = dims, hs
        R = self.comm.size
        self.W0 = max(halo, SlabSolver.W0)
        # levels 0..nd-1 are distributed, level nd is the top of the tail
        # (the tail is a plan of its own and a plan's finest level is odd: an even coarsest
        # level cannot be the whole tail)
        max_nd = self.depth - (1 if dims[-1] % 2 == 0 else 0)
        nd = 0
        while nd < max_nd and dims[nd] > tail_m:
            # would every rank still own at least a halo's worth of columns of every
            # level if one more level were distributed?
            Ev, Ef = view_capacity(nd + 1)
            base = (dims[nd + 1] - 2) // R  # least number of columns a rank owns of the would-be tail top
            fits = base >= 1
            for l in range(nd + 1):
                own = base * (1 << (nd + 1 - l)) - 1
                need = self.W0 if l == 0 else max(self.W0, Ef[l], Ev[l])
                fits = fits and own >= need
            if R > 1 and not fits:
                break
            nd += 1
        if nd == 0:
            raise ValueError("grid %d is not split over %d rank(s) with tail_m=%d: solve it with "
                             "multigrid_amd.MultiGrid on one GPU" % (M, R, tail_m))
        self.ndist = nd
        Ev_max, Ef_max = view_capacity(nd)
        mt = dims[nd]
        b = [0] + [1 + (g * (mt - 2)) // R for g in range(1, R)] + [mt]  # ownership on the tail's top level
        bounds = {nd: b}
        for l in range(nd - 1, -1, -1):
            bc = bounds[l + 1]
            bounds[l] = [0] + [2 * bc[g] - 1 for g in range(1, R)] + [dims[l]]
        self.bounds = bounds
        g = self.comm.rank
        self.levels = []
        for l in range(nd):
            Lv = _Level()
            Lv.index = l
            Lv.m, Lv.dx = dims[l], hs[l]
            Lv.W = self.W0 if l == 0 else max(self.W0, Ef_max[l], Ev_max[l])  # halo columns the arrays hold
            Lv.b0, Lv.b1 = bounds[l][g], bounds[l][g + 1]
            Lv.lo, Lv.hi = max(0, Lv.b0 - Lv.W), min(dims[l], Lv.b1 + Lv.W)
            Lv.ncols = Lv.hi - Lv.lo
            Lv.o0, Lv.o1 = Lv.b0 - Lv.lo, Lv.b1 - Lv.lo                # owned, local
            Lv.jfix_lo = 0 if Lv.lo == 0 else -1
            Lv.jfix_hi = Lv.ncols - 1 if Lv.hi == dims[l] else -1
            if Lv.b1 - Lv.b0 < Lv.W and R > 1:
                raise ValueError("rank %d owns %d < %d columns of level %d" % (g, Lv.b1 - Lv.b0, Lv.W, l))
            Lv.f = self.be.zeros(Lv.ncols, Lv.m, phase=1)
            Lv.v = [self.be.zeros(Lv.ncols, Lv.m), self.be.zeros(Lv.ncols, Lv.m, phase=2)]
            Lv.cur = 0
            self.levels.append(Lv)
        # the tail's top level: the coarse columns this rank produces, padded to a common count
        # Every rank contributes `pad` consecutive columns of the tail's right-hand side
        # (its own, plus at most a few of a neighbour's that the receiver ignores), so the
        # all-gather needs no packing; one indexed copy puts the owners' columns in place.
        self.tail = self.be.make_tail(dims[nd], hs[nd], coarsest)
        self.tail_f = self.tail.f  # the tail plan's own right-hand side buffer (mt x mt)
        tb = bounds[nd]
        pad = max(tb[k + 1] - tb[k] for k in range(R))
        self.tail_pad = pad
        self.tail_start = [min(tb[k], mt - pad) for k in range(R)]
        rows = np.empty(mt, dtype=np.int64)
        for k in range(R):
            for j in range(tb[k], tb[k + 1]):
                rows[j] = k * pad + (j - self.tail_start[k])
        self.tail_rows = self.be.index_array(rows)
        self.tail_gathered = self.be.zeros(R * pad, mt) if not self.comm.solo else None
        self._prof = False
        # post the level-0 halo exchange between the edge columns and the interior columns
        self.overlap = os.environ.get("MGX_SLAB_OVERLAP", "1") != "0"

    # ---- data in / out ---------------------------------------------------------
    def set_problem(self, f_global, v_global):
        """f, v: numpy M x M arrays indexed [i, j]; every rank passes the same arrays
        (only its own columns are used)."""
        f_global = np.asarray(f_global, dtype=np.float64)
        v_global = np.asarray(v_global, dtype=np.float64)
        L0 = self.levels[0]
        L0.f = self.be.from_numpy(np.ascontiguousarray(f_global[:, L0.lo:L0.hi].T))
        L0.v = [self.be.from_numpy(np.ascontiguousarray(v_global[:, L0.lo:L0.hi].T)) for _ in range(2)]
        L0.cur = 0

    def set_random_rhs(self):
        """Benchmark input: the splitmix64 random right-hand side, zero guess, generated
        directly on each GPU for its own columns."""
        from . import rhs
        L0 = self.levels[0]
        L0.f.copy_(rhs.random_torch_cols(self.M, L0.lo, L0.hi, device=self.be.device))
        L0.v[0].zero_()
        L0.v[1].zero_()
        L0.cur = 0

    def owned_solution(self):
        """(first global column, array [i, j]) of the columns this rank owns."""
        L0 = self.levels[0]
        a = self.be.to_numpy(L0.v[L0.cur][L0.o0:L0.o1])
        return L0.b0, np.asfortranarray(a.T)

    def gather_solution(self):
        """The full M x M solution on every rank (test helper; moves the whole grid)."""
        L0 = self.levels[0]
        R = self.comm.size
        width = max(self.bounds[0][k + 1] - self.bounds[0][k] for k in range(R))
        pad = self.be.zeros(width, L0.m)
        pad[:L0.o1 - L0.o0] = L0.v[L0.cur][L0.o0:L0.o1]
        allc = self.be.to_numpy(self.comm.allgather(self.be.comm_tensor(pad)))
        out = np.empty((self.M, self.M), order="F")
        for k in range(R):
            a, b = self.bounds[0][k], self.bounds[0][k + 1]
            out[:, a:b] = allc[k * width:k * width + (b - a)].T
        return out

    def tune(self, Ndown, Nup, omega):
        """Measure the launch shapes of this rank's big launches now (mgx_tune_mode) by running
        three cycles whose level-0 iterate is put back afterwards.  Collective: every rank calls it."""
        if self.be.name != "hip":
            return
        L0 = self.levels[0]
        keep = L0.v[L0.cur].clone()
        L = self.be.L
        was = L.mgx_tune_mode(1)
        try:
            self.cycles(3, Ndown, Nup, omega)
        finally:
            L.mgx_tune_mode(was)
        L0.v[0].copy_(keep)
        L0.v[1].copy_(keep)
        L0.cur = 0
        self.be.synchronize()

    # ---- profiling hooks (same surface as solver.Plan) -----------------------------
    def profile(self, enable=True):
        self._prof = bool(enable)
        self._prof_recs = []
        self._prof_tot = {}
        self._prof_pts = {}

    def profile_totals(self):
        for name, a, b in getattr(self, "_prof_recs", []):
            c, ms = self._prof_tot.get(name, (0, 0.0))
            self._prof_tot[name] = (c + 1, ms + a.elapsed_time(b))
        self._prof_recs = []
        return dict(getattr(self, "_prof_tot", {}))

    def profile_points(self):
        """{kernel name: grid points produced by its timed launches} since profile(True)."""
        return dict(getattr(self, "_prof_pts", {}))

    def _launch(self, level, name, op):
        if self._prof and level == 0 and self.be.name == "hip":
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            self.be.launch(op)
            b.record()
            self._prof_recs.append((name, a, b))
            self._prof_pts[name] = self._prof_pts.get(name, 0) + (op.jhi - op.jlo + (op.jhi2 or 0) - (op.jlo2 or 0)) * op.m
        else:
            self.be.launch(op)

    def _launch_level0(self, name, make_op, items_after, flip):
        """A level-0 launch in up to three parts, so that the halo exchange that follows it
        overlaps most of it: the columns the neighbours are waiting for first, then the
        exchange is posted, then the interior columns, then the exchange is awaited.
        make_op(jrange) builds the launch for a column range; items_after() lists what to
        exchange (called after the parts that produce it are enqueued)."""
        L0 = self.levels[0]
        jlo, jhi = self._range(L0)
        ef1 = self._Ef[1] if (self.ndist >= 2 and getattr(self, "_Ef", None)) else 0
        B = max(2 * ef1, self.W0)
        B += B & 1
        lo_nb, hi_nb = L0.lo > 0, L0.hi < L0.m
        ilo, ihi = jlo, jhi
        edges = []
        if self.overlap and (lo_nb or hi_nb) and (jhi - jlo) >= 4 * B:
            if lo_nb:
                edges.append((jlo, jlo + B))
                ilo = jlo + B
            if hi_nb:
                edges.append((jhi - B, jhi))
                ihi = jhi - B
        # the columns next to both neighbours go out in ONE launch of two column ranges
        edge_op = None
        if edges:
            edge_op = make_op(edges[0])
            if len(edges) == 2:
                edge_op.jlo2, edge_op.jhi2 = edges[1]
        interior = make_op((ilo, ihi))  # built before the flip: it names the buffers explicitly
        if edge_op is not None:
            self._launch(0, "k_stream:edge", edge_op)
        if flip:
            L0.cur ^= 1                 # from here on v[cur] is the buffer being written
        if edges:
            handle = self.comm.exchange_start(items_after())
            self._launch(0, name, interior)
            self.comm.exchange_finish(handle)
        else:
            self._launch(0, name, interior)
            items = items_after()
            if items:
                self.comm.exchange(items)

    # ---- building blocks -----------------------------------------------------------
    def _range(self, Lv, extra=0, coarse_extra=None):
        """Local [jlo, jhi) of the columns to produce: the rank's own columns widened by
        `extra` on each interior side (or, with coarse_extra = e, exactly the fine columns
        whose restriction gives the rank's own coarse columns widened by e), clipped to the
        interior columns 1..m-2 of the level."""
        if coarse_extra is not None:
            glo, ghi = Lv.b0 + 1 - 2 * coarse_extra, Lv.b1 + 2 * coarse_extra
            if coarse_extra == 0:
                glo = Lv.b0
        else:
            glo, ghi = Lv.b0 - extra, Lv.b1 + extra
        glo, ghi = max(glo, 1), min(ghi, Lv.m - 1)
        return glo - Lv.lo, ghi - Lv.lo

    def _op(self, Lv, nsweeps, jrange, pro_stage=-1, norm_stage=-1, restrict_out=0, zero_in=0, store=1, omega=0.0,
            v_in=None, v_out=None, coarse=None, coarse_v=None, coarse_f=None, norm_slot=None):
        mc, nccols, gc0 = coarse if coarse is not None else (0, 0, 0)
        return SlabOp(nsweeps=nsweeps, pro_stage=pro_stage, norm_stage=norm_stage, restrict_out=restrict_out,
                      zero_in=zero_in, store=store, v_in=v_in, v_out=v_out, f=Lv.f, coarse_v=coarse_v,
                      coarse_f=coarse_f, norm_slot=norm_slot, m=Lv.m, ncols=Lv.ncols, mc=mc, nccols=nccols,
                      jlo=jrange[0], jhi=jrange[1], jfix_lo=Lv.jfix_lo, jfix_hi=Lv.jfix_hi, gf0=Lv.lo, gc0=gc0,
                      dx=Lv.dx, omega=omega)

    def _halo_ranges(self, Lv, width):
        """Send/receive column ranges that give each neighbour `width` of the other's own columns."""
        s_lo = (Lv.o0, Lv.o0 + width) if Lv.lo > 0 else None        # my first own columns -> rank-1
        r_lo = (Lv.o0 - width, Lv.o0) if Lv.lo > 0 else None        # rank-1's last own columns
        s_hi = (Lv.o1 - width, Lv.o1) if Lv.hi < Lv.m else None
        r_hi = (Lv.o1, Lv.o1 + width) if Lv.hi < Lv.m else None
        return s_lo, r_lo, s_hi, r_hi

    def _xv(self, Lv, width=None):
        """Exchange item for the live iterate of a level."""
        return (self.be.comm_tensor(Lv.v[Lv.cur]),) + self._halo_ranges(Lv, Lv.W if width is None else width)

    def _xf(self, Lv, width=None):
        return (self.be.comm_tensor(Lv.f),) + self._halo_ranges(Lv, Lv.W if width is None else width)

    def _coarse_of(self, l):
        """(coarse level or None for the tail, (mc, nccols, gc0)) of distributed level l."""
        if l + 1 < self.ndist:
            C = self.levels[l + 1]
            return C, (C.m, C.ncols, C.lo)
        return None, (self.dims[self.ndist], self.dims[self.ndist], 0)

    def _chain(self, level, name, Lv, pro, ns, res, zero_in, omega, coarse=None, coarse_v=None, coarse_f=None,
               norm_slot=None):
        """[pro] -> ns sweeps -> [res: 1 restrict, 2 norm] on the rank's own columns as launches
        of at most SLAB_MAX_SWEEPS sweeps; every launch that stores the iterate flips the
        level's ping-pong buffers, and between two launches of a chain the halo is exchanged."""
        remaining, first = ns, True
        jr = self._range(Lv)
        while True:
            k = min(remaining, _lib.SLAB_MAX_SWEEPS)
            remaining -= k
            last = remaining == 0
            pr = pro if first else 0
            rs = res if last else 0
            zi = zero_in if first else 0
            if k > 0 or pr or rs:
                store = 1 if (k > 0 or pr) else 0
                op = self._op(Lv, k, jr, pro_stage=0 if pr else -1, norm_stage=k if rs == 2 else -1,
                              restrict_out=1 if rs == 1 else 0, zero_in=zi, store=store, omega=omega,
                              v_in=Lv.v[Lv.cur], v_out=Lv.v[Lv.cur ^ 1], coarse=coarse,
                              coarse_v=coarse_v if pr else None, coarse_f=coarse_f if rs == 1 else None,
                              norm_slot=norm_slot if rs == 2 else None)
                self._launch(level, name, op)
                if store:
                    Lv.cur ^= 1
                    if not last:
                        self.comm.exchange([self._xv(Lv)])
            first = False
            if remaining <= 0:
                break

    def _tail(self, Ndown, Nup, omega):
        """All-gather the tail's right-hand side and run the tail on every rank."""
        self._gather_tail(self.tail_f)
        return self.tail.run(Ndown, Nup, omega)

    def _gather_tail(self, t):
        """t: an array of the tail's top level (one row per column) of which every rank holds
        its own columns; afterwards every rank holds all of them."""
        if not self.comm.solo:
            s = self.tail_start[self.comm.rank]
            mine = t[s:s + self.tail_pad]
            self.comm.allgather(self.be.comm_tensor(mine), out=self.be.comm_tensor(self.tail_gathered))
            self.be.take_rows(self.tail_gathered, self.tail_rows, out=t)

    # ---- the general schedule: a chain per leg, an exchange after every launch ----------
    def _cycle_general(self, Ndown, Nup, omega, slot, first=False):
        nd = self.ndist
        for l in range(nd):  # main.cpp:317-328
            Lv = self.levels[l]
            C, cshape = self._coarse_of(l)
            cf = C.f if C is not None else self.tail_f
            items = []
            if l >= 1 and Ndown == 0:
                # the iterate is zero here: f -> coarse f, nothing stored
                op = self._op(Lv, 0, self._range(Lv), restrict_out=1, zero_in=1, store=0, omega=omega,
                              coarse=cshape, coarse_f=cf)
                self._launch(l, "k_stream:down", op)
            else:
                self._chain(l, "k_stream:down", Lv, 0, Ndown, 1, 1 if l >= 1 else 0, omega, coarse=cshape,
                            coarse_f=cf)
                items.append(self._xv(Lv))
            if C is not None:
                items.append(self._xf(C))
            if items:
                self.comm.exchange(items)
        vt = self._tail(Ndown, Nup, omega)
        if first:
            self._normalize_boundary()
        for l in range(nd - 1, -1, -1):  # main.cpp:335-346
            Lv = self.levels[l]
            C, cshape = self._coarse_of(l)
            cv = C.v[C.cur] if C is not None else vt
            zero_in = 1 if (l >= 1 and Ndown == 0) else 0
            res = 2 if (l == 0 and slot is not None) else 0
            self._chain(l, "k_stream:up", Lv, 1, Nup, res, zero_in, omega, coarse=cshape, coarse_v=cv,
                        norm_slot=slot)
            self.comm.exchange([self._xv(Lv)])

    # ---- the lean schedule: one exchange and one all-gather per cycle ---------------------
    def _lean_items(self, with_v=True):
        """What a lean cycle exchanges: the level-0 iterate and the level-1 right-hand side."""
        L0 = self.levels[0]
        items = [self._xv(L0, self.W0)] if with_v else []
        if self.ndist >= 2:
            items.append(self._xf(self.levels[1], self._Ef[1]))
        return items

    def _lean_down0(self, Ndown, omega):
        """Plain level-0 down leg (first cycle of a call), then the cycle's exchange."""
        L0 = self.levels[0]
        C, cshape = self._coarse_of(0)
        cf = C.f if C is not None else self.tail_f
        store = 1 if Ndown > 0 else 0
        vin, vout = L0.v[L0.cur], L0.v[L0.cur ^ 1]

        def make_op(jr):
            return self._op(L0, Ndown, jr, restrict_out=1, store=store, omega=omega, v_in=vin, v_out=vout,
                            coarse=cshape, coarse_f=cf)

        self._launch_level0("k_stream:down", make_op, lambda: self._lean_items(bool(store)), flip=bool(store))

    def _lean_coarse(self, Ndown, Nup, omega):
        """Levels 1..nd-1 down, the tail, levels nd-1..1 up: no communication but the tail's
        all-gather; halo columns are recomputed."""
        nd = self.ndist
        Ev, Ef = self._Ev, self._Ef
        for l in range(1, nd):
            Lv = self.levels[l]
            C, cshape = self._coarse_of(l)
            cf = C.f if C is not None else self.tail_f
            jr = self._range(Lv, coarse_extra=Ef[l + 1] if l + 1 < nd else 0)
            op = self._op(Lv, Ndown, jr, restrict_out=1, zero_in=1, store=0, omega=omega, coarse=cshape, coarse_f=cf)
            self._launch(l, "k_stream:down", op)
        vt = self._tail(Ndown, Nup, omega)
        for l in range(nd - 1, 0, -1):
            Lv = self.levels[l]
            C, cshape = self._coarse_of(l)
            cv = C.v[C.cur] if C is not None else vt
            op = self._op(Lv, Ndown + Nup, self._range(Lv, extra=Ev[l]), pro_stage=Ndown, zero_in=1, store=1,
                          omega=omega, v_out=Lv.v[0], coarse=cshape, coarse_v=cv)
            Lv.cur = 0
            self._launch(l, "k_stream:up", op)
        return vt

    def _normalize_boundary(self):
        """main.cpp:341 adds an array whose boundary is +0.0 to the fine iterate: Dirichlet values
        stay, except that -0.0 becomes +0.0.  The launches write interior points only, so the
        boundary rows (and the physical boundary columns this rank holds) get it here."""
        L0 = self.levels[0]
        for v in L0.v:
            v[:, 0] += 0.0
            v[:, L0.m - 1] += 0.0
            if L0.jfix_lo >= 0:
                v[L0.jfix_lo, :] += 0.0
            if L0.jfix_hi >= 0:
                v[L0.jfix_hi, :] += 0.0

    def _lean_up0(self, Ndown, Nup, omega, vt, slot, next_down):
        """Level-0 up leg with the cycle's residual norm; with next_down also the down leg
        of the following cycle (one launch), then that cycle's exchange."""
        L0 = self.levels[0]
        C, cshape = self._coarse_of(0)
        cv = C.v[C.cur] if C is not None else vt
        cf = C.f if C is not None else self.tail_f
        vin, vout = L0.v[L0.cur], L0.v[L0.cur ^ 1]
        if next_down:
            def make_op(jr):
                return self._op(L0, Nup + Ndown, jr, pro_stage=0, norm_stage=Nup, restrict_out=1, store=1,
                                omega=omega, v_in=vin, v_out=vout, coarse=cshape, coarse_v=cv, coarse_f=cf,
                                norm_slot=slot)

            self._launch_level0("k_stream:updown", make_op, lambda: self._lean_items(True), flip=True)
        else:
            def make_op(jr):
                return self._op(L0, Nup, jr, pro_stage=0, norm_stage=Nup, store=1, omega=omega, v_in=vin,
                                v_out=vout, coarse=cshape, coarse_v=cv, norm_slot=slot)

            self._launch_level0("k_stream:up", make_op, lambda: [self._xv(L0, self.W0)], flip=True)

    def cycles(self, Nv, Ndown, Nup, omega, pureJacobi=False):
        """Nv cycles (main.cpp:309-354).  Returns MGOutputs; Times is the cumulative device time
        of whole cycles on this rank's stream (events, like the single-device fused schedule;
        exchanges are part of a cycle); with the CPU test backend it is host time."""
        be = self.be
        slots = be.slots(Nv + 1)
        L0 = self.levels[0]
        lean = 1 <= Ndown + Nup <= MAX_LEAN and not pureJacobi
        if lean:
            self._Ev, self._Ef = halo_plan(self.ndist, Ndown, Nup)
            for l in range(1, self.ndist):
                if max(self._Ef[l], self._Ev[l]) > self.levels[l].W:
                    raise RuntimeError("level %d halo %d exceeds the allocated %d" %
                                       (l, max(self._Ef[l], self._Ev[l]), self.levels[l].W))
        # main.cpp:307: the norm of the initial residual (halo columns of v must be current)
        self.comm.exchange([self._xv(L0, self.W0)])
        self._chain(0, "k_stream:norm", L0, 0, 0, 2, 0, omega, norm_slot=be.slot_ptr(slots, 0))
        times = np.zeros(Nv + 1)
        on_gpu = be.name == "hip"
        marks = []
        if on_gpu:
            marks = [torch.cuda.Event(enable_timing=True) for _ in range(Nv + 1)]
            marks[0].record()
        for it in range(Nv):
            t0 = time.perf_counter()
            slot = be.slot_ptr(slots, it + 1)
            if pureJacobi:  # main.cpp:348 and :351 in one pass
                self._chain(0, "k_stream:up", L0, 0, 1, 2, 0, omega, norm_slot=slot)
                self.comm.exchange([self._xv(L0, self.W0)])
            elif lean:
                if it == 0:  # later cycles: the previous cycle's level-0 launch did the down leg
                    self._lean_down0(Ndown, omega)
                vt = self._lean_coarse(Ndown, Nup, omega)
                if it == 0:
                    self._normalize_boundary()
                self._lean_up0(Ndown, Nup, omega, vt, slot, next_down=it + 1 < Nv)
            else:
                self._cycle_general(Ndown, Nup, omega, slot, first=it == 0)
            if on_gpu:
                marks[it + 1].record()
            else:
                times[it + 1] = times[it] + (time.perf_counter() - t0)
        self.comm.allreduce_max_(slots)
        be.synchronize()
        for it in range(Nv if on_gpu else 0):
            times[it + 1] = times[it] + 1e-3 * marks[it].elapsed_time(marks[it + 1])
        return MGOutputs(be.decode_slots(slots), times)

    # ---- full multigrid (an extension; the reference has no such driver) ------------------
    def _view(self, l):
        """This solver restricted to the levels from l downwards: level l becomes level 0."""
        if l == 0:
            return self
        sub = object.__new__(SlabSolver)
        sub.__dict__.update(self.__dict__)
        sub.levels = self.levels[l:]
        sub.ndist = self.ndist - l
        sub.dims, sub.hs = self.dims[l:], self.hs[l:]
        sub.M, sub.dx, sub.depth = self.dims[l], self.hs[l], self.depth - l
        sub.bounds = {k - l: b for k, b in self.bounds.items() if k >= l}
        sub._prof = False
        return sub

    @staticmethod
    def _inject(src, F, dst, c0, c1, row0, mc, first, last):
        """Dirichlet values of a coarse level from the next finer one, coarse boundary point k
        <- fine boundary point 2k, for the coarse columns [c0, c1) this rank owns (stored in
        dst from row `row0` on); src is the fine slab F.  The two physical boundary columns go
        with the ranks that hold them."""
        mf = F.m
        fl0 = 2 * c0 - F.lo
        n = c1 - c0
        dst[row0:row0 + n, 0] = src[fl0:fl0 + 2 * n:2, 0]
        dst[row0:row0 + n, mc - 1] = src[fl0:fl0 + 2 * n:2, mf - 1]
        if first:
            dst[row0, :] = src[0, ::2]
        if last:
            dst[row0 + n - 1, :] = src[F.ncols - 1, ::2]

    def fmg(self, ncyc, Ndown, Nup, omega):
        """EXTENSION: full multigrid over the slabs -- the composition mgx_plan_fmg runs on one
        GPU.  The right-hand side is restricted to every level, the Dirichlet values of v are
        injected, the replicated tail runs its own full multigrid, and on the way up each
        distributed level starts from the interpolated solution of the level below and runs
        ncyc V(Ndown,Nup) cycles on the sub-hierarchy below it.  Only the boundary values of v
        are used.  Returns the finest level's MGOutputs.  (A coarse value of -0.0 is
        interpolated as +0.0: the kernel adds the interpolant to a zero iterate.)"""
        nd = self.ndist
        R, g = self.comm.size, self.comm.rank
        for l in range(nd):
            Lv = self.levels[l]
            C, cshape = self._coarse_of(l)
            cf = C.f if C is not None else self.tail_f
            # the residual of a zero iterate is the right-hand side itself: restriction of f
            op = self._op(Lv, 0, self._range(Lv), restrict_out=1, zero_in=1, store=0, omega=omega, coarse=cshape,
                          coarse_f=cf)
            self._launch(l, "k_stream:down", op)
            src = Lv.v[Lv.cur]
            if C is not None:
                self.comm.exchange([self._xf(C)])
                C.v[0][...] = 0.0
                self._inject(src, Lv, C.v[0], C.b0, C.b1, C.o0, C.m, C.lo == 0, C.hi == C.m)
                self.comm.exchange([(self.be.comm_tensor(C.v[0]),) + self._halo_ranges(C, C.W)])
                C.v[1][...] = C.v[0]
                C.cur = 0
            else:
                tb = self.bounds[nd]
                mt = self.dims[nd]
                tv = self.tail.v
                tv[...] = 0.0
                self._inject(src, Lv, tv, tb[g], tb[g + 1], tb[g], mt, g == 0, g == R - 1)
                self._gather_tail(tv)
        self._gather_tail(self.tail_f)
        vt = self.tail.fmg(ncyc, Ndown, Nup, omega)
        out = None
        for l in range(nd - 1, -1, -1):
            Lv = self.levels[l]
            C, cshape = self._coarse_of(l)
            cv = C.v[C.cur] if C is not None else vt
            # initial guess: the interpolated solution of the level below (boundary values stay)
            op = self._op(Lv, 0, self._range(Lv), pro_stage=0, zero_in=1, store=1, omega=omega, v_out=Lv.v[Lv.cur],
                          coarse=cshape, coarse_v=cv)
            self._launch(l, "k_stream:up", op)
            # from here on the level below holds coarse-grid corrections: zero, zero boundary
            if C is not None:
                C.v[0][...] = 0.0
                C.v[1][...] = 0.0
                C.cur = 0
            else:
                self.tail.reset()
            out = self._view(l).cycles(ncyc, Ndown, Nup, omega)
        return out

    def close(self):
        if self.tail is not None:
            self.tail.close()
            self.tail = None
