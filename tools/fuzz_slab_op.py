"""Randomised sweep of mgx_slab_launch against the numpy emulation of one slab launch
(tests/slab_numpy_backend.py): random slab geometry, column ranges, physical boundaries,
global offsets of both parities and every kind of leg.  Not part of the test suite.
   python -u tools/fuzz_slab_op.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from multigrid_amd import slab
from oracle import oracle as O
from slab_numpy_backend import NumpyBackend, _decode

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
hip, ref = slab.HipBackend(), NumpyBackend(O)
BIG = os.environ.get("FUZZ_BIG") == "1"  # bigger slabs: interior strips, long chunks (slower emulation)
t_end, t_say, n = time.time() + budget, time.time() + 30, 0
# (nsweeps, pro_stage, norm_stage, restrict_out, zero_in, store): the legs the schedules compose
def random_leg():
    kind = int(rng.integers(0, 6))
    ns = int(rng.integers(0, 5))
    if kind == 0:
        z = int(rng.integers(0, 2))
        return (ns, -1, -1, 1, z, 0 if (z and rng.integers(0, 2)) or ns == 0 else 1)      # down
    if kind == 1:
        return (max(ns, 1), -1, -1, 0, int(rng.integers(0, 2)), 1)                          # sweeps only
    if kind == 2:
        return (ns, -1, ns, 0, 0, 1 if ns else 0)                                          # sweeps + norm
    if kind == 3:
        return (ns, 0, ns if rng.integers(0, 2) else -1, 0, int(rng.integers(0, 2)), 1)    # up
    if kind == 4:
        ns = max(ns, 1)
        return (ns, 0, int(rng.integers(0, ns + 1)), 1, 0, 1)                              # up + down
    ns = max(ns, 1)
    return (ns, int(rng.integers(1, ns + 1)), -1, 0, 1, 1)                                 # recomputing up

while time.time() < t_end:
    if time.time() > t_say:
        print("... %d cases so far" % n, flush=True)
        t_say = time.time() + 30
    ns, pa, nm, res, zin, st = random_leg()
    halo = ns + (2 if res else (1 if nm == ns and nm >= 0 else 0))
    mc = int(rng.integers(3, 1500 if BIG else 200))
    m = 2 * mc - 1
    left, right = bool(rng.integers(0, 3) == 0), bool(rng.integers(0, 3) == 0)
    own = int(rng.integers(1, 700 if BIG else 60))
    extra_lo, extra_hi = int(rng.integers(0, 4)), int(rng.integers(0, 4))
    jlo = 1 if left else halo + extra_lo
    jhi = jlo + own
    ncols = jhi + (1 if right else halo + extra_hi)
    gf0 = 0 if left else int(rng.integers(1, 40))
    if right and (gf0 + ncols - 1) % 2:  # a physical boundary column has an even global index (m is odd)
        if left:
            jhi, ncols = jhi + 1, ncols + 1
        else:
            gf0 += 1
    gc0 = max((gf0 >> 1) - 3, 0) if not left else 0
    nccols = ((gf0 + ncols) >> 1) + 4 - gc0
    f = rng.standard_normal((ncols, m))
    vin = rng.standard_normal((ncols, m))
    vout0 = rng.standard_normal((ncols, m))
    cv = rng.standard_normal((nccols, mc))
    cf0 = rng.standard_normal((nccols, mc))
    dx, omega = float(rng.choice([1.0, 0.01, 0.37])), float(rng.choice([0.6, 0.8, 1.0]))
    # every third case produces TWO column ranges in the one launch (the bands next to both
    # neighbours of a slab): [jlo, a) and [b, jhi) with a gap between them
    j1, j2lo, j2hi = jhi, 0, 0
    if own >= 4 and rng.integers(0, 3) == 0:
        a = jlo + int(rng.integers(0, own // 2 + 1))        # the first range may be empty
        b = a + int(rng.integers(1, jhi - a))
        j1, j2lo, j2hi = a, b, jhi
    outs = []
    for be in (hip, ref):
        conv = be.from_numpy
        slots = be.slots(1)
        vo, cf = conv(vout0), conv(cf0)
        op = slab.SlabOp(nsweeps=ns, pro_stage=pa, norm_stage=nm, restrict_out=res, zero_in=zin, store=st,
                         v_in=None if zin else conv(vin), v_out=vo if st else None, f=conv(f),
                         coarse_v=conv(cv) if pa >= 0 else None, coarse_f=cf if res else None,
                         norm_slot=be.slot_ptr(slots, 0) if nm >= 0 else None, m=m, ncols=ncols, mc=mc, nccols=nccols,
                         jlo=jlo, jhi=j1, jlo2=j2lo, jhi2=j2hi, jfix_lo=0 if left else -1,
                         jfix_hi=ncols - 1 if right else -1, gf0=gf0, gc0=gc0, dx=dx, omega=omega)
        be.launch(op)
        be.synchronize()
        outs.append((be.to_numpy(vo), be.to_numpy(cf), be.decode_slots(slots)[0]))
    (va, ca, na), (vb, cb, nb) = outs
    same = np.array_equal(va.view(np.uint64), vb.view(np.uint64)) and np.array_equal(ca.view(np.uint64), cb.view(np.uint64)) \
        and (na == nb or (nm < 0))
    n += 1
    if not same:
        print("MISMATCH", dict(leg=(ns, pa, nm, res, zin, st), m=m, ncols=ncols, jlo=jlo, jhi=j1, jlo2=j2lo, jhi2=j2hi, left=left, right=right, gf0=gf0,
                               gc0=gc0, nccols=nccols, dx=dx, omega=omega))
        print(" v differs at", np.argwhere(va != vb)[:6].tolist(), " coarse f differs at", np.argwhere(ca != cb)[:6].tolist(),
              " norms", na, nb)
        sys.exit(1)
print("slab-op fuzz: %d random launches bit-identical to the numpy emulation" % n)
