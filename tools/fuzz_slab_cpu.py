"""Randomised multi-rank sweep of the slab driver on CPU (gloo, numpy backend; not part of
the test suite):   python tools/fuzz_slab_cpu.py [seconds] [seed] [numpy|hip]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import numpy as np
from test_slab_cpu import run_ranks

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
backend = sys.argv[3] if len(sys.argv) > 3 else "numpy"  # "hip": the ranks share the GPU of the box
t_end = time.time() + budget
t_say = time.time() + 30
n = refused = 0
while time.time() < t_end:
    if time.time() > t_say:
        print("... %d cases so far" % n, flush=True)
        t_say = time.time() + 30
    world = int(rng.integers(2, 6 if backend == "numpy" else 5))
    k = int(rng.choice([1, 1, 1, 3, 5, 11]))
    d = int(rng.integers(7, 11))
    M = k * 2 ** d + 1
    if M > 1100:
        continue
    coarsest = int(rng.choice([2, 2, 2, k, 2 * k, 4]))
    tail_m = int(rng.choice([9, 17, 33, 65, 129]))
    nd, nu = int(rng.integers(0, 5)), int(rng.integers(0, 5))
    if rng.integers(0, 4) == 0:
        nd, nu = int(rng.integers(0, 8)), int(rng.integers(0, 8))
    Nv = int(rng.integers(1, 4))
    fmg = int(rng.integers(0, 3) == 0)
    exact = int(rng.integers(0, 3) == 0)
    pj = int(not fmg and rng.integers(0, 10) == 0)
    args = [backend, M, Nv, nd, nu, float(rng.choice([0.6, 0.8, 1.0])), tail_m, pj, coarsest, exact, fmg]
    print("case", world, args, flush=True)
    try:
        codes, outs = run_ranks(world, args, timeout=int(os.environ.get("FUZZ_CASE_TIMEOUT", "600")))
    except Exception as e:  # a rank that never finishes
        print("TIMEOUT/ERROR", world, args, repr(e), flush=True)
        sys.exit(2)
    if all(c == 0 for c in codes):
        n += 1
    elif any("is not split over" in o or "owns" in o or "refused the hierarchy" in o or "depth" in o or "indexes out of range" in o for o in outs):
        refused += 1  # a configuration the driver (or the oracle) declines, on every rank alike
    else:
        print("FAILED", world, args)
        print("\n".join(o[-600:] for o in outs))
        sys.exit(1)
print("slab fuzz: %d random multi-rank cases bit-identical to the oracle, %d declined" % (n, refused))
