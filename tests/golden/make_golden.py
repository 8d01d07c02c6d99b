"""Regenerates the committed fixtures in tests/golden/.

  reference_known_answers.json  values printed by the reference itself
                                (RoyJames/MultiGrid main.cpp, -O2 -ffp-contract=off)
                                when it was surveyed; they are typed in below, not
                                computed, because main.cpp cannot be rebuilt in this
                                image (boost/program_options.hpp is absent).
  f_smooth_257.npy              input bytes of the 257^2 smooth right-hand side the
                                257^2 known answers were produced from (exp() is
                                libm-dependent, so the bytes are the fixture).
  oracle_vectors_33.npz         small per-operator vectors produced by the CPU oracle
                                (oracle/mg_oracle.c) from seeded inputs: a frozen copy
                                so that a later edit of the oracle cannot drift silently.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

KNOWN = {
    "_provenance": "Printed by the reference itself (main.cpp built -O2 -ffp-contract=off) when it was "
                   "surveyed. Decimal strings are %.17g prints of the reference's doubles; tests parse "
                   "them with float() and require exact equality.",
    "case9": {
        "m": 9, "dx": "1/8", "f": "f(i,j)=1+i+10*j on the interior, 0 on the boundary", "u0": "0",
        "relax_1_0.6": {"(1,1)": "-0.028124999999999997", "(3,5)": "-0.12656249999999999"},
        "residual_after_relax": {"(1,1)": "10.050000000000001", "(4,4)": "45", "norm": "67"},
        "coarsen_of_residual": {"(1,1)": "21.228124999999999", "(2,3)": "61.475000000000001"},
        "coarsen_indicator": {"fine(1,3)->c(1,1)": "0", "fine(3,1)->c(1,1)": "0.125"},
        "refine": {"coarse": {"(1,1)": 1, "(2,1)": 2, "(1,2)": 3, "(2,2)": 4},
                   "fine": {"(2,2)": "1", "(3,2)": "1.5", "(2,3)": "2", "(3,3)": "2.5", "(1,1)": "0.25"}},
        "multigrid_5_3_3_0.6_2": {
            "ResNorms": ["78", "13.306307323435817", "2.384344466143034", "0.40013521643743388",
                         "0.067322667063876906", "0.011695676067120075"],
            "Sol(4,4)": "-3.274802582413169"},
    },
    "case257": {
        "m": 257, "dx": "1/256",
        "f": "tests/golden/f_smooth_257.npy : -(x^2+y^2)*exp(x*y), x=i*dx, y=j*dx, all i,j (numpy)",
        "args": "-v 30 -b 2 -a 2 (omega 0.6, coarsest 2)",
        "ResNorms[0..5]": ["5.352277336483516", "2.0327668938236076", "0.6084966535668475",
                           "0.180776838287648", "0.053643159519877415", "0.016201320368280037"],
        "ResNorms[30]": "8.938849660466985e-12",
        "first_cycle_le_1e-8": 17,
        "max(Sol)": "0.07067014503834934",
    },
}


def main():
    with open(os.path.join(HERE, "reference_known_answers.json"), "w") as fh:
        json.dump(KNOWN, fh, indent=1)

    M = 257
    h = 1.0 / (M - 1)
    x = np.arange(M) * h
    X, Y = np.meshgrid(x, x, indexing="ij")
    F = np.asfortranarray(-(X ** 2 + Y ** 2) * np.exp(X * Y))
    np.save(os.path.join(HERE, "f_smooth_257.npy"), F)

    from oracle import oracle as O
    rng = np.random.default_rng(20240611)
    m = 33
    dx = 1.0 / (m - 1)
    u = np.asfortranarray(rng.standard_normal((m, m)))
    f = np.asfortranarray(rng.standard_normal((m, m)) * 100.0)
    mc = (m - 1) // 2 + 1
    c = np.asfortranarray(rng.standard_normal((mc, mc)))
    out = {
        "u": u, "f": f, "c": c, "dx": np.float64(dx),
        "relax3_w06": O.relax(u, f, dx, 3, 0.6),
        "relax1_w09": O.relax(u, f, dx, 1, 0.9),
        "residual": O.residual(u, f, dx),
        "coarsen_u": O.coarsen(u),
        "refine_c": O.refine(c),
        "norm_u": np.float64(O.calc_norm(u)),
    }
    v0 = np.zeros((m, m), order="F")
    sol, rn, _ = O.multigrid(f, v0, dx, 8, 2, 2, 0.6, 2)
    out["mg_sol"] = sol
    out["mg_resnorms"] = rn
    np.savez_compressed(os.path.join(HERE, "oracle_vectors_33.npz"), **out)
    print("fixtures written to", HERE)


if __name__ == "__main__":
    main()
