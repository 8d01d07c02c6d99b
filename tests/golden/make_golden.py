"""Regenerates the committed fixtures in tests/golden/.

  reference_known_answers.json  a few dozen scalars COMPUTED by the reference's own solver
                                functions (RoyJames/MultiGrid main.cpp:147-357, compiled from
                                the reference checkout into oracle/_ref/librefops.so by
                                oracle/Makefile, -O2 -DNDEBUG -ffp-contract=off) on a 9x9
                                and a 257x257 case.
  reference_vectors.npz         seeded inputs and the outputs the same library gave for them:
                                every operator at sizes 3..65 and whole MultiGrid() runs with
                                non-zero Dirichlet data, -0.0 boundary values, coarsest != 2,
                                an even coarsest level, V(5,5), V(0,0), pureJacobi.  This is
                                what pins the CPU oracle where the reference checkout is absent
                                (the GPU box).
  f_smooth_257.npy              input bytes of the 257^2 smooth right-hand side of the 257^2
                                known answers (exp() is libm-dependent, so the bytes are the
                                fixture; kept if present).
  oracle_vectors_33.npz         small per-operator vectors produced by the CPU oracle
                                (oracle/mg_oracle.c) from seeded inputs: a frozen copy so that
                                a later edit of the oracle cannot drift silently.

Needs the reference checkout (make -C oracle ref).  Run from the repo root:
    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def g17(x):
    return "%.17g" % float(x)


def case9():
    m = 9
    f = np.zeros((m, m), order="F")
    for i in range(1, m - 1):
        for j in range(1, m - 1):
            f[i, j] = 1 + i + 10 * j
    return m, 1.0 / 8, f


def known_answers(R, f257):
    """The known-answer set, computed by the reference (R = oracle.Ref())."""
    m, h, f = case9()
    u = R.relax(np.zeros((m, m), order="F"), f, h, 1, 0.6)
    r = R.residual(u, f, h)
    c = R.coarsen(r)
    e13 = np.zeros((9, 9), order="F")
    e13[1, 3] = 1.0
    e31 = np.zeros((9, 9), order="F")
    e31[3, 1] = 1.0
    cc = np.zeros((5, 5), order="F")
    coarse_in = {"(1,1)": 1, "(2,1)": 2, "(1,2)": 3, "(2,2)": 4}
    for key, val in coarse_in.items():
        a, b = key.strip("()").split(",")
        cc[int(a), int(b)] = val
    fine = R.refine(cc)
    sol9, rn9, _ = R.multigrid(f, np.zeros((m, m), order="F"), h, 5, 3, 3, 0.6, 2)
    M = 257
    sol, rn, _ = R.multigrid(f257, np.zeros((M, M), order="F"), 1.0 / (M - 1), 30, 2, 2, 0.6, 2)
    return {
        "_provenance": "Computed by tests/golden/make_golden.py from the reference's own functions "
                       "(main.cpp:147-357 compiled from the reference checkout, oracle/_ref/librefops.so, "
                       "-O2 -DNDEBUG -ffp-contract=off). Decimal strings are %.17g prints of the "
                       "reference's doubles; tests parse them with float() and require exact equality.",
        "case9": {
            "m": 9, "dx": "1/8", "f": "f(i,j)=1+i+10*j on the interior, 0 on the boundary", "u0": "0",
            "relax_1_0.6": {"(1,1)": g17(u[1, 1]), "(3,5)": g17(u[3, 5])},
            "residual_after_relax": {"(1,1)": g17(r[1, 1]), "(4,4)": g17(r[4, 4]), "norm": g17(R.calc_norm(r))},
            "coarsen_of_residual": {"(1,1)": g17(c[1, 1]), "(2,3)": g17(c[2, 3])},
            "coarsen_indicator": {"fine(1,3)->c(1,1)": g17(R.coarsen(e13)[1, 1]),
                                  "fine(3,1)->c(1,1)": g17(R.coarsen(e31)[1, 1])},
            "refine": {"coarse": coarse_in,
                       "fine": {k: g17(fine[int(k[1]), int(k[3])]) for k in ("(2,2)", "(3,2)", "(2,3)", "(3,3)", "(1,1)")}},
            "multigrid_5_3_3_0.6_2": {"ResNorms": [g17(x) for x in rn9], "Sol(4,4)": g17(sol9[4, 4])},
        },
        "case257": {
            "m": 257, "dx": "1/256",
            "f": "tests/golden/f_smooth_257.npy : -(x^2+y^2)*exp(x*y), x=i*dx, y=j*dx, all i,j (numpy)",
            "args": "-v 30 -b 2 -a 2 (omega 0.6, coarsest 2)",
            "ResNorms[0..5]": [g17(x) for x in rn[:6]],
            "ResNorms[30]": g17(rn[30]),
            "first_cycle_le_1e-8": int(np.argmax(rn <= 1e-8)),
            "max(Sol)": g17(sol.max()),
        },
    }


# Whole-solver cases of reference_vectors.npz: (name, m, Nv, Ndown, Nup, omega, coarsest, pureJacobi, data)
# data: "zero" = zero guess and boundary (what main.cpp:420-439 passes), "dirichlet" = random
# boundary values and a random interior guess, "negzero" = -0.0 on the whole boundary.
MG_CASES = [
    ("mg33_v22", 33, 4, 2, 2, 0.6, 2, False, "zero"),
    ("mg33_dirichlet", 33, 3, 2, 2, 0.6, 2, False, "dirichlet"),
    ("mg65_negzero_v33", 65, 3, 3, 3, 0.6, 2, False, "negzero"),
    ("mg65_v12_w08", 65, 3, 1, 2, 0.8, 2, False, "dirichlet"),
    ("mg33_v55", 33, 2, 5, 5, 0.7, 2, False, "dirichlet"),
    ("mg33_v00", 33, 2, 0, 0, 0.6, 2, False, "dirichlet"),
    ("mg33_coarsest4", 33, 3, 2, 2, 0.6, 4, False, "dirichlet"),     # 33,17,9,5: 5x5 coarsest, direct_solve a no-op
    ("mg49_coarsest6", 49, 3, 2, 2, 0.6, 6, False, "dirichlet"),     # 49,25,13,7
    ("mg45_coarsest11", 45, 3, 2, 2, 0.6, 11, False, "dirichlet"),   # 45,23,12: an EVEN coarsest level
    ("mg49_coarsest3", 49, 3, 2, 1, 0.6, 3, False, "negzero"),       # 49,25,13,7,4: even coarsest level
    ("mg17_shallow", 17, 3, 2, 2, 0.6, 8, False, "dirichlet"),       # depth 1
    ("mg33_purejacobi", 33, 5, 2, 2, 0.6, 2, True, "dirichlet"),
]
OP_SIZES = [3, 5, 9, 17, 33, 65]


def mg_inputs(name, m, data):
    seed = int.from_bytes(name.encode(), "little") % (2 ** 32)
    rng = np.random.default_rng(seed)
    f = np.asfortranarray(rng.standard_normal((m, m)) * 50.0)
    if data == "zero":
        v = np.zeros((m, m), order="F")
    else:
        v = np.asfortranarray(rng.standard_normal((m, m)))
        if data == "negzero":
            v[0, :] = -0.0
            v[-1, :] = -0.0
            v[:, 0] = -0.0
            v[:, -1] = -0.0
    return f, v


def reference_vectors(R):
    out = {}
    rng = np.random.default_rng(20250101)
    for m in OP_SIZES:
        dx = 1.0 / (m - 1)
        u = np.asfortranarray(rng.standard_normal((m, m)))
        f = np.asfortranarray(rng.standard_normal((m, m)) * 100.0)
        u[0, 0] = -0.0
        k = "op%d_" % m
        out[k + "u"], out[k + "f"] = u, f
        out[k + "relax1_w06"] = R.relax(u, f, dx, 1, 0.6)
        out[k + "relax3_w09"] = R.relax(u, f, dx, 3, 0.9)
        out[k + "residual"] = R.residual(u, f, dx)
        out[k + "norm"] = np.float64(R.calc_norm(u))
        out[k + "direct"] = R.direct_solve(u, f, dx)
        if m >= 5:
            out[k + "coarsen"] = R.coarsen(u)
        mc = (m - 1) // 2 + 1
        if m >= 5:
            c = np.asfortranarray(rng.standard_normal((mc, mc)))
            out[k + "c"] = c
            out[k + "refine"] = R.refine(c, m)
    # level transfers between an odd and an EVEN level (23 <-> 12, 7 <-> 4), legal in the reference
    for (mf, mc) in [(23, 12), (7, 4)]:
        fine = np.asfortranarray(rng.standard_normal((mf, mf)))
        c = np.asfortranarray(rng.standard_normal((mc, mc)))
        k = "even%d_" % mf
        out[k + "fine"], out[k + "c"] = fine, c
        out[k + "coarsen"] = R.coarsen(fine, mc)
        out[k + "refine"] = R.refine(c, mf)
    for (name, m, Nv, nd, nu, om, co, pj, data) in MG_CASES:
        f, v = mg_inputs(name, m, data)
        sol, rn, _ = R.multigrid(f, v, 1.0 / (m - 1), Nv, nd, nu, om, co, pj)
        out[name + "_f"], out[name + "_v"] = f, v
        out[name + "_sol"], out[name + "_resnorms"] = sol, rn
        out[name + "_depth"] = np.int64(R.depth(m, co))
    out["depth_table"] = np.array([[m, c, R.depth(m, c)] for m in (3, 5, 9, 17, 33, 45, 49, 177, 257, 1025, 2817, 4097,
                                                                   16385, 22529, 32769, 45057, 65537)
                                   for c in (1, 2, 3, 4, 6, 8, 11)], dtype=np.int64)
    return out


def main():
    from oracle import oracle as O
    R = O.Ref()

    p257 = os.path.join(HERE, "f_smooth_257.npy")
    if not os.path.exists(p257):
        M = 257
        h = 1.0 / (M - 1)
        x = np.arange(M) * h
        X, Y = np.meshgrid(x, x, indexing="ij")
        np.save(p257, np.asfortranarray(-(X ** 2 + Y ** 2) * np.exp(X * Y)))
    f257 = np.load(p257)

    with open(os.path.join(HERE, "reference_known_answers.json"), "w") as fh:
        json.dump(known_answers(R, f257), fh, indent=1)
    np.savez_compressed(os.path.join(HERE, "reference_vectors.npz"), **reference_vectors(R))

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
