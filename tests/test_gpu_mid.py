"""GPU parity tests of the grouped small-level launches (csrc/mgx_mid.hip): several levels
of a cycle per launch, each workgroup recomputing in LDS whatever its tile of the output
depends on.  Sol and ResNorms must equal the CPU oracle bit for bit for every way of
cutting the levels into groups, every tile size and every sweep count of the lean schedule,
and equal what the plan computes with the grouping switched off."""
import numpy as np
import pytest

from conftest import assert_bits

pytestmark = pytest.mark.gpu

KNOBS = ("MGX_NO_MID", "MGX_NO_TAIL", "MGX_MID_MAX_M", "MGX_MID_DOWN_NL", "MGX_MID_UP_NL",
         "MGX_MID_DOWN_TI", "MGX_MID_DOWN_TJ", "MGX_MID_UP_TI", "MGX_MID_UP_TJ", "MGX_MID_LDS_KB", "MGX_TAIL_SMALL_M", "MGX_MID_SINGLE_DOWN", "MGX_MID_SINGLE_UP",
         "MGX_MID_UP_THREADS", "MGX_MID_DOWN_THREADS", "MGX_MID_UP_Q", "MGX_MID_DOWN_Q")


def _problem(M, seed=0):
    from multigrid_amd import rhs
    f = rhs.random_np(M)
    rng = np.random.default_rng(M + seed)
    v0 = np.zeros((M, M), order="F")
    v0[0, :], v0[-1, :], v0[:, 0], v0[:, -1] = rng.standard_normal((4, M))
    v0[1:-1, 1:-1] = rng.standard_normal((M - 2, M - 2)) * 0.01
    return f, v0


def _run(mg, monkeypatch, env, f, v0, dx, Nv, nd, nu, omega, coarsest):
    for k in KNOBS:
        monkeypatch.delenv(k, raising=False)
    for k, val in env.items():
        monkeypatch.setenv(k, str(val))
    v = v0.copy(order="F")
    out = mg.solve_host(f, v, dx, Nv, nd, nu, omega, coarsest)
    for k in KNOBS:
        monkeypatch.delenv(k, raising=False)
    return v, out.ResidualNorms


SWEEPS = [(2, 2), (1, 1), (0, 1), (1, 0), (0, 0), (3, 1), (1, 3), (4, 0), (0, 4), (2, 1), (1, 2), (2, 0), (0, 2)]


@pytest.mark.parametrize("nd,nu", SWEEPS)
@pytest.mark.parametrize("M,coarsest", [(513, 2), (1025, 2), (705, 11), (1153, 9)])
def test_default_groups_match_the_oracle(mg, O, monkeypatch, M, coarsest, nd, nu):
    f, v0 = _problem(M)
    dx = 1.0 / (M - 1)
    omega = 0.6 if (nd + nu) % 2 == 0 else 0.85
    ref_sol, ref_rn, _ = O.multigrid(f, v0, dx, 3, nd, nu, omega, coarsest, omp=M > 1025)
    sol, rn = _run(mg, monkeypatch, {}, f, v0, dx, 3, nd, nu, omega, coarsest)
    assert_bits(rn, ref_rn, "ResNorms")
    assert_bits(sol, ref_sol, "Sol")


@pytest.mark.parametrize("env", [
    {"MGX_MID_DOWN_NL": 3, "MGX_MID_UP_NL": 2},
    {"MGX_MID_DOWN_NL": 2, "MGX_MID_UP_NL": 3},
    {"MGX_MID_DOWN_NL": 3, "MGX_MID_UP_NL": 5, "MGX_MID_MAX_M": 4097},
    {"MGX_MID_DOWN_NL": 4, "MGX_MID_UP_NL": 5, "MGX_NO_TAIL": 1, "MGX_MID_LDS_KB": 160},
    {"MGX_NO_TAIL": 1},
    {"MGX_TAIL_SMALL_M": 65},
    {"MGX_MID_SINGLE_DOWN": 0, "MGX_MID_SINGLE_UP": 0},
    {"MGX_MID_SINGLE_DOWN": 2049, "MGX_MID_SINGLE_UP": 2049, "MGX_MID_UP_THREADS": 128, "MGX_MID_UP_Q": 4,
     "MGX_MID_DOWN_THREADS": 256, "MGX_MID_DOWN_Q": 4},
    {"MGX_MID_UP_THREADS": 1024, "MGX_MID_UP_Q": 1, "MGX_MID_DOWN_THREADS": 512, "MGX_MID_DOWN_Q": 1},
    {"MGX_MID_UP_THREADS": 320, "MGX_MID_UP_Q": 4, "MGX_MID_DOWN_THREADS": 192, "MGX_MID_DOWN_Q": 2, "MGX_MID_MAX_M": 1025},
    {"MGX_MID_DOWN_TI": 3, "MGX_MID_DOWN_TJ": 5, "MGX_MID_UP_TI": 7, "MGX_MID_UP_TJ": 3},
    {"MGX_MID_DOWN_TI": 1, "MGX_MID_DOWN_TJ": 1, "MGX_MID_UP_TI": 1, "MGX_MID_UP_TJ": 1, "MGX_MID_MAX_M": 257},
    {"MGX_MID_DOWN_TI": 9, "MGX_MID_DOWN_TJ": 2, "MGX_MID_UP_TI": 50, "MGX_MID_UP_TJ": 11, "MGX_MID_MAX_M": 2049},
])
@pytest.mark.parametrize("nd,nu", [(2, 2), (1, 1), (0, 3), (3, 0), (1, 2)])
def test_every_grouping_and_tile_size_gives_the_same_bits(mg, O, monkeypatch, env, nd, nu):
    M = 4097
    f, v0 = _problem(M, 1)
    dx = 1.0 / (M - 1)
    ref_sol, ref_rn, _ = O.multigrid(f, v0, dx, 2, nd, nu, 0.7, 2, omp=True)
    off_sol, off_rn = _run(mg, monkeypatch, {"MGX_NO_MID": 1}, f, v0, dx, 2, nd, nu, 0.7, 2)
    assert_bits(off_rn, ref_rn, "ResNorms, grouping off")
    assert_bits(off_sol, ref_sol, "Sol, grouping off")
    sol, rn = _run(mg, monkeypatch, env, f, v0, dx, 2, nd, nu, 0.7, 2)
    assert_bits(rn, ref_rn, "ResNorms %r" % (env,))
    assert_bits(sol, ref_sol, "Sol %r" % (env,))


@pytest.mark.parametrize("M,coarsest", [(1537, 3), (897, 7), (2305, 9), (641, 5), (769, 6)])
def test_sizes_whose_levels_are_not_powers_of_two(mg, O, monkeypatch, M, coarsest):
    """k*2^d+1 hierarchies: ragged tiles on every level, an even coarsest level for some."""
    f, v0 = _problem(M, 2)
    dx = 0.37
    ref_sol, ref_rn, _ = O.multigrid(f, v0, dx, 3, 2, 2, 0.6, coarsest, omp=True)
    for env in ({}, {"MGX_NO_TAIL": 1, "MGX_MID_UP_NL": 5, "MGX_MID_DOWN_NL": 3}):
        sol, rn = _run(mg, monkeypatch, env, f, v0, dx, 3, 2, 2, 0.6, coarsest)
        assert_bits(rn, ref_rn, "ResNorms %r" % (env,))
        assert_bits(sol, ref_sol, "Sol %r" % (env,))


def test_non_finite_right_hand_side(mg, O, monkeypatch):
    """NaN and infinity travel through the grouped launches as through the reference's operators."""
    M = 1025
    f, v0 = _problem(M, 3)
    f[300, 411] = np.nan
    f[700, 95] = np.inf
    dx = 1.0 / (M - 1)
    ref_sol, ref_rn, _ = O.multigrid(f, v0, dx, 2, 2, 2, 0.6, 2)
    sol, rn = _run(mg, monkeypatch, {}, f, v0, dx, 2, 2, 2, 0.6, 2)
    assert_bits(rn, ref_rn, "ResNorms")
    # NaN payloads are not part of the contract: same NaN positions, same bits elsewhere
    assert np.isnan(ref_sol).sum() > 1000
    assert np.array_equal(np.isnan(sol), np.isnan(ref_sol))
    ok = ~np.isnan(ref_sol)
    assert np.array_equal(sol[ok].view(np.uint64), ref_sol[ok].view(np.uint64))


@pytest.mark.parametrize("threads", [64, 256, 1024])
def test_tail_workgroup_sizes_give_the_same_bits(threads):
    """The one-workgroup tail has an instance for levels that one pass of the workgroup covers
    (the default 1024 threads and a top level of at most 33 x 33), one whose top level takes
    several passes, and a general one (fewer threads, MGX_TAIL_THREADS): all three must give
    the oracle's bits.  The thread count is read once per process, hence the child process
    (a short run of the randomised parity sweep over the schedules that use the tail)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MGX_TAIL_THREADS=str(threads), FUZZ_LEAN="1", FUZZ_CASES="150", FUZZ_MODES="2,6,7,8,9")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "60", str(threads), "600"],
                         env=env, cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "bit-identical to the oracle" in out.stdout
