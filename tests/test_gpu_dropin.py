"""The reference-side binding, for real: oracle/_ref/dt_dropin is MultiGrid() with the
signature of main.cpp:283 implemented by ONE call into libmgx's C ABI and compiled against
the reference's own DTSource containers and MAT-file code (oracle/ref_dt_dropin.cpp; built
in the development container where the reference checkout lives, shipped as a binary).
It reads Input.mat and writes Output.mat through DTMatlabDataFile exactly as the reference's
main() does, so this also checks this build's MAT-v4 reader/writer against the
reference's."""
import os
import subprocess

import numpy as np
import pytest

import mat4
from conftest import ROOT, assert_bits

pytestmark = pytest.mark.gpu

BIN = os.path.join(ROOT, "oracle", "_ref", "dt_dropin")
CLI = os.path.join(ROOT, "multigrid_amd", "mgx_multigrid")


def _run(cmd, cwd):
    p = subprocess.run([str(c) for c in cmd], cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       timeout=300)
    return p.returncode, p.stdout


@pytest.mark.parametrize("M,nv,nd,nu,omega,coarsest", [(129, 5, 2, 2, 0.6, 2), (257, 4, 3, 3, 0.6, 2),
                                                       (65, 3, 1, 2, 0.8, 2), (513, 3, 2, 2, 0.6, 2)])
def test_dtsource_binding_matches_oracle(O, tmp_path, M, nv, nd, nu, omega, coarsest):
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/dt_dropin not built (needs the reference checkout at build time)")
    from multigrid_amd import rhs
    h = 1.0 / (M - 1)
    f = rhs.random_np(M)
    mat4.write(tmp_path / "Input.mat", [("f", f), ("f_loc", np.array([[0.0, 0.0, h, h]]))])
    rc, out = _run([BIN, nv, nd, nu, omega, coarsest], tmp_path)
    assert rc == 0, out
    res = mat4.read(tmp_path / "Output.mat")  # written by the reference's DTMatlabDataFile
    sol, rn, _ = O.multigrid(f, np.zeros((M, M), order="F"), h, nv, nd, nu, omega, coarsest)
    assert_bits(res["ResNorms"].ravel(), rn, "ResNorms")
    assert_bits(res["Sol"], sol, "Sol")
    assert res["Times"].ravel()[0] == 0

    # this build's own command-line driver on the same input gives the same file contents
    os.rename(tmp_path / "Output.mat", tmp_path / "Output_dt.mat")
    rc, out = _run([CLI, "-v", nv, "-b", nd, "-a", nu, "-o", omega, "-c", coarsest], tmp_path)
    assert rc == 0, out
    own = mat4.read(tmp_path / "Output.mat")
    assert_bits(own["Sol"], res["Sol"], "Sol, CLI vs DTSource binding")
    assert_bits(own["ResNorms"].ravel(), res["ResNorms"].ravel(), "ResNorms, CLI vs DTSource binding")


def test_dtsource_binding_over_several_slabs(O, tmp_path):
    """The same binding with mgx_options.ngpus = 3: MultiGrid() as main.cpp:445 calls it, over
    three j-slabs (sharing the box's GPU), Output.mat bit-identical to the oracle."""
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/dt_dropin not built (needs the reference checkout at build time)")
    from multigrid_amd import rhs
    M = 1025
    h = 1.0 / (M - 1)
    f = rhs.random_np(M)
    mat4.write(tmp_path / "Input.mat", [("f", f), ("f_loc", np.array([[0.0, 0.0, h, h]]))])
    rc, out = _run([BIN, 4, 2, 2, 0.6, 2, 3, 129], tmp_path)
    assert rc == 0, out
    res = mat4.read(tmp_path / "Output.mat")
    sol, rn, _ = O.multigrid(f, np.zeros((M, M), order="F"), h, 4, 2, 2, 0.6, 2)
    assert_bits(res["ResNorms"].ravel(), rn, "ResNorms")
    assert_bits(res["Sol"], sol, "Sol")
    # this build's command-line driver with --ngpus
    os.rename(tmp_path / "Output.mat", tmp_path / "Output_dt.mat")
    rc, out = _run([CLI, "-v", 4, "-b", 2, "-a", 2, "--ngpus", 3, "--tail-m", 129], tmp_path)
    assert rc == 0, out
    own = mat4.read(tmp_path / "Output.mat")
    assert_bits(own["Sol"], sol, "Sol, CLI --ngpus")
    assert_bits(own["ResNorms"].ravel(), rn, "ResNorms, CLI --ngpus")


def test_dtsource_binding_reports_errors_like_the_reference(tmp_path):
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/dt_dropin not built")
    mat4.write(tmp_path / "Input.mat", [("f", np.zeros((6, 6))), ("f_loc", np.array([[0.0, 0.0, 0.2, 0.2]]))])
    rc, out = _run([BIN, 1, 2, 2, 0.6, 2], tmp_path)
    assert rc == 1 and "Error:" in out and "even" in out  # main.cpp:413-417 exits 1 with a message
