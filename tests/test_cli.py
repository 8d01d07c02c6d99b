"""The command-line driver multigrid_amd/mgx_multigrid: the reference's process
contract (main.cpp:360-459): flags, Input.mat in, Output.mat out, error
messages.  Argument and file validation run on the CPU; the end-to-end solves
are GPU tests compared bit for bit with the oracle."""
import os
import subprocess

import numpy as np
import pytest

import mat4
from conftest import ROOT, assert_bits

BIN = os.path.join(ROOT, "multigrid_amd", "mgx_multigrid")


def run(args, cwd):
    p = subprocess.run([BIN] + [str(a) for a in args], cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=300)
    return p.returncode, p.stdout


def test_help_lists_the_reference_flags(tmp_path):
    rc, out = run(["--help"], tmp_path)
    assert rc == 0
    for flag in ("--Nv", "--Nbefore", "--Nafter", "--omega", "--coarsest", "-v", "-b", "-a", "-o", "-c"):
        assert flag in out
    assert "(=100)" in out and "(=3)" in out and "(=0.6)" in out and "(=2)" in out  # main.cpp:366-370
    assert run(["-h"], tmp_path)[0] == 0


def test_bad_command_lines(tmp_path):
    assert run(["--nope"], tmp_path)[0] == 2
    assert run(["-v"], tmp_path)[0] == 2
    assert run(["-v", "abc"], tmp_path)[0] == 2
    assert run(["3", "4"], tmp_path)[0] == 2  # only Nv is positional (main.cpp:373-374)


def test_input_validation_matches_reference_messages(tmp_path):
    rc, out = run(["-v", "1"], tmp_path)
    assert rc == 1 and "cannot open Input.mat" in out
    mat4.write(tmp_path / "Input.mat", [("f", np.zeros((5, 7)))])
    rc, out = run(["-v", "1"], tmp_path)
    assert rc == 1 and "Error: Input is not square matrix!" in out           # main.cpp:410
    mat4.write(tmp_path / "Input.mat", [("f", np.zeros((6, 6)))])
    rc, out = run(["-v", "1"], tmp_path)
    assert rc == 1 and "Error: Input has even rows and columns!" in out      # main.cpp:415
    mat4.write(tmp_path / "Input.mat", [("g", np.zeros((5, 5)))])
    rc, out = run(["-v", "1"], tmp_path)
    assert rc == 1 and "no variable f" in out
    (tmp_path / "Input.mat").write_bytes(b"this is not a mat file, not at all.....")
    rc, out = run(["-v", "1"], tmp_path)
    assert rc == 1 and "Invalid file format" in out


def test_mat4_helper_round_trip(tmp_path):
    a = np.arange(12.0).reshape(3, 4)
    mat4.write(tmp_path / "x.mat", [("Seq_f", np.ones((1, 3))), ("f", a), ("f_loc", np.array([[0.0, 0.0, 0.5, 0.5]]))])
    got = mat4.read(tmp_path / "x.mat")
    assert np.array_equal(got["f"], a) and got["f_loc"].ravel().tolist() == [0.0, 0.0, 0.5, 0.5]
    mat4.write(tmp_path / "y.mat", [("f", a)], big_endian=True)
    assert np.array_equal(mat4.read(tmp_path / "y.mat")["f"], a)


@pytest.mark.gpu
@pytest.mark.parametrize("flags,nv,nd,nu,omega", [
    (["-v", "6", "-b", "2", "-a", "2"], 6, 2, 2, 0.6),
    (["4"], 4, 3, 3, 0.6),                                         # positional Nv, default sweeps (main.cpp:366-368)
    (["--Nv=3", "--Nbefore", "1", "--Nafter=2", "-o", "0.8"], 3, 1, 2, 0.8),
    (["-v5", "-b", "7", "-a", "7", "--stepwise"], 5, 7, 7, 0.6),
])
def test_end_to_end_matches_oracle(O, tmp_path, flags, nv, nd, nu, omega):
    from multigrid_amd import rhs
    M = 129
    h = 1.0 / (M - 1)
    f = rhs.random_np(M)
    # MGTester.m:146 writes Seq_f, f and f_loc = [x0 y0 dx dy]
    mat4.write(tmp_path / "Input.mat", [("Seq_f", np.zeros((1, 2))), ("f", f), ("f_loc", np.array([[0.0, 0.0, h, h]]))])
    rc, out = run(flags, tmp_path)
    assert rc == 0, out
    res = mat4.read(tmp_path / "Output.mat")
    assert set(res) == {"Sol", "ResNorms", "Times"}                                   # main.cpp:453-455
    sol, rn, _ = O.multigrid(f, np.zeros((M, M), order="F"), h, nv, nd, nu, omega, 2)
    assert res["ResNorms"].shape == (nv + 1, 1) and res["Times"].shape == (nv + 1, 1)
    assert_bits(res["ResNorms"].ravel(), rn, "ResNorms")
    assert_bits(res["Sol"], sol, "Sol")
    assert res["Times"][0, 0] == 0 and np.all(np.diff(res["Times"].ravel()) >= 0)


@pytest.mark.gpu
def test_default_grid_without_f_loc_and_big_endian_float_input(O, tmp_path):
    """No f_loc: unit spacing (DTSource/DTMesh2D.cpp:466-480); the file may be big-endian."""
    from multigrid_amd import rhs
    M = 65
    f = rhs.smooth_np(M)
    mat4.write(tmp_path / "Input.mat", [("f", f)], big_endian=True)
    rc, out = run(["-v", "3", "-b", "2", "-a", "2", "--output", "Out2.mat"], tmp_path)
    assert rc == 0, out
    res = mat4.read(tmp_path / "Out2.mat")
    sol, rn, _ = O.multigrid(f, np.zeros((M, M), order="F"), 1.0, 3, 2, 2, 0.6, 2)
    assert_bits(res["ResNorms"].ravel(), rn, "ResNorms")
    assert_bits(res["Sol"], sol, "Sol")


@pytest.mark.gpu
def test_unsupported_coarsest_prints_the_reference_message(tmp_path):
    from multigrid_amd import rhs
    M = 65
    mat4.write(tmp_path / "Input.mat", [("f", rhs.smooth_np(M)), ("f_loc", np.array([[0, 0, 1 / 64, 1 / 64]]))])
    rc, out = run(["-v", "2", "-c", "4"], tmp_path)
    assert rc == 0
    assert "Error: Direct solver not defined for 5x5 matrices!" in out               # main.cpp:167


@pytest.mark.gpu
def test_exact_coarse_flag(O, tmp_path):
    """--exact-coarse (an extension): -c 4 ends on a 5x5 grid, which is then solved exactly."""
    from multigrid_amd import rhs
    M = 129
    h = 1.0 / (M - 1)
    f = rhs.random_np(M)
    mat4.write(tmp_path / "Input.mat", [("f", f), ("f_loc", np.array([[0.0, 0.0, h, h]]))])
    rc, out = run(["-v", "5", "-b", "2", "-a", "2", "-c", "4", "--exact-coarse"], tmp_path)
    assert rc == 0 and "Direct solver not defined" not in out, out
    res = mat4.read(tmp_path / "Output.mat")
    sol, rn, _ = O.multigrid(f, np.zeros((M, M), order="F"), h, 5, 2, 2, 0.6, 4, coarse_exact=True)
    assert np.allclose(res["ResNorms"].ravel(), rn, rtol=1e-6, atol=1e-9 * rn[0])
    assert np.abs(res["Sol"] - sol).max() <= 1e-9 * np.abs(sol).max()


def test_hostile_dimensions_in_the_file_are_refused(tmp_path):
    """A MAT-v4 header whose dimensions are negative or promise more data than the file holds
    must not reach an allocation or a wrapped size: the variable is not accepted."""
    import struct
    for mrows, ncols in ((-5, 5), (5, -5), (2 ** 31 - 1, 2 ** 31 - 1), (65536, 65536), (5, 5)):
        body = struct.pack("<5i", 0, mrows, ncols, 0, 2) + b"f\0" + b"\0" * 64  # 8 doubles of data at most
        (tmp_path / "Input.mat").write_bytes(body)
        rc, out = run(["-v", "1"], tmp_path)
        assert rc == 1 and ("do not fit the file" in out or "no variable f" in out), (mrows, ncols, out)


@pytest.mark.gpu
def test_the_command_line_mgtester_runs(O, tmp_path):
    """MGTester.m:150-153 shells out to `build/Release/multigrid -v 50 -b Nb -a Na -o omega` from
    the directory that holds Input.mat.  `make mgtester` puts the driver at that path; the
    tester's command line, character for character, gives the oracle's Output.mat."""
    from multigrid_amd import rhs
    exe = os.path.join(ROOT, "build", "Release", "multigrid")
    if not os.path.exists(exe):
        pytest.skip("build/Release/multigrid not built (make -C multigrid_amd/csrc mgtester)")
    (tmp_path / "build" / "Release").mkdir(parents=True)
    os.symlink(exe, tmp_path / "build" / "Release" / "multigrid")
    M = 129
    h = 1.0 / (M - 1)
    f = rhs.smooth_np(M)                                          # MGTester.m:136-141, libm-free variant
    mat4.write(tmp_path / "Input.mat", [("Seq_f", np.zeros((1, 2))), ("f", f), ("f_loc", np.array([[0.0, 0.0, h, h]]))])
    cmd = "build/Release/multigrid -v 50 -b 3 -a 3 -o 0.8"        # runMG(3, 3, 0.8)
    p = subprocess.run(cmd, shell=True, cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert p.returncode == 0, p.stdout
    got = mat4.read(tmp_path / "Output.mat")
    sol, rn, _ = O.multigrid(f, np.zeros((M, M), order="F"), h, 50, 3, 3, 0.8, 2)
    assert_bits(got["ResNorms"].ravel(), rn, "ResNorms")
    assert_bits(got["Sol"], sol, "Sol")
    assert got["Times"].shape[0] == 51
