import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_oracle_built():
    so = os.path.join(ROOT, "oracle", "libmg_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libmg_oracle.so"])


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (oracle/mg_oracle.c through ctypes)."""
    _ensure_oracle_built()
    from oracle import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def known():
    with open(os.path.join(GOLDEN, "reference_known_answers.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def f_smooth_257():
    return np.load(os.path.join(GOLDEN, "f_smooth_257.npy"))


@pytest.fixture(scope="session")
def vectors33():
    return np.load(os.path.join(GOLDEN, "oracle_vectors_33.npz"))


@pytest.fixture(scope="session")
def mg():
    """The product package; GPU tests only."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("a test marked gpu ran without a GPU")
    import multigrid_amd
    multigrid_amd._lib.load()
    return multigrid_amd


def bits_equal(a, b):
    """Exact equality of float64 arrays, bit for bit (NaN == NaN, -0.0 != +0.0)."""
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    b = np.ascontiguousarray(np.asarray(b, dtype=np.float64))
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


def assert_bits(a, b, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, a.shape, b.shape)
    if not bits_equal(a, b):
        diff = np.argwhere(np.ascontiguousarray(a).view(np.uint64) != np.ascontiguousarray(b).view(np.uint64))
        k = tuple(diff[0])
        raise AssertionError("%s: %d of %d entries differ; first at %s: %r vs %r"
                             % (what, len(diff), a.size, k, a[k], b[k]))
