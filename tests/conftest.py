import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import grhip_loader  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def po():
    """the TEST-ONLY CPU checker"""
    p = grhip_loader.import_oracle()
    if not p.have_oracle():
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
        import importlib
        p = importlib.reload(p)
    return p


@pytest.fixture(scope="session")
def g():
    return grhip_loader.import_grhip()


@pytest.fixture(scope="session")
def gpu(g):
    """the product library on a real device; fails (does not skip) when the
    library is missing or no GPU is visible, so a silent fallback cannot pass"""
    g.lib()
    n = g.device_count()
    assert n >= 1, "no HIP device visible: -m gpu tests must run on the GPU box"
    return g


@pytest.fixture(scope="session")
def wl(g):
    return g.workload


def rel_err_max(a, ref):
    """||a - ref||_inf / ||ref||_inf  """
    a = np.asarray(a); ref = np.asarray(ref)
    s = float(np.abs(ref).max()) if ref.size else 0.0
    if s == 0.0:
        return float(np.abs(a).max()) if a.size else 0.0
    return float(np.abs(a - ref).max()) / s


def bits_equal(a, b):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    return bool(np.array_equal(a.view(np.uint8), b.view(np.uint8)))


from parity_util import demod_close, demod_report  # noqa: E402,F401  (shared with __graft_entry__.smoke)
