import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ora():
    """the CPU oracle (checker)"""
    return ge.load_oracle()


@pytest.fixture(scope="session")
def pkg():
    return ge.load_package()


@pytest.fixture(scope="session")
def gpu(pkg):
    """(ctx, device) on cuda:0 — fails loudly when the HIP library or the device is missing"""
    import torch
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    ctx = pkg.capi.Ctx(0)
    yield ctx, torch.device("cuda:0")
    ctx.close()


def make_cloud(rng, n, kind="sphere", noise=0.0):
    """small random surfaces with normals for the unit tests"""
    if kind == "sphere":
        p = rng.normal(size=(n, 3)); p /= np.linalg.norm(p, axis=1, keepdims=True)
        nrm = p.copy()
        p = p * (1.0 + noise * rng.normal(size=(n, 1)))
    elif kind == "ellipsoid":
        d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        s = np.array([1.0, 0.6, 0.35])
        p = d * s
        nrm = d / s; nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    elif kind == "plane":
        p = np.concatenate([rng.uniform(-1, 1, size=(n, 2)), noise * rng.normal(size=(n, 1))], axis=1)
        nrm = np.tile([0.0, 0.0, 1.0], (n, 1))
    else:
        raise ValueError(kind)
    return p.astype(np.float32), nrm.astype(np.float32)
