import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(HERE, "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


@pytest.fixture(params=["fused", "tile", "layers"])
def mbgd_path(request, monkeypatch):
    """Runs a test once per implementation of the minibatch step: the fused per-sample kernel, the
    sample-tile kernel and the layer kernels (the library picks by batch size and shape unless
    CCNN_MBGD_PATH says otherwise; "tile" falls back to the layer kernels for shapes whose tile
    does not fit the LDS)."""
    monkeypatch.setenv("CCNN_MBGD_PATH", request.param)
    return request.param
