import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "montecarlo-risk-engine_amd")
for p in (PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle_backend import OracleBackend
    return OracleBackend()


@pytest.fixture(scope="session")
def hip():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mcx import _native
    return _native.HipBackend()
