import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (test infrastructure only)."""
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def ffm():
    from ffm_import import ffm as pkg
    if not os.path.exists(pkg.libpath()):
        pkg.build()
    return pkg


@pytest.fixture(scope="session")
def ctx(ffm):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    c = ffm.Context(0)
    yield c
    c.close()
