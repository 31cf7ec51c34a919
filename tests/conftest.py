import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as ge
    ge.build()
    return ge.load_package()


@pytest.fixture(scope="session")
def orc():
    import __graft_entry__ as ge
    ge.build_oracle()
    import oracle_binding
    return oracle_binding.load_oracle()


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
