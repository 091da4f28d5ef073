import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def R():
    """The CPU oracle (oracle/fdm_ref_py.py) — the checker, never the product."""
    import fdm_ref_py
    fdm_ref_py.load()
    return fdm_ref_py


@pytest.fixture(scope="session")
def gpu():
    """fastdem_amd with a usable device; loading fails loudly if the HIP library is missing."""
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    import fastdem_amd
    fastdem_amd.capi.load()
    return fastdem_amd
