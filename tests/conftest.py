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


@pytest.fixture(scope="session", params=["default", "tiled_all"])
def gpu(request):
    """fastdem_amd with a usable device; loading fails loudly if the HIP library is missing.
    Every GPU test runs twice: with the engine's own choice of pipeline by scan size, and with every
    scan pushed through the large-scan pipelines (per-tile record pools: `tiled_min` = 1; raycasting with the ray queue
    ordered by sector and the walk on an LDS window, fdm_raywedge.hpp: `ray_large_min` = 1) — ENGINE options that
    every Engine the tests construct receives through `Engine.default_options`; nothing process-wide, nothing the
    product reads from the environment."""
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    import fastdem_amd
    fastdem_amd.capi.load()
    saved = dict(fastdem_amd.Engine.default_options)
    fastdem_amd.Engine.default_options = {"tiled_min": 1, "ray_large_min": 1} if request.param == "tiled_all" else {}
    # (test campaigns of a switch that is off by default, e.g. FDM_TEST_EXTRA_OPTIONS="ray_overlap=1,voxel_small=0": read by
    #  the TEST harness, handed to the engines as options like the variant's own)
    for kv in filter(None, os.environ.get("FDM_TEST_EXTRA_OPTIONS", "").split(",")):
        fastdem_amd.Engine.default_options[kv.split("=")[0]] = int(kv.split("=")[1])
    yield fastdem_amd
    fastdem_amd.Engine.default_options = saved
