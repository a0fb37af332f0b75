import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "blockcopy-video-processing-pytorch_amd")
for p in (PKG, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__)), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def pytest_collection_modifyitems(config, items):
    """Plain `pytest tests` on a CPU-only host: GPU-tier tests are skipped instead of failing in torch.cuda init."""
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs an MI355X (run with -m gpu on the GPU box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The C-ABI library must exist for every test tier (hipcc cross-compiles without a GPU)."""
    import build as bc_build  # blockcopy-video-processing-pytorch_amd/build.py

    bc_build.build_hip_library()
    import oracle

    oracle.build()
    yield


@pytest.fixture()
def oracle_backend():
    """Inject the CPU-oracle checker backend for host-logic tests; restored afterwards."""
    import blockcopy.backend as bk
    from oracle_backend import OracleBackend

    prev = bk.set_backend(OracleBackend())
    yield
    bk.set_backend(prev)


@pytest.fixture(autouse=True)
def _engine_is_restored():
    """Every test starts (and leaves the next one) on the default engine: a test that selects the reference decomposition and fails --
    or forgets -- must not change what later tests measure."""
    from blockcopy.core import tensorwrapper as tw

    tw.set_engine("fused")
    yield
    tw.set_engine("fused")
