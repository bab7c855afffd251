import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_ctx():
    """One engine context on GPU 0; fails loudly (no CPU fallback) when there is no GPU."""
    from rpvg_amd import hip
    ctx = hip.Context(0)
    yield ctx
    ctx.close()
