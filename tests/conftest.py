import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The C restatement (checker).  Built on demand from oracle/mmdx_oracle.c."""
    from oracle.pyoracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def hip_lib():
    """The product library; building it needs hipcc only (cross-compiles without a GPU)."""
    from simple_mmd_renderer_amd import build, _capi
    build.build()
    return _capi.lib()
