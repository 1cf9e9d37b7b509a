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
    """The CPU oracle (checker only)."""
    from oracle import c_oracle
    c_oracle.lib()
    return c_oracle


@pytest.fixture(scope="session")
def gpu_ctx():
    """One fec_ctx on cuda:0 through the C ABI.  Fails loudly if the HIP library or GPU is missing."""
    import forge_ec_amd as F
    ctx = F.Context(0)
    yield ctx
    ctx.close()
