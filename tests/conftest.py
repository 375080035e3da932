import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def checker():
    """The CPU checker: the reference's own gssw.c when oracle/_ref is present, else the plain-C port."""
    from oracle import oracle as orc
    if orc.have_ref():
        return orc.RefOracle()
    if not orc.have_port():
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "port"], check=True)
    return orc.PortOracle()


@pytest.fixture(scope="session")
def gpu_ctx():
    from paragraph_amd import capi
    ctx = capi.Context(0)
    yield ctx
    ctx.close()
