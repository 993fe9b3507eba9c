import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "lis-slam_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_ctypes as oc
    oc.build()
    return oc


@pytest.fixture(scope="session")
def gpu_ctx():
    import lisreg
    ctx = lisreg.Context(0)          # raises if the HIP library is missing or no device: no silent fallback
    yield ctx
    ctx.close()
