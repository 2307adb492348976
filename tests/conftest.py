import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from tests import _oracle
    return _oracle.load()


@pytest.fixture(scope="session")
def diag():
    """librt_hip_diag.so beside the product library: unit-test entry points (include/rt_hip_diag.h), the wavefront pipeline.
    It recognises the product's material tokens (native._Lazy), so scenes built once serve both libraries."""
    import raytracing_c_amd as rt
    assert rt.diag.rt_init(0) == 0, rt.last_error(rt.diag)
    return rt.diag
