import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def prt_lib():
    """libprt_hip.so, built in-tree if stale (hipcc cross-compiles without a GPU)."""
    from pooraytracer_amd import api, build
    build.build()
    return api.load()


@pytest.fixture(scope="session")
def gpu(prt_lib):
    from pooraytracer_amd import api
    if api.device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests need a real GPU (there is no CPU fallback)")
    return 0


@pytest.fixture
def dev_lib(gpu):
    """Scenes created inside this test come from libprt_hip_dev.so: the -DPRT_DEV_HOOKS=1 build of the same sources, the
    only one that reads the PRT_TUNE_* / PRT_TEST_* environment hooks (failure injection, forced layouts)."""
    from pooraytracer_amd import api
    with api.dev_hooks() as L:
        yield L
