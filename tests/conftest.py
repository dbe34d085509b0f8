import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


@pytest.fixture(scope="session")
def sw():
    import stringwars_amd
    return stringwars_amd


@pytest.fixture(scope="session")
def orc():
    import oracle
    return oracle


@pytest.fixture(scope="session")
def scope(sw):
    """One GPU scope for the whole session; a missing device is a hard failure under `-m gpu`."""
    return sw.DeviceScope(gpu_device=0)
