import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def coracle():
    from oracle.c_oracle import COracle
    return COracle()


@pytest.fixture(scope='session')
def ctx():
    """One HIP context for the whole GPU session (fails loudly when the library is missing)."""
    from nexoclom_amd import hip_api
    if hip_api.device_count() < 1:
        pytest.fail('GPU test selected but no HIP device is visible')
    c = hip_api.Context(0)
    yield c
    c.close()
