import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run by the driver with -m gpu)')


@pytest.fixture(scope='session')
def pkg():
    return importlib.import_module('matlab-code_amd')


@pytest.fixture(scope='session')
def eng(pkg):
    """One engine per test session; creating it fails loudly without a GPU / built library."""
    e = pkg.Engine(0)
    yield e
    e.close()


@pytest.fixture
def tensor_passes():
    """Solver MTTKRPs through the tensor-pass kernels (partial contraction + reduction, dimension-tree cache) even for
    blocks small enough for the one-launch kernel (contract.hip small_mttkrp_k); the library reads the switch per call."""
    os.environ['AOADMM_NO_SMALL_MTTKRP'] = '1'
    yield
    os.environ.pop('AOADMM_NO_SMALL_MTTKRP', None)
