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
