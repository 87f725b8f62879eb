"""A fixed slice of the randomised parity sweeps (tools/fuzz_solver.py, tools/fuzz_coupled.py) in the GPU suite:
randomly drawn shapes, ranks, constraint cells over the whole catalogue, masks, precisions, inner iteration counts, the
coupled and PARAFAC2 families of the example scripts -- each case through both MTTKRP paths against the oracle."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, 'tools', name + '.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize('case', range(500, 524))
def test_random_cp_model(eng, case):
    assert _tool('fuzz_solver').one_case(eng, case) == []


@pytest.mark.parametrize('case', range(500, 518))
def test_random_coupled_or_parafac2_model(eng, case):
    assert _tool('fuzz_coupled').one_case(eng, case) == []
