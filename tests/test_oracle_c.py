"""The C/OpenMP CPU restatement (oracle/c/aoadmm_cpu.c, bench.py's cpu_baseline) against the numpy oracle: same
model, same init, fixed iteration counts.  Both are test infrastructure; this pins the compiled port to the
restatement the GPU parity tests use."""
import copy

import numpy as np
import pytest

from oracle import aoadmm as OA
from oracle import c_port
from oracle.tensor_ops import mttkrp as o_mttkrp
from helpers import cp_model, options, rel_fro


@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_c_mttkrp_matches_numpy(dtype):
    rng = np.random.default_rng(3)
    X = rng.standard_normal((13, 9, 11)).astype(dtype)
    U = [rng.standard_normal((n, 5)) for n in X.shape]
    for mode in range(3):
        ref = o_mttkrp(np.asarray(X, dtype=np.float64), U, mode)
        assert rel_fro(c_port.mttkrp(X, U, mode), ref) < 1e-12


@pytest.mark.parametrize('cons', [
    [('TV regularization', 0.001), ('non-negativity',), ('non-negativity',)],     # config 5
    [('non-negativity',)] * 3,                                                    # config 2
    [None, ('non-negativity',), ('TV regularization', 0.01)],
])
def test_c_solver_matches_numpy_oracle(cons):
    rng = np.random.default_rng(5)
    Z, io, _ = cp_model((31, 23, 27), 4, rng, cons)
    G = OA.init_coupled_AOADMM_CMTF({**Z, 'prox_operators': None}, io, rng=np.random.default_rng(6))
    opt = options(MaxOuterIters=5, MaxInnerIters=4)
    _, Fo, _, oo = OA.cmtf_AOADMM(Z, alg_options=opt, init=copy.deepcopy(G))
    F, Zs, Ms, ft, inner = c_port.solve_cp3(Z['object'][0], cons, G['fac'], G['constraint_fac'], G['constraint_dual_fac'], 5, 4)
    for a, b in zip(Fo['fac'], F):
        assert rel_fro(b, a) < 1e-10
    for m in range(3):
        if cons[m] is not None:
            assert rel_fro(Zs[m], Fo['constraint_fac'][m]) < 1e-10
            assert rel_fro(Ms[m], Fo['constraint_dual_fac'][m]) < 1e-9
    assert np.array_equal(inner, np.asarray(oo['innerIters'], dtype=np.int32))
    assert np.allclose(ft, oo['func_val_conv'], rtol=1e-9, atol=1e-12)


def test_c_synth_has_the_bench_statistics():
    X, A = c_port.synth(40, 30, 20, 3, noise=0.05, seed=0)
    assert X.dtype == np.float32 and abs(np.linalg.norm(X.astype(np.float64)) - 1.0) < 1e-5
    M = np.einsum('ir,jr,kr->ijk', *A)
    M /= np.linalg.norm(M)
    rel = np.linalg.norm(X / np.linalg.norm(X) - M) / np.linalg.norm(M)
    assert 0.03 < rel < 0.08                      # noise level 0.05 of the model's norm
