#!/usr/bin/env python3
"""Development timing of BASELINE configs 1-4 on one MI355X next to the numpy oracle (config 5 is bench.py).
Per-iteration time = slope between two solves of different length, so upload/download and first-call costs drop out."""
import copy, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
pkg = importlib.import_module('matlab-code_amd')
from oracle import aoadmm as OA
from helpers import options, script1_model, script3_model, script4_model


def cfg2_model(rng):
    n, R = 500, 10
    A = [rng.random((n, R)) for _ in range(3)]
    X = np.einsum('ir,jr,kr->ijk', *A, optimize=True)
    X += 0.05 * np.linalg.norm(X) / np.sqrt(X.size) * rng.standard_normal(X.shape)
    X /= np.linalg.norm(X)
    Z = dict(loss_function=['Frobenius'], model=['CP'], modes=[[1, 2, 3]], size=[n, n, n],
             coupling=dict(lin_coupled_modes=[0, 0, 0], coupling_type=[], coupl_trafo_matrices=[None] * 3),
             constrained_modes=[1, 1, 1], constraints=[('non-negativity',)] * 3, weights=[1.0],
             object=[np.asfortranarray(X)])
    io = dict(lambdas_init=[[1] * R], nvecs=0, distr=[lambda a, b: rng.random((a, b))] * 3, normalize=1)
    return Z, io


def slope(eng, Z, G, n1, n2, precision='f64'):
    t = {}
    for n in (n1, n1, n2, n1, n2, n1, n2):      # the first solve also pays one-time allocations; fastest of three each
        t0 = time.perf_counter()
        pkg.cmtf_AOADMM(Z, alg_options=options(MaxOuterIters=n), init=copy.deepcopy(G), engine=eng, precision=precision)
        dt = time.perf_counter() - t0
        t[n] = min(t.get(n, dt), dt)
    return (t[n2] - t[n1]) / (n2 - n1) * 1e3, t[n2]


def oracle_ms(Z, G, n):
    t0 = time.perf_counter()
    OA.cmtf_AOADMM(Z, alg_options=options(MaxOuterIters=n), init=copy.deepcopy(G))
    return (time.perf_counter() - t0) / n * 1e3


def main():
    eng = pkg.Engine(0)
    rows = []
    cases = [('cfg1 script1 CP 40x50x60 + PARAFAC2 (K=20), coupled, nonneg', lambda r: script1_model(r, dims=(40, 50, 60)), 50, 250, 5),
             ('cfg2 CP 500^3 R=10 nonneg, fp64 tensor', cfg2_model, 5, 25, 1),
             ('cfg3 script3 matrix + CP, partial coupling (type 4)', script3_model, 50, 250, 20),
             ('cfg4 PARAFAC2 K=256 slabs, I=40, J_k 61..120, R=3', lambda r: script4_model(r, K=256), 50, 250, 3)]
    for name, make, n1, n2, no in cases:
        rng = np.random.default_rng(4)
        Z, io = make(rng)
        G = OA.init_coupled_AOADMM_CMTF({**Z, 'prox_operators': None}, io, rng=np.random.default_rng(7))
        ms, whole = slope(eng, Z, G, n1, n2)
        oms = oracle_ms(Z, G, no)
        rows.append((name, ms, oms))
        print('%-62s %9.3f ms/iter on the GPU (%d iterations incl. transfers: %.3f s)   oracle %9.1f ms/iter' % (name, ms, n2, whole, oms), flush=True)
    eng.close()


if __name__ == '__main__':
    main()
