#!/usr/bin/env python3
"""Randomised solver parity sweep (development tool, GPU): random shapes, ranks, constraint cells per mode, optional
missing entries and precision; every case runs the HIP solver and the oracle from the same init with tolerances 0 and
compares factors.  usage: fuzz_solver.py [ncases] [seed0]"""
import copy, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
pkg = importlib.import_module('matlab-code_amd')
from oracle import aoadmm as OA
from helpers import cp_model, cp_cp_exact_model, script3_model, options, rel_fro

CELLS = [None, ('non-negativity',), ('box', 0.0, 0.7), ('simplex column-wise', 1.0), ('simplex row-wise', 1.0),
         ('non-decreasing',), ('non-increasing',), ('unimodality', True), ('unimodality', False), ('l1-ball', 2.0),
         ('l2-ball', 1.0), ('non-negative l2-ball', 1.0), ('non-negative l2-sphere', 1.0), ('orthonormal',),
         ('l1 regularization', 0.01), ('l0 regularization', 0.001), ('l2 regularization', 0.01), ('ridge', 0.05),
         ('GL smoothness', 0.1), ('TV regularization', 0.005)]

def one_case(eng, case):
    """Runs one randomly drawn model on both MTTKRP paths; returns the list of mismatch reports (empty = parity)."""
    fails = []
    rng = np.random.default_rng(10_000 + case)
    nd = 3 if rng.random() < 0.8 else 2
    dims = tuple(int(rng.integers(6, 90)) for _ in range(nd))
    R = int(rng.integers(1, 7))
    cons = []
    for d in dims:
        c = CELLS[int(rng.integers(0, len(CELLS)))]
        if c is not None and c[0] == 'orthonormal' and d < R:
            c = ('non-negativity',)
        cons.append(c)
    prec = 'f32' if rng.random() < 0.25 else 'f64'
    Z, io, _ = cp_model(dims, R, rng, cons)
    if rng.random() < 0.2:
        mask = rng.random(dims) > 0.15
        X = np.array(Z['object'][0], dtype=float)
        X[~mask] = 0.0
        Z['object'] = [X]
        Z['miss'] = [mask]
    inner = int(rng.integers(1, 8))
    opt = options(MaxOuterIters=int(rng.integers(2, 7)), MaxInnerIters=inner)
    for path in ('one-launch', 'tensor-pass'):
        if path == 'tensor-pass':
            os.environ['AOADMM_NO_SMALL_MTTKRP'] = '1'
        else:
            os.environ.pop('AOADMM_NO_SMALL_MTTKRP', None)
        try:
            G = OA.init_coupled_AOADMM_CMTF({**Z, 'prox_operators': None}, io, rng=np.random.default_rng(case))
            _, Fo, _, oo = OA.cmtf_AOADMM(Z, alg_options=opt, init=copy.deepcopy(G))
            _, Fg, _, og = pkg.cmtf_AOADMM(Z, alg_options=opt, init=copy.deepcopy(G), engine=eng, precision=prec)
            tol = 2e-4 if prec == 'f32' else 1e-7
            errs = [rel_fro(b, a) for a, b in zip(Fo['fac'], Fg['fac'])]
            ok = all(e < tol for e in errs)
            # objective values too (they carry the regulariser values of constraints_to_prox.m:50-81)
            if 'miss' not in Z:
                ok = ok and np.allclose(og['func_val_conv'], oo['func_val_conv'], rtol=1e-3 if prec == 'f32' else 1e-6, atol=1e-8)
                if not ok:
                    errs = errs + ['f: %s vs %s' % (np.asarray(og['func_val_conv'])[-2:], np.asarray(oo['func_val_conv'])[-2:])]
        except Exception as e:
            ok = False
            errs = [repr(e)[:200]]
        if not ok:
            fails.append(('CASE', case, path, dims, 'R', R, cons, prec, 'miss' if 'miss' in Z else '', 'inner', inner, errs))
    os.environ.pop('AOADMM_NO_SMALL_MTTKRP', None)
    return fails


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    eng = pkg.Engine(0)
    bad = 0
    for case in range(s0, s0 + n):
        for f in one_case(eng, case):
            bad += 1
            print(*f, flush=True)
        if case % 20 == 19:
            print('... %d cases, %d bad' % (case - s0 + 1, bad), flush=True)
    os.environ.pop('AOADMM_NO_SMALL_MTTKRP', None)
    print('cases', n, 'bad', bad)
    eng.close()

if __name__ == '__main__':
    main()
