#!/usr/bin/env python3
"""Randomised sweep of the TV prox kernel against the oracle (sequential Condat scan): column lengths across the three
memory forms (LDS up to 6400 rows, hybrid to 12000, workspace beyond), data kinds, eta / rho.
usage: fuzz_tv.py <cases> <seed>"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
pkg = importlib.import_module('matlab-code_amd')
from oracle import prox as OP
from helpers import rel_fro
n_cases, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
bad = 0
with pkg.Engine(0) as eng:
    for case in range(n_cases):
        kind = rng.integers(0, 6)
        rows = int(rng.choice([rng.integers(2, 300), rng.integers(300, 1100), rng.integers(1024, 4200), rng.integers(4000, 6500),
                               rng.integers(6390, 6410), rng.integers(6400, 12100), rng.integers(12000, 16000)]))
        R = int(rng.integers(1, 5))
        eta = float(10 ** rng.uniform(-4, 2))
        rho = float(10 ** rng.uniform(-1, 1))
        if kind == 0:
            X = rng.standard_normal((rows, R))
        elif kind == 1:
            seg = int(rng.integers(2, 400))
            X = np.repeat(rng.standard_normal((rows // seg + 1, R)), seg, axis=0)[:rows] + 10 ** rng.uniform(-3, -0.5) * rng.standard_normal((rows, R))
        elif kind == 2:
            X = np.full((rows, R), rng.standard_normal())
        elif kind == 3:
            X = np.linspace(-1, 1, rows)[:, None] * rng.standard_normal((1, R))
        elif kind == 4:
            X = np.round(rng.standard_normal((rows, R)) * 3)              # exact ties
        else:
            X = np.cumsum(rng.standard_normal((rows, R)), axis=0) / np.sqrt(rows)
        c = ('TV regularization', eta)
        ops, _ = OP.constraints_to_prox([1], [c], [rows])
        ref = ops[0](X, rho)
        got = eng.prox(c, X, rho)
        ok = rel_fro(got, ref) < 1e-10 or np.max(np.abs(got - ref)) < 1e-12
        if not ok:
            bad += 1
            print('CASE', case, 'kind', kind, 'rows', rows, 'R', R, 'eta', eta, 'rho', rho, 'err', rel_fro(got, ref), flush=True)
        if case % 50 == 49:
            print('...', case + 1, 'cases,', bad, 'bad', flush=True)
print('cases', n_cases, 'bad', bad)
