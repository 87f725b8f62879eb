#!/usr/bin/env python3
"""Development check: op-level fp32 MTTKRP (all modes) against the oracle for a sweep of shapes/ranks.
With AOADMM_FORCE_LEAD=1 modes 2,3 go through the leading-mode contraction kernel."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module('matlab-code_amd')
from oracle.tensor_ops import mttkrp
eng = pkg.Engine(0)
rng = np.random.default_rng(0)
shapes = [(200, 17, 23), (192, 17, 23), (208, 17, 23), (256, 9, 16), (200, 16, 24), (131, 37, 29), (64, 8, 16), (130, 8, 16), (320, 5, 7)]
for dims in shapes:
    for R in (7, 16, 20, 33):
        X = rng.standard_normal(dims)
        U = [rng.standard_normal((n, R)) for n in dims]
        errs = []
        for n in range(3):
            ref = mttkrp(X, U, n)
            got = eng.mttkrp(X, U, n, precision='f32')
            errs.append(np.linalg.norm(got - ref) / np.linalg.norm(ref))
        flag = 'BAD' if max(errs) > 1e-5 else 'ok'
        print(dims, R, ' '.join('%.1e' % e for e in errs), flag, flush=True)
eng.close()
