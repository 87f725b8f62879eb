#!/usr/bin/env python3
"""Development timing of the column-wise prox kernels at 2000 x 20 (run under rocprofv3 --kernel-trace --stats)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module('matlab-code_amd')
eng = pkg.Engine(0)
rng = np.random.default_rng(0)
X = rng.standard_normal((2000, 20))
for c in [('non-decreasing',), ('non-increasing',), ('unimodality', True), ('unimodality', False), ('GL smoothness', 0.7),
          ('TV regularization', 0.4), ('simplex column-wise', 1.0), ('l1-ball', 3.0), ('orthonormal',)]:
    for _ in range(5):
        eng.prox(c, X, 1.7)
eng.close()
