#!/usr/bin/env python3
"""Development timing of the TV and GL-smoothness prox on columns beyond the LDS-resident 4096 rows (run under rocprofv3 --kernel-trace
--stats; with AOADMM_TV_SEQ_LONG=1 the one-thread scan it replaced)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module('matlab-code_amd')
eng = pkg.Engine(0)
rng = np.random.default_rng(0)
for rows in (2000, 4096, 6000, 8000, 20000, 100000):
    steps = np.repeat(rng.standard_normal((rows // 200 + 1, 20)), 200, axis=0)[:rows]
    X = steps + 0.05 * rng.standard_normal((rows, 20))
    for _ in range(5):
        eng.prox(('TV regularization', 0.3), X, 1.7)
    for _ in range(5):
        eng.prox(('GL smoothness', 0.7), X, 1.7)
eng.close()
