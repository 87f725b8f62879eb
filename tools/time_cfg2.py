#!/usr/bin/env python3
"""Development timing of BASELINE config 2 (CP 500^3, R = 10, non-negativity, fp64 tensor)."""
import copy, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
pkg = importlib.import_module('matlab-code_amd')
from oracle import aoadmm as OA
from helpers import options
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from time_configs import cfg2_model
rng = np.random.default_rng(4)
Z, io = cfg2_model(rng)
G = OA.init_coupled_AOADMM_CMTF({**Z, 'prox_operators': None}, io, rng=np.random.default_rng(7))
eng = pkg.Engine(0)
t = {}
for n in (5, 5, 45):
    t0 = time.perf_counter()
    pkg.cmtf_AOADMM(Z, alg_options=options(MaxOuterIters=n), init=copy.deepcopy(G), engine=eng)
    t[n] = time.perf_counter() - t0
print('cfg2: %.3f ms per outer iteration' % ((t[45] - t[5]) / 40 * 1e3))
eng.close()
