#!/usr/bin/env python3
"""Development timing of BASELINE config 1 (example_script1: CP 40x50x60 + PARAFAC2 K = 20, first modes coupled, non-negative)."""
import copy, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
pkg = importlib.import_module('matlab-code_amd')
from oracle import aoadmm as OA
from helpers import script1_model, options
rng = np.random.default_rng(4)
Z, io = script1_model(rng, dims=(40, 50, 60))
G = OA.init_coupled_AOADMM_CMTF({**Z, 'prox_operators': None}, io, rng=np.random.default_rng(7))
eng = pkg.Engine(0)
t = {}
for n in (50, 50, 300):
    t0 = time.perf_counter()
    pkg.cmtf_AOADMM(Z, alg_options=options(MaxOuterIters=n), init=copy.deepcopy(G), engine=eng)
    t[n] = time.perf_counter() - t0
print('cfg1: %.3f ms per outer iteration' % ((t[300] - t[50]) / 250 * 1e3))
eng.close()
