#!/usr/bin/env python3
"""Development timing of BASELINE config 3 (example_script3: matrix 50x70 + CP 50x30x40, partial coupling type 4)."""
import copy, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
pkg = importlib.import_module('matlab-code_amd')
from oracle import aoadmm as OA
from helpers import script3_model, options
rng = np.random.default_rng(4)
Z, io = script3_model(rng)
G = OA.init_coupled_AOADMM_CMTF({**Z, 'prox_operators': None}, io, rng=np.random.default_rng(7))
eng = pkg.Engine(0)
t = {}
for n in (50, 50, 300):
    t0 = time.perf_counter()
    pkg.cmtf_AOADMM(Z, alg_options=options(MaxOuterIters=n), init=copy.deepcopy(G), engine=eng)
    t[n] = time.perf_counter() - t0
print('cfg3: %.3f ms per outer iteration' % ((t[300] - t[50]) / 250 * 1e3))
eng.close()
