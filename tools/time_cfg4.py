#!/usr/bin/env python3
"""Development timing of BASELINE config 4 (irregular PARAFAC2, K = 256 slabs, I = 40, J_k in 61..120, R = 3)."""
import copy, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
pkg = importlib.import_module('matlab-code_amd')
from oracle import aoadmm as OA
from helpers import script4_model, options
rng = np.random.default_rng(4)
Z, io = script4_model(rng, K=256)
G = OA.init_coupled_AOADMM_CMTF({**Z, 'prox_operators': None}, io, rng=np.random.default_rng(7))
eng = pkg.Engine(0)
times = {}
for n in (5, 50, 250):
    t = time.perf_counter()
    _, F, _, out = pkg.cmtf_AOADMM(Z, alg_options=options(MaxOuterIters=n), init=copy.deepcopy(G), engine=eng)
    times[n] = time.perf_counter() - t
    print('cfg4 K=256: %d outer iterations in %.3f s (incl. upload and download of the %d slabs)' % (n, times[n], 256), flush=True)
print('cfg4 K=256: %.3f ms per outer iteration (slope between 50 and 250 iterations)'
      % ((times[250] - times[50]) / 200 * 1e3), flush=True)
t = time.perf_counter()
OA.cmtf_AOADMM(Z, alg_options=options(MaxOuterIters=5), init=copy.deepcopy(G))
print('oracle (numpy): %.1f ms/iteration' % ((time.perf_counter() - t) / 5 * 1e3))
eng.close()
