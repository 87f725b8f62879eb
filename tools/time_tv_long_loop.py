#!/usr/bin/env python3
"""Development timing: TV on a 6000-row mode inside the ADMM loop (warm-started, fused dual update), i.e. the in-loop use of
prox_tv_fast_k<1024, true>; run under rocprofv3 --kernel-trace --stats."""
import copy, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
pkg = importlib.import_module('matlab-code_amd')
from helpers import cp_model, options
from oracle import aoadmm as OA
eng = pkg.Engine(0)
rng = np.random.default_rng(0)
Z, io, _ = cp_model((6000, 40, 30), 5, rng, [('TV regularization', 0.01), ('non-negativity',), ('non-negativity',)])
G = OA.init_coupled_AOADMM_CMTF({**Z, 'prox_operators': None}, io, rng=np.random.default_rng(1))
pkg.cmtf_AOADMM(Z, alg_options=options(MaxOuterIters=30), init=copy.deepcopy(G), engine=eng)
eng.close()
