#!/usr/bin/env python3
"""Development probe: can two ranks share device 0 under RCCL? (used to rehearse the sharded path on a 1-GPU box)"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch.distributed as dist
pkg = importlib.import_module('matlab-code_amd')
rank = int(os.environ['RANK']); world = int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo')
eng = pkg.Engine(0)
pkg.init_engine_comm(eng, dist)
from helpers import cp_model, options
from oracle import aoadmm as OA
import copy
rng = np.random.default_rng(1)
Z, io, _ = cp_model((37, 14, 12), 3, rng, [('TV regularization', 0.01), ('non-negativity',), ('non-negativity',)])
G = OA.init_coupled_AOADMM_CMTF(Z, io, rng=np.random.default_rng(5))
opt = options(MaxOuterIters=6)
_, Fo, _, oo = OA.cmtf_AOADMM(Z, alg_options=opt, init=copy.deepcopy(G))
_, Fg, _, og = pkg.cmtf_AOADMM(Z, alg_options=opt, init=copy.deepcopy(G), engine=eng)
err = max(np.linalg.norm(a - b) / np.linalg.norm(a) for a, b in zip(Fo['fac'], Fg['fac']))
print('rank', rank, 'sharded-vs-oracle max rel err', err, flush=True)
dist.barrier()
