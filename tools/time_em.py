#!/usr/bin/env python3
"""Development timing of the EM (Z.miss) path at 1000^3, R = 20, fp32 tensor: outer iteration with and without a mask.
usage: time_em.py [missing fraction, default 0.2]"""
import copy, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module('matlab-code_amd')
capi = importlib.import_module('matlab-code_amd._capi')
import ctypes as C
import bench
n, R = 1000, 20
frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
eng = pkg.Engine(0)
rng = np.random.default_rng(1)
for masked in (False, True):
    Z = bench.build_Z(n, n, n, R, seed=0, noise=0.05); Z['_ranks'] = [R] * 3
    io = dict(lambdas_init=[[1] * R], nvecs=0, distr=[lambda a, b: rng.random((a, b))] * 3, normalize=1)
    pkg.build_model(eng, Z, 'f32')
    if masked:
        mk = (np.random.default_rng(2).random(n * n * n, dtype=np.float32) > frac).astype(np.uint8)
        capi.check(eng.lib.aoadmm_tensor_mask_upload(eng.h, 0, mk.ctypes.data_as(C.POINTER(C.c_uint8))))
    G = pkg.init_coupled_AOADMM_CMTF(Z, io, rng=rng, engine=eng)
    pkg.upload_state(eng, Z, G)
    t = {}
    for k in (3, 3, 23):
        opts = dict(MaxOuterIters=k, MaxInnerIters=5, AbsFuncTol=0.0, OuterRelTol=0.0, innerRelPrTol_coupl=0.0, innerRelPrTol_constr=0.0,
                    innerRelDualTol_coupl=0.0, innerRelDualTol_constr=0.0, bsum=0)
        pkg.upload_state(eng, Z, G)
        eng.synchronize()
        t0 = time.perf_counter()
        pkg.run_solver(eng, opts, 3, has_missing=masked)
        t[k] = time.perf_counter() - t0
    print('1000^3 R=20 fp32, mask=%s (%.1f %% missing): %.3f ms per outer iteration' % (masked, 100 * frac if masked else 0.0, (t[23] - t[3]) / 20 * 1e3), flush=True)
eng.close()
