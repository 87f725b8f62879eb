#!/usr/bin/env python3
"""Kernel-level timing of the tensor passes on a resident synthetic tensor (development tool).
usage: python tools/perf_mttkrp.py [size] [rank] [prec] [reps]"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module('matlab-code_amd')
capi = importlib.import_module('matlab-code_amd._capi')

size = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 20
prec = sys.argv[3] if len(sys.argv) > 3 else 'f32'
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
WHICH = 1 if os.environ.get('AOADMM_FORCE_LEAD') else 0
eng = pkg.Engine(0)
Z = dict(loss_function=['Frobenius'], model=['CP'], modes=[[1, 2, 3]], size=[size] * 3,
         coupling=dict(lin_coupled_modes=[0, 0, 0], coupling_type=[], coupl_trafo_matrices=[None] * 3),
         constrained_modes=[0, 0, 0], constraints=[None] * 3, weights=[1.0],
         object=[dict(synthetic=True, rank=R, seed=0, noise=0.05)], _ranks=[R] * 3)
pkg.build_model(eng, Z, prec)
rng = np.random.default_rng(1)
G = dict(fac=[rng.random((size, R)) for _ in range(3)])
pkg.upload_state(eng, Z, G)
sx = 4 if prec == 'f32' else 8
for pos in (0, 1, 2):
    el = C.c_float()
    ms = C.c_double(); nl = C.c_int64(); by = C.c_double(); fl = C.c_double()
    WHICH = 1 if (os.environ.get('AOADMM_FORCE_LEAD') and pos != 0) else 0
    capi.check(eng.lib.aoadmm_resident_mttkrp(eng.h, 0, pos, None, C.byref(el)))      # warm-up
    capi.check(eng.lib.aoadmm_kernel_stats(eng.h, WHICH, 1, C.byref(ms), C.byref(nl), C.byref(by), C.byref(fl)))
    tot = []
    for _ in range(reps):
        capi.check(eng.lib.aoadmm_resident_mttkrp(eng.h, 0, pos, None, C.byref(el)))
        tot.append(el.value)
    capi.check(eng.lib.aoadmm_kernel_stats(eng.h, WHICH, 1, C.byref(ms), C.byref(nl), C.byref(by), C.byref(fl)))
    k_ms = ms.value / nl.value
    print('mode %d: contraction kernel %.3f ms  %.1f GB/s algorithmic (%.1f TF/s) | whole mttkrp median %.3f ms  %.1f TF/s'
          % (pos + 1, k_ms, by.value / nl.value / k_ms / 1e6, fl.value / nl.value / k_ms / 1e9, float(np.median(tot)),
             2.0 * size ** 3 * R / np.median(tot) / 1e9), flush=True)
eng.close()
