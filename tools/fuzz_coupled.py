#!/usr/bin/env python3
"""Randomised parity sweep over the coupled / PARAFAC2 model families of the example scripts (development tool, GPU):
script 3 (matrix + CP, coupling type 4), two CP tensors with an exact coupling (type 0), script 1 (CP + PARAFAC2) and a
lone PARAFAC2 block; random mode lengths, constraint cells on the uncoupled modes, inner iteration counts.
usage: fuzz_coupled.py [ncases] [seed0]"""
import copy, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
pkg = importlib.import_module('matlab-code_amd')
from oracle import aoadmm as OA
from helpers import cp_cp_exact_model, script3_model, script1_model, options, rel_fro

ROWCELLS = [None, ('non-negativity',), ('box', 0.0, 0.8), ('simplex row-wise', 1.0), ('l1 regularization', 0.01), ('ridge', 0.05),
            ('l2-ball', 1.0), ('non-negative l2-ball', 1.0), ('non-negative l2-sphere', 1.0), ('l2 regularization', 0.01),
            ('TV regularization', 0.005), ('unimodality', True), ('simplex column-wise', 1.0)]

def flat(x, out):
    if x is None:
        return
    if isinstance(x, (list, tuple)):
        for y in x: flat(y, out)
    elif isinstance(x, dict):
        for k in sorted(x): flat(x[k], out)
    else:
        out.append(np.asarray(x, dtype=float))

def one_case(eng, case):
    """Runs one randomly drawn model on both MTTKRP paths; returns the list of mismatch reports (empty = parity)."""
    fails = []
    rng = np.random.default_rng(20_000 + case)
    fam = case % 3
    rows = int(rng.integers(5, 320))
    if fam == 0:
        Z, io = script3_model(rng, rows=rows)
        free = [1, 2, 4]
    elif fam == 1:
        Z, io = cp_cp_exact_model(rng, rows=rows)
        free = [1, 2, 4, 5]
    else:
        Z, io = script1_model(rng, dims=(int(rng.integers(5, 60)), int(rng.integers(5, 40)), int(rng.integers(5, 40))),
                              K=int(rng.integers(2, 40)), Jk=int(rng.integers(4, 50)), noise=0.05)
        free = [1, 2]
    Z = dict(Z); Z['constraints'] = list(Z['constraints']); Z['constrained_modes'] = list(Z['constrained_modes'])
    for m in free:
        c = ROWCELLS[int(rng.integers(0, len(ROWCELLS)))]
        Z['constraints'][m] = c
        Z['constrained_modes'][m] = 0 if c is None else 1
    inner = int(rng.integers(1, 8))
    opt = options(MaxOuterIters=int(rng.integers(2, 6)), MaxInnerIters=inner)
    for path in ('one-launch', 'tensor-pass'):
        if path == 'tensor-pass':
            os.environ['AOADMM_NO_SMALL_MTTKRP'] = '1'
        else:
            os.environ.pop('AOADMM_NO_SMALL_MTTKRP', None)
        try:
            G = OA.init_coupled_AOADMM_CMTF({**Z, 'prox_operators': None}, io, rng=np.random.default_rng(case))
            _, Fo, _, oo = OA.cmtf_AOADMM(Z, alg_options=opt, init=copy.deepcopy(G))
            _, Fg, _, og = pkg.cmtf_AOADMM(Z, alg_options=opt, init=copy.deepcopy(G), engine=eng)
            a, b = [], []
            for key in ('fac', 'constraint_fac', 'coupling_fac'):
                flat(Fo[key], a); flat(Fg[key], b)
            errs = [rel_fro(y, x) for x, y in zip(a, b)]
            ok = len(a) == len(b) and all(e < 1e-7 for e in errs) and np.array_equal(og['innerIters'], oo['innerIters'])
            errs = ['%.1e' % max(errs)]
        except Exception as e:
            ok = False
            errs = [repr(e)[:300]]
        if not ok:
            fails.append(('CASE', case, path, 'family', fam, 'rows', rows, Z['constraints'], 'inner', inner, errs))
    os.environ.pop('AOADMM_NO_SMALL_MTTKRP', None)
    return fails


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    eng = pkg.Engine(0)
    bad = 0
    for case in range(s0, s0 + n):
        for f in one_case(eng, case):
            bad += 1
            print(*f, flush=True)
        if case % 10 == 9:
            print('... %d cases, %d bad' % (case - s0 + 1, bad), flush=True)
    os.environ.pop('AOADMM_NO_SMALL_MTTKRP', None)
    print('cases', n, 'bad', bad)
    eng.close()

if __name__ == '__main__':
    main()
