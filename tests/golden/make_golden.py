#!/usr/bin/env python3
"""Builds the fixtures of tests/golden/ (run in the development container, where /root/reference exists):

  script11_data.npz      the only data files the reference ships: functions_for_example_scripts/noisy_dataset.mat
                         (input of example_script11_tPARAFAC2.m) and gnd_factors.mat (its ground truth), converted
                         from MATLAB v5 .mat with scipy.io.loadmat (no pickle involved) to plain arrays
  script11_expected.npz  the oracle's output for the script-11 model on that data set: fixed init (stored), 40 outer
                         iterations with all tolerances 0 (fixed work), every factor and the objective history

  known_answers.npz      (`--known-answers`; no reference file involved) generating factors and init structs of the four
                         script-shaped known-answer cases of known_answers.py; `--export-mat DIR` also writes them as .mat

The fixtures are data only; nothing of the reference's source is stored.  tests/test_golden.py checks that the
oracle still reproduces script11_expected.npz, that a longer run recovers the ground-truth factors (known answer),
and -- on the GPU -- that the HIP path reproduces the same numbers.
"""
import copy
import os
import sys

import numpy as np
import scipy.io as sio

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = '/root/reference/functions_for_example_scripts'


def script11_model(dataset):
    """example_script11_tPARAFAC2.m:31-112: regular PARAFAC2, K = 25 slabs of 100 x 80, R = 3, tPARAFAC2(1000) on
    B_k, non-negativity on C, ridge [100, 0, 100]; the data are NOT normalised there (:105-107 are commented out)."""
    K = dataset.shape[2]
    X = [np.ascontiguousarray(dataset[:, :, k]) for k in range(K)]
    return dict(loss_function=['Frobenius'], model=['PAR2'], modes=[[1, 2, 3]],
                size=[dataset.shape[0], [dataset.shape[1]] * K, K],
                coupling=dict(lin_coupled_modes=[0, 0, 0], coupling_type=[], coupl_trafo_matrices=[None] * 3),
                constrained_modes=[0, 1, 1], constraints=[None, ('tPARAFAC2', 1000), ('non-negativity',)],
                weights=[1.0], object=[X], ridge=[100, 0, 100])


def script11_options(max_outer, tol0=True):
    t = 0.0 if tol0 else None
    return dict(Display='no', DisplayIters=100, MaxOuterIters=max_outer, MaxInnerIters=5,
                AbsFuncTol=0.0 if tol0 else 1e-14, OuterRelTol=0.0 if tol0 else 1e-8,
                innerRelPrTol_coupl=0.0 if tol0 else 1e-4, innerRelPrTol_constr=0.0 if tol0 else 1e-4,
                innerRelDualTol_coupl=0.0 if tol0 else 1e-4, innerRelDualTol_constr=0.0 if tol0 else 1e-4,
                bsum=0, eps_log=1e-10)


def script11_init(Z, seed=11):
    from oracle import aoadmm as OA
    rng = np.random.default_rng(seed)
    io = dict(lambdas_init=[[1, 1, 1]], nvecs=0, distr=[lambda a, b: rng.random((a, b))] * 3, normalize=0)
    return OA.init_coupled_AOADMM_CMTF({**Z, 'prox_operators': None}, io, rng=rng)


def pack_G(G):
    """dict of plain arrays for the PARAFAC2 state (p = 0)"""
    out = {'fac0': G['fac'][0], 'fac1': np.stack(G['fac'][1]), 'fac2': G['fac'][2],
           'Z1': np.stack(G['constraint_fac'][1]), 'Z2': G['constraint_fac'][2],
           'mu1': np.stack(G['constraint_dual_fac'][1]), 'mu2': G['constraint_dual_fac'][2],
           'DeltaB': G['DeltaB'][0], 'P': np.stack(G['P'][0]), 'mu_DeltaB': np.stack(G['mu_DeltaB'][0])}
    return out


def unpack_G(d, prefix=''):
    g = lambda k: d[prefix + k]
    K = g('fac1').shape[0]
    return {'fac': [g('fac0').copy(), [g('fac1')[k].copy() for k in range(K)], g('fac2').copy()],
            'constraint_fac': [None, [g('Z1')[k].copy() for k in range(K)], g('Z2').copy()],
            'constraint_dual_fac': [None, [g('mu1')[k].copy() for k in range(K)], g('mu2').copy()],
            'coupling_fac': [], 'coupling_dual_fac': [None, None, None],
            'DeltaB': {0: g('DeltaB').copy()}, 'P': {0: [g('P')[k].copy() for k in range(K)]},
            'mu_DeltaB': {0: [g('mu_DeltaB')[k].copy() for k in range(K)]}}


def known_answers_main(export_mat=None):
    """tests/golden/known_answers.npz: generating factors (+ the noisy tensor of script 10) and the init struct of the four
    script-shaped known-answer cases (known_answers.py).  `export_mat`: directory for <case>.mat files holding the same
    arrays plus the model's data, for matlab-code_amd/mex/parity_known_answers.m."""
    import known_answers as KA
    store = {}
    for name in KA.CASES:
        t, Z, norms, G, opt = KA.build_case(name)
        for k, v in t.items():
            store['%s/truth_%s' % (name, k)] = np.asarray(v)
        for k, v in KA.pack_state(G).items():
            store['%s/init_%s' % (name, k)] = v
        if export_mat:
            os.makedirs(export_mat, exist_ok=True)
            m = {'truth_' + k: np.asarray(v) for k, v in t.items()}
            m.update({'init_' + k: v for k, v in KA.pack_state(G).items()})
            for p, obj in enumerate(Z['object']):
                m['object_%d' % (p + 1)] = np.stack(obj) if isinstance(obj, list) else np.asarray(obj)
            for p, mk in enumerate(Z.get('miss') or []):          # Z.miss{p}: 1 = observed
                m['miss_%d' % (p + 1)] = (np.stack(mk) if isinstance(mk, list) else np.asarray(mk)).astype(np.uint8)
            m['options'] = {k: v for k, v in opt.items()}
            sio.savemat(os.path.join(export_mat, name + '.mat'), m, do_compression=True)
    np.savez_compressed(os.path.join(HERE, 'known_answers.npz'), **store)
    print('wrote known_answers.npz (%d arrays)' % len(store))


def main():
    if '--known-answers' in sys.argv or '--export-mat' in sys.argv:
        d = sys.argv[sys.argv.index('--export-mat') + 1] if '--export-mat' in sys.argv else None
        sys.path.insert(0, HERE)
        known_answers_main(d)
        return
    from oracle import aoadmm as OA
    ds = sio.loadmat(os.path.join(REF, 'noisy_dataset.mat'))['dataset'].astype(np.float64)
    gnd = sio.loadmat(os.path.join(REF, 'gnd_factors.mat'))
    np.savez_compressed(os.path.join(HERE, 'script11_data.npz'), dataset=ds, A=gnd['A'], B=gnd['B'], C=gnd['C'])
    Z = script11_model(ds)
    G = script11_init(Z)
    _, Fac, _, out = OA.cmtf_AOADMM(Z, alg_options=script11_options(40), init=copy.deepcopy(G))
    exp = {'init_' + k: v for k, v in pack_G(G).items()}
    exp.update({'out_' + k: v for k, v in pack_G(Fac).items()})
    for k in ('func_val_conv', 'func_constr_conv', 'func_PAR2_coupl'):
        exp[k] = out[k]
    np.savez_compressed(os.path.join(HERE, 'script11_expected.npz'), **exp)
    print('wrote fixtures; f_tensors[-1] = %.10e' % out['f_tensors'])


if __name__ == '__main__':
    main()
