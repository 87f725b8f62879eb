"""Host-side mirror of the reference's public interface, running on the HIP engine.

    Zhat, Fac, G, out = cmtf_AOADMM(Z, alg_options=options, init=G|'random', init_options=init)
    G = init_coupled_AOADMM_CMTF(Z, init_options=init[, Delta=...])
    prox, reg = constraints_to_prox(constrained_modes, constraints, sz)

Same names, argument meaning and error behaviour as `functions/cmtf_AOADMM.m:1-206`,
`functions/init_coupled_AOADMM_CMTF.m:1-174` and `functions/constraints_to_prox.m:1-94`.
The structs are Python dicts with the MATLAB field names; `Z['modes']` keeps
MATLAB's 1-based mode numbers and `lin_coupled_modes` its 1-based coupling ids
(0 = uncoupled), so the example scripts translate line by line.

What this layer does is exactly what the MEX gateway does for MATLAB
(`matlab-code_amd/mex/`): validate, turn constraint cells into descriptors,
marshal arrays column-major through the C ABI, run `aoadmm_solve` (which replaces
`cmtf_fun_AOADMM`, cmtf_AOADMM.m:193), and pack `Zhat/Fac/out`.  Features the
device path does not cover raise `UnsupportedOnDevice` so a caller can fall
back to the original MATLAB implementation.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi as capi
from .engine import Engine, constraint_descriptor, default_engine


def _which_p(Z):
    nb_modes = len(Z['size'])
    out = [None] * nb_modes
    for i in range(nb_modes):
        for p, ms in enumerate(Z['modes']):
            if (i + 1) in list(ms):
                out[i] = p
    if any(v is None for v in out) or max(max(ms) for ms in Z['modes']) != nb_modes:
        raise ValueError('Mismatch between size and modes inputs')     # init_coupled_AOADMM_CMTF.m:32-35
    return out


def constraints_to_prox(constrained_modes, constraints, sz, engine=None):
    """functions/constraints_to_prox.m:1-94 -- returns (prox_operators, reg_func).

    Each prox operator is a callable `(x, rho) -> array` evaluated by the HIP
    kernels (op-level entry `aoadmm_op_prox`); `reg_func` entries are host
    closures (they are only used for reporting).
    """
    eng = engine or default_engine()
    n = len(constrained_modes)
    prox_ops = [None] * n
    reg = [None] * n
    for m in range(n):
        if not constrained_modes[m]:
            continue
        c = constraints[m]
        if c is None or len(c) == 0:
            raise ValueError('No constraint provided for mode %d.' % (m + 1))
        constraint_descriptor(c)     # validates / raises UnsupportedOnDevice
        prox_ops[m] = (lambda x, rho, c=c: eng.prox(c, x, rho))
        name = c[0]
        if name == 'l1 regularization':
            reg[m] = lambda x, eta=c[1]: eta * np.sum(np.abs(x))
        elif name == 'l0 regularization':
            reg[m] = lambda x, eta=c[1]: eta * float(np.count_nonzero(x))
        elif name == 'l2 regularization':
            reg[m] = lambda x, eta=c[1]: eta * np.sum(np.sqrt(np.sum(x * x, axis=0)))
        elif name == 'ridge':
            reg[m] = lambda x, eta=c[1]: eta * np.linalg.norm(x, 'fro') ** 2
        elif name == 'TV regularization':
            reg[m] = lambda x, eta=c[1]: eta * np.sum(x[1:, :] - x[:-1, :])     # quirk of :81 kept
        elif name == 'GL smoothness':
            reg[m] = lambda x, eta=c[1]: eta * np.sum((x[1:, :] - x[:-1, :]) ** 2)
    return prox_ops, reg


def _leading_eigvecs(Y, r):
    """`[U,~] = eigs(Y, r, 'LM')`: eigenvectors of the r eigenvalues of largest magnitude, in that order
    (the sign of each vector is arbitrary, in MATLAB too)."""
    w, V = np.linalg.eigh((Y + Y.T) / 2)
    idx = np.argsort(-np.abs(w), kind='stable')[:r]
    return np.asfortranarray(V[:, idx])


def cmtf_nvecs(Z, n, r, engine=None):
    """functions/cmtf_nvecs.m:1-58 for a CP block: first r left singular vectors of the mode-n unfolding (0-based n)
    of the data set that owns mode n.  The I_n x I_n Gram matrix of the unfolding comes from the device
    (`aoadmm_op_unfold_gram`), the r leading eigenvectors from LAPACK on the host."""
    eng = engine or default_engine()
    which_p = _which_p(Z)
    p = which_p[n]
    md = [m - 1 for m in Z['modes'][p]]
    Y = _resident_gram(eng, Z, p, md.index(n), int(Z['size'][n]))
    if Y is None:
        Y = eng.unfold_gram(np.asarray(Z['object'][p], dtype=np.float64), md.index(n))
    return _leading_eigvecs(Y, r)


def _resident_gram(eng, Z, p, pos, n, slab=0):
    """Gram matrix of an unfolding from the data `build_model(eng, Z)` already put on the device (cmtf_nvecs.m unfolds the
    array it already holds: no second transfer of the tensor); None when this Z is not the engine's resident model or
    the engine cannot answer locally (first mode of a row-sharded block)."""
    if getattr(eng, '_resident_model', None) is not Z:
        return None
    try:
        return eng.resident_unfold_gram(p, pos, n, slab)
    except capi.UnsupportedOnDevice:
        return None


def init_coupled_AOADMM_CMTF(Z, init_options, Delta=None, rng=None, engine=None):
    """functions/init_coupled_AOADMM_CMTF.m:1-174: random initialisation (`nvecs = 0`) or SVD-based (`nvecs = 1`,
    :50-73), the latter with the unfolding Gram matrices computed on the device."""
    if rng is None:
        rng = np.random.default_rng()
    nvecs = bool(init_options.get('nvecs', 0))
    eng_nv = (engine or default_engine()) if nvecs else None
    sz = Z['size']
    lambdas = init_options['lambdas_init']
    distr = init_options['distr']
    normalize = init_options['normalize']
    nb_modes = len(sz)
    _which_p(Z)
    lin = [int(v) for v in Z['coupling']['lin_coupled_modes']]
    nb_couplings = max(lin) if lin else 0
    ctm = Z['coupling'].get('coupl_trafo_matrices', [None] * nb_modes)
    P = len(Z['modes'])
    A = {'fac': [None] * nb_modes, 'coupling_fac': [None] * nb_couplings, 'constraint_fac': [None] * nb_modes,
         'coupling_dual_fac': [None] * nb_modes, 'constraint_dual_fac': [None] * nb_modes,
         'DeltaB': {}, 'P': {}, 'mu_DeltaB': {}}

    def colnorm(M):
        return M / np.sqrt(np.sum(M * M, axis=0))[None, :]

    for p in range(P):
        md = [m - 1 for m in Z['modes'][p]]
        R = len(lambdas[p])
        for n in md:
            if nvecs:                                                           # :50-73
                if Z['model'][p] == 'CP':
                    A['fac'][n] = cmtf_nvecs(Z, n, R, eng_nv)
                elif md.index(n) == 0:
                    Y = _resident_gram(eng_nv, Z, p, 0, int(sz[n]))
                    if Y is None:
                        M = np.hstack([np.asarray(Xk, dtype=np.float64) for Xk in Z['object'][p]])
                        Y = eng_nv.unfold_gram(M, 0)
                    A['fac'][n] = _leading_eigvecs(Y, R)
                elif md.index(n) == 1:
                    A['DeltaB'][p] = rng.random((R, R))
                    A['fac'][n] = []
                    A['P'][p] = []
                    A['mu_DeltaB'][p] = []
                    for k in range(len(sz[n])):
                        Y = _resident_gram(eng_nv, Z, p, 1, int(sz[n][k]), k)
                        if Y is None:
                            Y = eng_nv.unfold_gram(np.asarray(Z['object'][p][k], dtype=np.float64), 1)
                        A['fac'][n].append(_leading_eigvecs(Y, R))
                        A['P'][p].append(np.eye(sz[n][k], R))
                        A['mu_DeltaB'][p].append(rng.random((sz[n][k], R)))
                else:
                    A['fac'][n] = np.ones((sz[n], R))
                continue
            if Z['model'][p] == 'PAR2' and md.index(n) == 1:
                A['DeltaB'][p] = rng.random((R, R))
                A['fac'][n] = []
                A['P'][p] = []
                A['mu_DeltaB'][p] = []
                for k in range(len(sz[n])):
                    F = np.asarray(distr[n](sz[n][k], R), dtype=np.float64)
                    A['P'][p].append(np.eye(sz[n][k], R))
                    A['mu_DeltaB'][p].append(rng.random((sz[n][k], R)))
                    A['fac'][n].append(colnorm(F) if normalize else F)
            else:
                F = np.asarray(distr[n](sz[n], R), dtype=np.float64)
                A['fac'][n] = colnorm(F) if normalize else F
    if any(Z['constrained_modes']):
        prox_ops, _ = constraints_to_prox(Z['constrained_modes'], Z['constraints'], sz, engine)
        for p in range(P):
            md = [m - 1 for m in Z['modes'][p]]
            for n in md:
                if not Z['constrained_modes'][n]:
                    continue
                if Z['model'][p] == 'PAR2' and md.index(n) == 1:
                    A['constraint_fac'][n] = []
                    A['constraint_dual_fac'][n] = []
                    for k in range(len(sz[n])):
                        Zk = np.asarray(distr[n](*A['fac'][n][k].shape), dtype=np.float64)
                        if Z['constraints'][n][0] != 'tPARAFAC2':
                            Zk = prox_ops[n](Zk, 1.0)
                        A['constraint_fac'][n].append(Zk)
                        A['constraint_dual_fac'][n].append(rng.random(A['fac'][n][k].shape))
                else:
                    if Z['constraints'][n][0] == 'tPARAFAC2':
                        raise ValueError('The tPARAFAC2 constraint can only be impsed on the second mode of a PARAFAC2 model')
                    Zn = np.asarray(distr[n](*A['fac'][n].shape), dtype=np.float64)
                    A['constraint_fac'][n] = prox_ops[n](Zn, 1.0)
                    A['constraint_dual_fac'][n] = rng.random(A['fac'][n].shape)
    for n in range(nb_couplings):
        cmodes = [i for i, v in enumerate(lin) if v == n + 1]
        mode1 = cmodes[0]
        ct = int(Z['coupling']['coupling_type'][n])
        F1 = A['fac'][mode1]
        if ct == 0:
            A['coupling_fac'][n] = rng.random(F1.shape)
            for m in cmodes:
                A['coupling_dual_fac'][m] = rng.random(A['coupling_fac'][n].shape)
        elif ct == 1:
            A['coupling_fac'][n] = rng.random((ctm[mode1].shape[0], F1.shape[1]))
            for m in cmodes:
                A['coupling_dual_fac'][m] = rng.random(A['coupling_fac'][n].shape)
        elif ct == 2:
            A['coupling_fac'][n] = rng.random((F1.shape[0], ctm[mode1].shape[1]))
            for m in cmodes:
                A['coupling_dual_fac'][m] = rng.random(A['coupling_fac'][n].shape)
        elif ct == 3:
            A['coupling_fac'][n] = rng.random((ctm[mode1].shape[1], F1.shape[1]))
            for m in cmodes:
                A['coupling_dual_fac'][m] = rng.random(A['fac'][m].shape)
        elif ct == 4:
            A['coupling_fac'][n] = rng.random((F1.shape[0], ctm[mode1].shape[0]))
            for m in cmodes:
                A['coupling_dual_fac'][m] = rng.random(A['fac'][m].shape)
        else:
            A['coupling_fac'][n] = rng.random(np.shape(Delta[n]))
            for m in cmodes:
                A['coupling_dual_fac'][m] = rng.random((A['coupling_fac'][n].shape[0], A['fac'][m].shape[1]))
    return A


def _make_options(alg_options):
    o = capi.Options()
    for name in ('MaxOuterIters', 'MaxInnerIters', 'AbsFuncTol', 'OuterRelTol', 'innerRelPrTol_coupl',
                 'innerRelPrTol_constr', 'innerRelDualTol_coupl', 'innerRelDualTol_constr', 'bsum'):
        if name not in alg_options:
            raise KeyError("Reference to non-existent field '%s'." % name)   # MATLAB: missing option field errors
        setattr(o, name, alg_options[name])
    o.bsum = int(bool(alg_options['bsum']))
    if o.bsum:
        o.bsum_weight = float(alg_options['bsum_weight'])
    o.iter_start_PAR2Bkconstraint = int(alg_options.get('iter_start_PAR2Bkconstraint', 0))   # cmtf_fun_AOADMM.m:7-9
    if 'increase_factor_rhoBk' in alg_options:
        o.has_increase_factor_rhoBk = 1
        o.increase_factor_rhoBk = float(alg_options['increase_factor_rhoBk'])
    hip = alg_options.get('hip', {})
    o.use_dimtree = int(hip.get('use_dimtree', 1))
    o.no_permuted_copy = int(hip.get('no_permuted_copy', 0))
    o.par2_slab_sharding = int(hip.get('par2_slab_sharding', 0))
    return o


def build_model(eng, Z, precision='f64'):
    """Describe the struct Z to the engine and upload Z.object (cmtf_AOADMM.m:23-41,124-156)."""
    lib = eng.lib
    nb_modes = len(Z['size'])
    which_p = _which_p(Z)
    P = len(Z['object'])
    lin = [int(v) for v in Z['coupling']['lin_coupled_modes']]
    nb_couplings = max(lin) if lin else 0
    if nb_couplings != len(Z['coupling']['coupling_type']):
        raise ValueError('Mismatch between number of coulings and coupling types')   # check_data_input.m:17-19
    for p in range(P):
        if Z['loss_function'][p] != 'Frobenius':
            raise capi.UnsupportedOnDevice(capi.ERR_UNSUPPORTED,
                                           "loss '%s' needs the L-BFGS-B path of the MATLAB code" % Z['loss_function'][p])
    miss = Z.get('miss') if Z.get('miss') is not None else [None] * P
    eng._resident_model = None
    capi.check(lib.aoadmm_model_begin(eng.h, nb_modes, P, nb_couplings))
    R_of = {}
    for p in range(P):
        md = [m - 1 for m in Z['modes'][p]]
        for m in md:
            R_of[m] = None
    # ranks come from the initial factors when given; here from the caller via Z['_ranks']
    ranks = Z['_ranks']
    for m in range(nb_modes):
        p = which_p[m]
        md = [q - 1 for q in Z['modes'][p]]
        if Z['model'][p] == 'PAR2' and md.index(m) == 1:
            rows = (C.c_int64 * len(Z['size'][m]))(*[int(v) for v in Z['size'][m]])
            capi.check(lib.aoadmm_model_set_mode_slabs(eng.h, m, len(Z['size'][m]), rows, int(ranks[m])))
        else:
            capi.check(lib.aoadmm_model_set_mode(eng.h, m, int(Z['size'][m]), int(ranks[m])))
    for p in range(P):
        md = [m - 1 for m in Z['modes'][p]]
        arr = (C.c_int * len(md))(*md)
        if Z['model'][p] == 'CP':
            capi.check(lib.aoadmm_model_add_cp(eng.h, p, len(md), arr, float(Z['weights'][p])))
        elif Z['model'][p] == 'PAR2':
            capi.check(lib.aoadmm_model_add_par2(eng.h, p, arr, float(Z['weights'][p])))
        else:
            raise ValueError("unknown model '%s'" % Z['model'][p])
    for m in range(nb_modes):
        if Z['constrained_modes'][m]:
            c = Z['constraints'][m]
            if c is None or len(c) == 0:
                raise ValueError('No constraint provided for mode %d.' % (m + 1))       # constraints_to_prox.m:10-12
            cid, params, Lmat = constraint_descriptor(c)
            capi.check(lib.aoadmm_model_set_constraint(eng.h, m, cid, capi.dptr(params) if params.size else None,
                                                       params.size, capi.dptr(Lmat) if Lmat is not None else None))
    ctm = Z['coupling'].get('coupl_trafo_matrices', [None] * nb_modes)
    ctm2 = Z['coupling'].get('coupl_trafo_matrices2', [None] * nb_modes)
    keep = []
    for m in range(nb_modes):
        H = capi.as_f(ctm[m]) if (lin[m] and ctm[m] is not None) else None
        H2 = capi.as_f(ctm2[m]) if (lin[m] and ctm2[m] is not None) else None
        keep += [H, H2]
        capi.check(lib.aoadmm_model_set_coupling(
            eng.h, m, lin[m] - 1, capi.dptr(H), H.shape[0] if H is not None else 0, H.shape[1] if H is not None else 0,
            capi.dptr(H2), H2.shape[0] if H2 is not None else 0, H2.shape[1] if H2 is not None else 0))
    for n in range(nb_couplings):
        capi.check(lib.aoadmm_model_set_coupling_type(eng.h, n, int(Z['coupling']['coupling_type'][n])))
    if Z.get('ridge') is not None:
        r = np.asarray(Z['ridge'], dtype=np.float64)
        capi.check(lib.aoadmm_model_set_ridge(eng.h, capi.dptr(r)))
    capi.check(lib.aoadmm_model_end(eng.h))
    prec = capi.PREC_F32 if precision == 'f32' else capi.PREC_F64
    for p in range(P):
        if Z['model'][p] == 'CP':
            obj = Z['object'][p]
            if isinstance(obj, dict) and obj.get('synthetic'):
                capi.check(lib.aoadmm_tensor_synth(eng.h, p, int(obj['rank']), int(obj['seed']), float(obj['noise']), prec))
            else:
                X = capi.as_f(obj)
                md = [m - 1 for m in Z['modes'][p]]
                if tuple(X.shape) != tuple(int(Z['size'][m]) for m in md):
                    raise ValueError('Z.object{%d} has size %s, Z.size says %s' % (p + 1, X.shape, [Z['size'][m] for m in md]))
                capi.check(lib.aoadmm_tensor_upload(eng.h, p, capi.dptr(X), prec))
            if miss[p] is not None:                                                  # cmtf_AOADMM.m:78-97
                if isinstance(obj, dict):
                    raise ValueError('Z.miss needs explicit data in Z.object{%d}' % (p + 1))
                mk = np.asarray(miss[p])
                if tuple(mk.shape) != tuple(X.shape):
                    raise ValueError('Z.miss{%d} size does not match Z.object{%d}.' % (p + 1, p + 1))
                mk = np.asfortranarray(mk != 0, dtype=np.uint8)
                capi.check(lib.aoadmm_tensor_mask_upload(eng.h, p, mk.ctypes.data_as(C.POINTER(C.c_uint8))))
        else:
            # the slabs back to back (each I x J_k, column-major) in one transfer
            Xall = np.concatenate([np.asarray(Xk, dtype=np.float64).ravel(order='F') for Xk in Z['object'][p]])
            capi.check(lib.aoadmm_par2_slab_upload(eng.h, p, capi.ALL_SLABS, capi.dptr(Xall)))
            if miss[p] is not None:                                                  # :98-120
                K = len(Z['object'][p])
                if not isinstance(miss[p], (list, tuple)) or len(miss[p]) != K:
                    raise ValueError('Z.miss{%d} must be a cell array of length %d for PAR2.' % (p + 1, K))
                for k in range(K):
                    mk = np.asarray(miss[p][k])
                    if not np.all((mk == 0) | (mk == 1)):
                        raise ValueError('Z.miss{%d}{%d} must be a logical or binary (0/1) array.' % (p + 1, k + 1))
                    if tuple(mk.shape) != tuple(np.asarray(Z['object'][p][k]).shape):
                        raise ValueError('Z.miss{%d}{%d} size does not match Z.object{%d}{%d}.' % (p + 1, k + 1, p + 1, k + 1))
                    mk = np.asfortranarray(mk != 0, dtype=np.uint8)
                    capi.check(lib.aoadmm_par2_slab_mask_upload(eng.h, p, k, mk.ctypes.data_as(C.POINTER(C.c_uint8))))
    eng._resident_model = Z            # init_coupled_AOADMM_CMTF(nvecs = 1) on this Z takes its Gram matrices from here


def _put_cells(eng, field, index, cells):
    """A cell array of J_k x R matrices -> the device in one transfer (slab = AOADMM_ALL_SLABS)."""
    cells = [np.asarray(c, dtype=np.float64) for c in cells]
    cols = cells[0].shape[1]
    packed = np.concatenate([c.ravel(order='F') for c in cells])
    capi.check(eng.lib.aoadmm_state_set(eng.h, field, index, capi.ALL_SLABS, capi.dptr(packed),
                                        sum(c.shape[0] for c in cells), cols))


def _get_cells(eng, field, index, shapes):
    rows = sum(s[0] for s in shapes)
    cols = shapes[0][1]
    packed = np.zeros(rows * cols)
    capi.check(eng.lib.aoadmm_state_get(eng.h, field, index, capi.ALL_SLABS, capi.dptr(packed), rows, cols))
    out, o = [], 0
    for (r, c) in shapes:
        out.append(np.array(packed[o:o + r * c].reshape((r, c), order='F'), order='F'))
        o += r * c
    return out


def _put(eng, field, index, slab, a):
    a = capi.as_f(a)
    if a.ndim == 1:
        a = a.reshape(-1, 1, order='F')
    capi.check(eng.lib.aoadmm_state_set(eng.h, field, index, slab, capi.dptr(a), a.shape[0], a.shape[1]))


def _get(eng, field, index, slab, shape):
    out = np.zeros(shape, order='F')
    capi.check(eng.lib.aoadmm_state_get(eng.h, field, index, slab, capi.dptr(out), shape[0], shape[1]))
    return out


def upload_state(eng, Z, G):
    """The struct G -> device (init_coupled_AOADMM_CMTF.m:41-45)."""
    nb_modes = len(Z['size'])
    for m in range(nb_modes):
        F = G['fac'][m]
        if isinstance(F, (list, tuple)):
            _put_cells(eng, capi.F_FAC, m, F)
        else:
            _put(eng, capi.F_FAC, m, 0, F)
        for field, key in ((capi.F_CONSTRAINT_FAC, 'constraint_fac'), (capi.F_CONSTRAINT_DUAL, 'constraint_dual_fac'),
                           (capi.F_COUPLING_DUAL, 'coupling_dual_fac')):
            v = G.get(key, [None] * nb_modes)[m]
            if v is None or (isinstance(v, (list, tuple)) and len(v) == 0):
                continue
            if isinstance(v, (list, tuple)):
                _put_cells(eng, field, m, v)
            else:
                _put(eng, field, m, 0, v)
    for n, D in enumerate(G.get('coupling_fac', [])):
        if D is not None:
            _put(eng, capi.F_COUPLING_FAC, n, 0, D)
    for p, DB in G.get('DeltaB', {}).items():
        _put(eng, capi.F_DELTAB, p, 0, DB)
        _put_cells(eng, capi.F_P, p, G['P'][p])
        _put_cells(eng, capi.F_MU_DELTAB, p, G['mu_DeltaB'][p])


def download_state(eng, Z, G):
    """Device -> a struct with the same fields as G (the `Fac` output, cmtf_AOADMM.m:193)."""
    nb_modes = len(Z['size'])
    out = {k: (list(v) if isinstance(v, list) else dict(v) if isinstance(v, dict) else v) for k, v in G.items()}

    def pull(field, index, ref):
        if isinstance(ref, (list, tuple)):
            return _get_cells(eng, field, index, [np.shape(rk) for rk in ref])
        ref = np.asarray(ref)
        shp = ref.shape if ref.ndim == 2 else (ref.shape[0], 1)
        return _get(eng, field, index, 0, shp)

    for m in range(nb_modes):
        out['fac'][m] = pull(capi.F_FAC, m, G['fac'][m])
        for field, key in ((capi.F_CONSTRAINT_FAC, 'constraint_fac'), (capi.F_CONSTRAINT_DUAL, 'constraint_dual_fac'),
                           (capi.F_COUPLING_DUAL, 'coupling_dual_fac')):
            v = G.get(key, [None] * nb_modes)[m]
            if v is None or (isinstance(v, (list, tuple)) and len(v) == 0):
                continue
            out[key][m] = pull(field, m, v)
    for n, D in enumerate(G.get('coupling_fac', [])):
        if D is not None:
            out['coupling_fac'][n] = pull(capi.F_COUPLING_FAC, n, D)
    for p, DB in G.get('DeltaB', {}).items():
        out['DeltaB'][p] = pull(capi.F_DELTAB, p, DB)
        out['P'][p] = pull(capi.F_P, p, G['P'][p])
        out['mu_DeltaB'][p] = pull(capi.F_MU_DELTAB, p, G['mu_DeltaB'][p])
    return out


def _display_line(it, f, frm=None):
    """One row of the iteration table, format of cmtf_fun_AOADMM.m:55-58."""
    line = '%6d %12f %12f %12f %17f %12f' % (it, sum(f), f[0], f[1], f[2], f[3])
    return line + (' %12f' % frm if frm is not None else '')


def run_solver(eng, alg_options, nb_modes, has_missing=False):
    """`[Fac,out] = cmtf_fun_AOADMM(...)` (cmtf_AOADMM.m:193) -> the `out` struct (cmtf_fun_AOADMM.m:480-494)."""
    o = _make_options(alg_options)
    n = int(o.MaxOuterIters) + 1
    bufs = {k: np.zeros(n) for k in ('func_val_conv', 'func_coupl_conv', 'func_constr_conv', 'func_PAR2_coupl', 'time_at_it')}
    inner = np.zeros((nb_modes, max(int(o.MaxOuterIters), 1)), order='F')
    frm = np.full(n, np.nan)
    res = capi.Result()
    for k, b in bufs.items():
        setattr(res, k, capi.dptr(b))
    res.innerIters = capi.dptr(inner)
    res.func_rel_missing = capi.dptr(frm)
    # options.Display (cmtf_fun_AOADMM.m:44-59, :462-468, :498-504): 'iter' lines come live from inside the solve
    display = str(alg_options.get('Display', 'no')) if isinstance(alg_options, dict) else 'no'
    every = int(alg_options.get('DisplayIters', 10)) if isinstance(alg_options, dict) else 10
    rank0 = eng.comm_rank()[0] == 0
    cb = None
    if display in ('iter', 'final') and rank0:
        print(' Iter  f total      f tensors      f couplings    f constraints    f PAR2 couplings' +
              ('  f_rel_miss' if has_missing else ''))
        print('------ ------------ -------------  -------------- ---------------- ----------------')
    if display == 'iter' and rank0:
        def _line(_user, it, f, frm):
            print(_display_line(it, [f[i] for i in range(4)], frm if has_missing else None), flush=True)
        cb = capi.PROGRESS_FN(_line)
        capi.check(eng.lib.aoadmm_set_progress(eng.h, cb, None, every))
    try:
        capi.check(eng.lib.aoadmm_solve(eng.h, C.byref(o), C.byref(res)))
    finally:
        if cb is not None:
            capi.check(eng.lib.aoadmm_set_progress(eng.h, capi.PROGRESS_FN(0), None, 0))
    it = int(res.OuterIterations)
    if display in ('iter', 'final') and rank0:                                 # :496-503
        print('%6d %12f %12f %12f %12f %12f' % (it, res.f_tensors + res.f_couplings + res.f_constraints + res.f_PAR2_couplings,
                                                res.f_tensors, res.f_couplings, res.f_constraints, res.f_PAR2_couplings) +
              (' %12f' % res.f_rel_missing if has_missing else ''))
    out = {
        'f_tensors': res.f_tensors, 'f_couplings': res.f_couplings, 'f_constraints': res.f_constraints,
        'f_PAR2_couplings': res.f_PAR2_couplings, 'f_rel_missing': res.f_rel_missing,
        'OuterIterations': it,
        'innerIters': inner[:, :max(it, 1)].copy(),
    }
    for k, b in bufs.items():
        out[k] = b[:it + 1].copy()
    if has_missing:
        out['func_rel_missing'] = frm[:it + 1].copy()                          # cmtf_fun_AOADMM.m:490-492
    names = ['f_tensors', 'f_couplings', 'f_constraints', 'f_PAR2_couplings']
    if res.exit_code == 0:
        out['exit_flag'] = 'maxIterations'                                   # make_exit_flag.m:4-5
    else:
        out['exit_flag'] = {nm: ('AbsFuncTol' if res.exit_abs[i] else 'RelFuncTol') for i, nm in enumerate(names)}
    return out


def cmtf_AOADMM(Z, alg_options=None, init='random', init_options=None, rng=None, engine=None, precision='f64'):
    """functions/cmtf_AOADMM.m:1-206.  Returns (Zhat, Fac, G, out)."""
    if alg_options is None:
        raise ValueError('alg_options are missing as input in cmtf_AOADMM.')
    eng = engine or default_engine()
    Z = dict(Z)
    which_p = _which_p(Z)
    for m, c in enumerate(Z['constraints']):                                          # cmtf_AOADMM.m:33-41
        if Z['constrained_modes'][m] and c is not None and c[0] == 'tPARAFAC2':
            p = which_p[m]
            if Z['model'][p] != 'PAR2' or [q - 1 for q in Z['modes'][p]].index(m) != 1:
                raise ValueError('The tPARAFAC2 constraint can only be impsed on the second mode of a PARAFAC2 model')
    if isinstance(init, dict):                                                        # :44-53
        G = init
    elif isinstance(init, str) and init.lower() == 'random':
        if init_options is None:
            raise ValueError('init_options are missing as input in cmtf_AOADMM.')
        G = init_coupled_AOADMM_CMTF(Z, init_options=init_options, rng=rng, engine=eng)
    else:
        raise ValueError('Initialization type not supported')
    nb_modes = len(Z['size'])
    ranks = []
    for m in range(nb_modes):
        F = G['fac'][m]
        ranks.append(int((F[0] if isinstance(F, (list, tuple)) else F).shape[1]))
    for p in range(len(Z['object'])):                                                 # :55-65
        if Z['model'][p] == 'PAR2':
            md = [q - 1 for q in Z['modes'][p]]
            for k, jk in enumerate(Z['size'][md[1]]):
                if jk < ranks[md[0]]:
                    raise ValueError('Number of components for PARAFAC2 is larger than size of slice %d of data tensor %d.' % (k + 1, p + 1))
    Z['_ranks'] = ranks
    build_model(eng, Z, precision)
    upload_state(eng, Z, G)
    out = run_solver(eng, alg_options, nb_modes,
                     has_missing=Z.get('miss') is not None and any(m is not None for m in Z['miss']))
    Fac = download_state(eng, Z, G)
    Zhat = []
    for p in range(len(Z['object'])):                                                 # :197-206
        md = [q - 1 for q in Z['modes'][p]]
        if Z['model'][p] == 'CP':
            Zhat.append([Fac['fac'][m] for m in md])
        else:
            Zhat.append({'A': Fac['fac'][md[0]], 'Bk': Fac['fac'][md[1]], 'C': Fac['fac'][md[2]]})
    return Zhat, Fac, G, out
