"""Oracle (test infrastructure): fp64 numpy restatement of the AO-ADMM solver.

Follows, statement by statement, the reference MATLAB sources (citations are
relative to /root/reference/):

  * `functions/cmtf_fun_AOADMM.m`       -> `cmtf_fun_AOADMM`
  * `functions/cmtf_AOADMM.m`           -> `cmtf_AOADMM`
  * `functions/init_coupled_AOADMM_CMTF.m` -> `init_coupled_AOADMM_CMTF`
  * `functions/evaluate_stopping_conditions.m`, `functions/make_exit_flag.m`,
    `functions/cp_func.m`, `functions/pca_func.m`

Scope: Frobenius loss; CP blocks (tensors and matrices) and PARAFAC2 blocks;
coupling types 0-5 for CP modes and for the PARAFAC2 C mode;
every constraint of `constraints_to_prox.m`.  Out of scope (raises):
KL/IS/beta losses (need the external L-BFGS-B MEX).

Data model (mirrors the MATLAB structs; MATLAB's 1-based *mode numbers* inside
`Z['modes']` and coupling ids inside `lin_coupled_modes` are kept so that the
example scripts translate one-to-one; list *positions* are 0-based):

  Z = dict(loss_function=[..], model=['CP','PAR2'], modes=[[1,2,3],[4,5,6]],
           size=[20,30,40, 20,[30]*20,20],
           coupling=dict(lin_coupled_modes=[1,0,0,1,0,0], coupling_type=[0],
                         coupl_trafo_matrices=[None]*6[, coupl_trafo_matrices2]),
           constrained_modes=[..], constraints=[None|tuple, ..], weights=[..],
           object=[ndarray | list of K ndarrays][, ridge=[..]])
  G = dict(fac=[ndarray | list of K ndarrays], constraint_fac, constraint_dual_fac,
           coupling_fac (per coupling), coupling_dual_fac (per mode),
           DeltaB / P / mu_DeltaB (dicts keyed by tensor position p))

This is the checker for the HIP path; it is never the thing measured or shipped.
"""
from __future__ import annotations

import copy
import time

import numpy as np
import scipy.linalg as sla

from . import prox as _prox
from .tensor_ops import full_ktensor, mttkrp, tensor_norm

inf = float('inf')


# --------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------

def _which_p(Z):
    """cmtf_fun_AOADMM.m:12-15 (0-based mode -> 0-based tensor position)."""
    nb_modes = len(Z['size'])
    out = [None] * nb_modes
    for i in range(nb_modes):
        for p, ms in enumerate(Z['modes']):
            if (i + 1) in list(ms):
                out[i] = p
    return out


def _modes0(Z, p):
    return [m - 1 for m in Z['modes'][p]]


def _K_of(Z, p):
    return len(Z['size'][_modes0(Z, p)[1]])


def _is_tensor(obj):
    return np.ndim(obj) >= 3


def _chol_lower(B):
    """`chol(B','lower')` (cmtf_fun_AOADMM.m:142); raises like MATLAB if not PD."""
    return np.linalg.cholesky(np.asarray(B).T)


def _solve_llt_right(A_inner, L):
    """`(A_inner/L')/L` (cmtf_fun_AOADMM.m:609): A_inner * inv(L*L')."""
    return sla.cho_solve((L, True), A_inner.T).T


def _mrdivide(A, B):
    """MATLAB `A/B`."""
    return np.linalg.solve(B.T, A.T).T


def _fro(x):
    # numpy scalar on purpose: x/0 gives inf/nan like MATLAB instead of raising
    return np.float64(np.linalg.norm(x, 'fro')) if np.ndim(x) == 2 else np.float64(np.linalg.norm(x))


# --------------------------------------------------------------------------
# cp_func / pca_func / stopping / exit flag
# --------------------------------------------------------------------------

def cp_func(Zt, A, Znormsqr, weight):
    """functions/cp_func.m:19-56."""
    W = np.ones((A[0].shape[1],) * 2)
    for a in A:
        W = W * (a.T @ a)                       # :24-33
    U = mttkrp(Zt, A, 0)                        # :47
    f_2 = np.sum(A[0] * U)                      # :48-49
    f_3 = np.sum(W)                             # :52
    return weight * (Znormsqr - 2.0 * f_2 + f_3)  # :55-56


def pca_func(Zm, A, Znormsqr, weight):
    """functions/pca_func.m:18-40."""
    U, V = A[0], A[1]
    f2 = 0.0
    for r in range(U.shape[1]):
        f2 += float(U[:, r] @ Zm @ V[:, r])     # :31-34 (ttv)
    W = (U.T @ U) * (V.T @ V)                   # :36
    return weight * (Znormsqr - 2.0 * f2 + np.sum(W))


def evaluate_stopping_conditions(f, f_old, options):
    """functions/evaluate_stopping_conditions.m:1-47; f, f_old = 4-tuples
    (tensors, couplings, constraints, PAR2_couplings)."""
    stops = []
    for v, vo in zip(f, f_old):
        rel = abs(vo - v) / vo if vo > 0 else abs(vo - v)
        stops.append(bool(v < options['AbsFuncTol'] or rel < options['OuterRelTol']))
    return all(stops)


def make_exit_flag(it, f, options, illconditioned=0):
    """functions/make_exit_flag.m:1-31."""
    if it > options['MaxOuterIters']:
        return 'maxIterations'
    if illconditioned:
        return 'illconditioned lin system'
    names = ['f_tensors', 'f_couplings', 'f_constraints', 'f_PAR2_couplings']
    return {n: ('AbsFuncTol' if v < options['AbsFuncTol'] else 'RelFuncTol') for n, v in zip(names, f)}


# --------------------------------------------------------------------------
# the solver core
# --------------------------------------------------------------------------

def cmtf_fun_AOADMM(Z, Znorm_const, G, options, trace=None):
    """functions/cmtf_fun_AOADMM.m:1-1431 (Frobenius loss only).

    `trace`, if a dict, receives per-outer-iteration snapshots used by the
    op-level parity tests (not part of the reference interface).
    """
    options = dict(options)
    options.setdefault('iter_start_PAR2Bkconstraint', 0)          # :7-9
    G = copy.deepcopy(G)
    lin = [int(v) for v in Z['coupling']['lin_coupled_modes']]
    couplings = sorted(set(lin))                                  # :10 unique()
    nb_modes = len(Z['size'])
    which_p = _which_p(Z)
    P = len(Z['object'])
    ctm = Z['coupling'].get('coupl_trafo_matrices', [None] * nb_modes)
    ctm2 = Z['coupling'].get('coupl_trafo_matrices2', [None] * nb_modes)
    weights = [float(w) for w in Z['weights']]
    constrained = [int(bool(c)) for c in Z['constrained_modes']]
    has_ridge = 'ridge' in Z and Z['ridge'] is not None
    # EM missing data (:29, cmtf_AOADMM.m:68-121): Z['miss'][p] is a boolean array (True = observed) for a CP
    # block, a list of K boolean matrices for a PARAFAC2 block, or None.  The imputation writes into
    # Z.object (a local copy in the reference: MATLAB value semantics), so the data are copied first.
    miss = Z.get('miss') if Z.get('miss') is not None else [None] * P
    has_missing = any(mk is not None for mk in miss)
    if has_missing:
        Z = dict(Z)
        Z['object'] = list(Z['object'])
        for p in range(P):
            if miss[p] is None:
                continue
            if Z['model'][p] == 'CP':
                Z['object'][p] = np.array(Z['object'][p], dtype=np.float64, copy=True)
            else:
                Z['object'][p] = [np.array(Xk, dtype=np.float64, copy=True) for Xk in Z['object'][p]]
    f_rel_missing = float('nan')
    for p in range(P):
        if Z['loss_function'][p] != 'Frobenius':
            raise NotImplementedError('only Frobenius loss is in scope (SURVEY 2.1)')

    G_transp_G = [None] * nb_modes
    A = [None] * nb_modes
    C = [None] * nb_modes
    B = [None] * nb_modes
    B2 = [None] * nb_modes
    L = [None] * nb_modes
    rho = [None] * nb_modes
    last_m = [0] * P
    last_mttkrp = [None] * P
    last_had = [None] * P
    innerIters = {}

    def prox_of(m):
        return Z['prox_operators'][m]

    # ---------------- nested: update_constraint  :1420-1429
    def update_constraint(m, rho_m):
        oldZ = G['constraint_fac'][m]
        if np.size(rho_m) > 1:
            G['constraint_fac'][m] = prox_of(m)(G['fac'][m] + G['constraint_dual_fac'][m], float(np.max(rho_m)))
        else:
            G['constraint_fac'][m] = prox_of(m)(G['fac'][m] + G['constraint_dual_fac'][m], float(rho_m))
        G['constraint_dual_fac'][m] = G['constraint_dual_fac'][m] + G['fac'][m] - G['constraint_fac'][m]
        return oldZ

    # ---------------- nested: residual evaluators  :1079-1210
    def eval_res_ADMM_constr(modes, oldZ):
        pr = du = 0.0
        for mm in modes:
            pr += _fro(G['fac'][mm] - G['constraint_fac'][mm]) / _fro(G['fac'][mm])
            scaling = _fro(G['constraint_dual_fac'][mm])
            d = _fro(G['constraint_fac'][mm] - oldZ[mm])
            du += d / scaling if scaling > 0 else d
        return pr / len(modes), du / len(modes)

    def eval_res_ADMM_coupl(ctype, modes, cid, oldDelta):
        pr = du = 0.0
        D = G['coupling_fac'][cid]
        dD = D - oldDelta
        for mm in modes:
            F = G['fac'][mm]
            H = ctm[mm]
            if ctype == 0:                                   # :1099-1115
                num, den, dd = F - D, F, dD
            elif ctype == 1:                                 # :1118-1134
                num, den, dd = H @ F - D, H @ F, dD
            elif ctype == 2:                                 # :1137-1153
                num, den, dd = F @ H - D, F @ H, dD
            elif ctype == 3:                                 # :1156-1172
                num, den, dd = F - H @ D, F, H @ dD
            elif ctype == 4:                                 # :1175-1191
                num, den, dd = F - D @ H, F, dD @ H
            else:                                            # :1194-1210
                H2 = ctm2[mm]
                num, den, dd = H @ F - D @ H2, F, dD @ H2
            pr += _fro(num) / _fro(den)
            scaling = _fro(G['coupling_dual_fac'][mm])
            du += _fro(dd) / scaling if scaling > 0 else _fro(dd)
        return pr / len(modes), du / len(modes)

    def is_par2_C(mm):
        pp = which_p[mm]
        return Z['model'][pp] == 'PAR2' and _modes0(Z, pp).index(mm) == 2

    def _loop_cond(ii, pc, pz, dc, dz):
        return ii <= options['MaxInnerIters'] and (
            pc > options['innerRelPrTol_coupl'] or pz > options['innerRelPrTol_constr'] or
            dc > options['innerRelDualTol_coupl'] or dz > options['innerRelDualTol_constr'])

    # ---------------- nested: ADMM_constrained_only  :591-623
    def ADMM_constrained_only(Am, Lm, m, p):
        inner_iter = 1
        pr = du = inf
        oldZ = [None] * nb_modes
        while inner_iter <= options['MaxInnerIters'] and (
                pr > options['innerRelPrTol_constr'] or du > options['innerRelDualTol_constr']):
            if is_par2_C(m):                                                      # :602-606
                for kk in range(_K_of(Z, p)):
                    A_inner = Am[kk] + rho[m][kk] / 2 * (G['constraint_fac'][m][kk, :] - G['constraint_dual_fac'][m][kk, :])
                    G['fac'][m][kk, :] = sla.cho_solve((Lm[kk], True), A_inner)
            else:
                A_inner = Am + rho[m] / 2 * (G['constraint_fac'][m] - G['constraint_dual_fac'][m])   # :608
                G['fac'][m] = _solve_llt_right(A_inner, Lm)                         # :609
            oldZ[m] = update_constraint(m, rho[m])                                  # :617
            inner_iter += 1
            pr, du = eval_res_ADMM_constr([m], oldZ)                                # :620
        return inner_iter - 1

    # ---------------- nested: coupled ADMM, types 0..5
    def ADMM_coupled(ctype, cmodes, cid):
        inner_iter = 1
        pc = pz = dc = dz = inf
        oldZ = [None] * nb_modes
        while _loop_cond(inner_iter, pc, pz, dc, dz):
            # ---- primal updates (":635", ":913" ...)
            for mm in cmodes:
                pp = which_p[mm]
                D = G['coupling_fac'][cid]
                muD = G['coupling_dual_fac'][mm]
                H = ctm[mm]
                if is_par2_C(mm) and ctype in (1, 5):                              # :710-724, :998-1010
                    # one (K*R) x (K*R) system for vec(C') with the common rhoC = mean(rho) (:712, :1000)
                    rhoC = float(np.mean(rho[mm]))
                    A_large = np.concatenate([np.ravel(A[mm][kk]) for kk in range(_K_of(Z, pp))])       # :714-716
                    img = D if ctype == 1 else D @ ctm2[mm]                        # :717 / :1005 (Delta*H2)
                    A_inner = A_large + rhoC / 2 * np.ravel(H.T @ (img - muD))     # HcI'*vec(Y') = vec((H'Y)')
                    if constrained[mm]:
                        A_inner = A_inner + rhoC / 2 * np.ravel(G['constraint_fac'][mm] - G['constraint_dual_fac'][mm])   # :719
                    x = sla.cho_solve((L[mm], True), A_inner)                        # :721
                    G['fac'][mm] = x.reshape(G['fac'][mm].shape)                     # :722 (vec of C' = rows of C back to back)
                elif is_par2_C(mm):
                    for kk in range(_K_of(Z, pp)):
                        r2 = rho[mm][kk] / 2
                        if ctype == 0:                                             # :640
                            tgt = D[kk, :] - muD[kk, :]
                        elif ctype == 2:                                           # case2
                            tgt = (D[kk, :] - muD[kk, :]) @ H.T
                        elif ctype == 3:
                            tgt = H[kk, :] @ D - muD[kk, :]
                        else:                                                      # :918
                            tgt = D[kk, :] @ H - muD[kk, :]
                        A_inner = A[mm][kk] + r2 * tgt
                        if constrained[mm]:
                            A_inner = A_inner + r2 * (G['constraint_fac'][mm][kk, :] - G['constraint_dual_fac'][mm][kk, :])
                        G['fac'][mm][kk, :] = sla.cho_solve((L[mm][kk], True), A_inner)
                else:
                    r2 = rho[mm] / 2
                    if ctype == 0:                                                 # :647
                        A_inner = A[mm] + r2 * (D - muD)
                    elif ctype == 1:                                               # case1
                        A_inner = A[mm] + r2 * (H.T @ (D - muD))
                    elif ctype == 2:
                        A_inner = A[mm] + r2 * ((D - muD) @ H.T)
                    elif ctype == 3:
                        A_inner = A[mm] + r2 * (H @ D - muD)
                    elif ctype == 4:                                               # :925
                        A_inner = A[mm] + r2 * (D @ H - muD)
                    else:                                                          # :1012
                        A_inner = A[mm] + r2 * (H.T @ (D @ ctm2[mm] - muD))
                    if constrained[mm]:
                        A_inner = A_inner + r2 * (G['constraint_fac'][mm] - G['constraint_dual_fac'][mm])
                    if ctype in (1, 5):
                        # sylvester(B2,B,A_inner): B2*X + X*B = A_inner   (:1016)
                        G['fac'][mm] = sla.solve_sylvester(B2[mm], B[mm], A_inner)
                    else:
                        G['fac'][mm] = _solve_llt_right(A_inner, L[mm])             # :651,:929
            # ---- Delta update
            oldDelta = G['coupling_fac'][cid]
            if ctype in (0, 1, 2):
                newD = np.zeros_like(oldDelta)
                sum_rho = 0.0
                for jj in cmodes:
                    rj = rho[jj]
                    if ctype == 0:                                                 # :661-675
                        if np.size(rj) > 1:
                            newD = newD + np.asarray(rj)[:, None] * (G['fac'][jj] + G['coupling_dual_fac'][jj])
                        else:
                            newD = newD + rj * (G['fac'][jj] + G['coupling_dual_fac'][jj])
                        sum_rho = sum_rho + np.asarray(rj)
                    elif ctype == 1:                                               # case1: sum(rho)
                        newD = newD + np.sum(rj) * (ctm[jj] @ G['fac'][jj] + G['coupling_dual_fac'][jj])
                        sum_rho = sum_rho + np.sum(rj)
                    else:                                                          # case2: rho'.*
                        sc = np.asarray(rj)[:, None] if np.size(rj) > 1 else rj
                        newD = newD + sc * (G['fac'][jj] @ ctm[jj] + G['coupling_dual_fac'][jj])
                        sum_rho = sum_rho + np.asarray(rj)
                if np.size(sum_rho) > 1:
                    newD = (1.0 / np.asarray(sum_rho))[:, None] * newD
                else:
                    newD = 1.0 / float(sum_rho) * newD
                G['coupling_fac'][cid] = newD
            elif ctype == 3:
                H1 = ctm[cmodes[0]]
                AA = np.zeros((H1.shape[1], H1.shape[1]))
                BB = np.zeros((H1.shape[1], G['fac'][cmodes[0]].shape[1]))
                for jj in cmodes:
                    sc = np.asarray(rho[jj])[:, None] if np.size(rho[jj]) > 1 else rho[jj]
                    AA = AA + ctm[jj].T @ (sc * ctm[jj])
                    BB = BB + ctm[jj].T @ (sc * (G['fac'][jj] + G['coupling_dual_fac'][jj]))
                G['coupling_fac'][cid] = np.linalg.solve(AA, BB)
            elif ctype == 4:                                                       # :939-963
                H1 = ctm[cmodes[0]]
                AA = np.zeros((H1.shape[0], H1.shape[0]))
                BB = np.zeros((G['fac'][cmodes[0]].shape[0], H1.shape[0]))
                par2 = None
                for jj in cmodes:
                    if is_par2_C(jj):
                        par2 = jj
                        AAA = ctm[jj] @ ctm[jj].T
                    else:
                        AA = AA + rho[jj] * (ctm[jj] @ ctm[jj].T)
                    sc = np.asarray(rho[jj])[:, None] if np.size(rho[jj]) > 1 else rho[jj]
                    BB = BB + (sc * (G['fac'][jj] + G['coupling_dual_fac'][jj])) @ ctm[jj].T   # :955
                if par2 is not None:
                    newD = np.array(G['coupling_fac'][cid], copy=True)
                    for kk in range(newD.shape[0]):
                        newD[kk, :] = _mrdivide(BB[kk:kk + 1, :], AA + rho[par2][kk] * AAA)[0]
                    G['coupling_fac'][cid] = newD
                else:
                    G['coupling_fac'][cid] = _mrdivide(BB, AA)
            else:                                                                  # case5 :1026-1054
                H2_1 = ctm2[cmodes[0]]
                AA = np.zeros((H2_1.shape[0], H2_1.shape[0]))
                BB = np.zeros((ctm[cmodes[0]].shape[0], H2_1.shape[0]))
                mm_last = cmodes[-1]           # QUIRK :1032 -- rhoC uses the loop variable `mm` left over
                par2 = None
                for jj in cmodes:
                    rhoC = float(np.mean(rho[mm_last]))
                    if is_par2_C(jj):                                              # :1034-1040
                        par2 = jj
                        AAA = ctm2[jj] @ ctm2[jj].T
                    else:
                        AA = AA + rhoC * (ctm2[jj] @ ctm2[jj].T)
                    BB = BB + rhoC * (ctm[jj] @ G['fac'][jj] + G['coupling_dual_fac'][jj]) @ ctm2[jj].T
                if par2 is not None:                                               # :1049-1052 (rho_k indexed by Delta's row)
                    newD = np.array(G['coupling_fac'][cid], copy=True)
                    for kk in range(newD.shape[0]):
                        newD[kk, :] = _mrdivide(BB[kk:kk + 1, :], AA + rho[par2][kk] * AAA)[0]
                    G['coupling_fac'][cid] = newD
                else:
                    G['coupling_fac'][cid] = _mrdivide(BB, AA)
            # ---- dual + constraint updates
            D = G['coupling_fac'][cid]
            for mm in cmodes:
                F = G['fac'][mm]
                H = ctm[mm]
                if ctype == 0:
                    upd = F - D                                                     # :679
                elif ctype == 1:
                    upd = H @ F - D
                elif ctype == 2:
                    upd = F @ H - D
                elif ctype == 3:
                    upd = F - H @ D
                elif ctype == 4:
                    upd = F - D @ H                                                 # :967
                else:
                    upd = H @ F - D @ ctm2[mm]                                      # :1059
                G['coupling_dual_fac'][mm] = G['coupling_dual_fac'][mm] + upd
                if constrained[mm]:
                    oldZ[mm] = update_constraint(mm, rho[mm])
            inner_iter += 1
            pc, dc = eval_res_ADMM_coupl(ctype, cmodes, cid, oldDelta)
            cm = [mm for mm in cmodes if constrained[mm]]
            if cm:
                pz, dz = eval_res_ADMM_constr(cm, oldZ)
            else:
                pz = dz = 0.0
        return inner_iter - 1

    # ---------------- nested: ADMM_B_Parafac2  :509-589
    def ADMM_B_Parafac2(Am, Lm, m, p, rho_m, it):
        K = _K_of(Z, p)
        inner_iter = 1
        pz = dz = pc = dc = inf
        oldP = [None] * K
        use_constr = bool(constrained[m]) and it >= options['iter_start_PAR2Bkconstraint']
        while _loop_cond(inner_iter, pc, pz, dc, dz):
            pz = dz = pc = dc = 0.0
            for kk in range(K):
                A_inner = Am[kk] + rho_m[kk] / 2 * (G['P'][p][kk] @ G['DeltaB'][p] - G['mu_DeltaB'][p][kk])    # :526
                if use_constr:
                    A_inner = A_inner + rho_m[kk] / 2 * (G['constraint_fac'][m][kk] - G['constraint_dual_fac'][m][kk])
                G['fac'][m][kk] = _solve_llt_right(A_inner, Lm[kk])                                           # :530
                U, _, Vt = np.linalg.svd((G['fac'][m][kk] + G['mu_DeltaB'][p][kk]) @ G['DeltaB'][p].T, full_matrices=False)
                oldP[kk] = G['P'][p][kk]
                G['P'][p][kk] = U @ Vt                                                                        # :534
            oldDeltaB = G['DeltaB'][p]
            newDB = np.zeros_like(oldDeltaB)
            sum_rho_k = 0.0
            for kk in range(K):
                newDB = newDB + rho_m[kk] * G['P'][p][kk].T @ (G['fac'][m][kk] + G['mu_DeltaB'][p][kk])     # :541
                sum_rho_k += rho_m[kk]
            G['DeltaB'][p] = newDB / sum_rho_k
            for kk in range(K):
                G['mu_DeltaB'][p][kk] = G['mu_DeltaB'][p][kk] + G['fac'][m][kk] - G['P'][p][kk] @ G['DeltaB'][p]   # :546
            if use_constr:
                oldZ = list(G['constraint_fac'][m])
                if Z['constraints'][m][0] == 'tPARAFAC2':                                                      # :553
                    G['constraint_fac'][m] = list(prox_of(m)(
                        [G['fac'][m][kk] + G['constraint_dual_fac'][m][kk] for kk in range(K)], rho_m))
                else:
                    G['constraint_fac'][m] = [prox_of(m)(G['fac'][m][kk] + G['constraint_dual_fac'][m][kk], float(rho_m[kk]))
                                              for kk in range(K)]
                for kk in range(K):
                    G['constraint_dual_fac'][m][kk] = G['constraint_dual_fac'][m][kk] + G['fac'][m][kk] - G['constraint_fac'][m][kk]
                    pz += _fro(G['fac'][m][kk] - G['constraint_fac'][m][kk]) / _fro(G['fac'][m][kk]) / K
                    scaling = _fro(G['constraint_dual_fac'][m][kk])
                    d = _fro(oldZ[kk] - G['constraint_fac'][m][kk])
                    dz += (d / scaling if scaling > 0 else d) / K
            for kk in range(K):                                                                                # :582-585
                PD = G['P'][p][kk] @ G['DeltaB'][p]
                pc += _fro(G['fac'][m][kk] - PD) / _fro(G['fac'][m][kk]) / K
                dc += _fro(oldP[kk] @ oldDeltaB - PD) / _fro(G['mu_DeltaB'][p][kk]) / K
            inner_iter += 1
        return inner_iter - 1

    # ---------------- nested: CMTF_AOADMM_func_eval  :1213-1363
    def func_eval(first):
        fp = np.zeros(P)
        for pp in range(P):
            md = _modes0(Z, pp)
            if Z['model'][pp] == 'CP' and miss[pp] is not None:              # :1224-1226
                Mfull = full_ktensor([G['fac'][m] for m in md])
                M = np.where(np.asarray(miss[pp], dtype=bool), Mfull, 0.0)
                fp[pp] = weights[pp] * (Znorm_const[pp] - 2 * float(np.sum(np.asarray(Z['object'][pp]) * M)) + float(np.sum(M * M)))
            elif Z['model'][pp] == 'CP':
                if first:                                                    # :1228-1233
                    facs = [G['fac'][m] for m in md]
                    if _is_tensor(Z['object'][pp]):
                        fp[pp] = cp_func(Z['object'][pp], facs, Znorm_const[pp], weights[pp])
                    else:
                        fp[pp] = pca_func(np.asarray(Z['object'][pp]), facs, Znorm_const[pp], weights[pp])
                else:                                                        # :1235-1241
                    lm = md[last_m[pp]]
                    f_2 = np.sum(last_mttkrp[pp] * G['fac'][lm])
                    f_3 = np.sum(last_had[pp] * G_transp_G[lm])
                    fp[pp] = weights[pp] * (Znorm_const[pp] - 2 * f_2 + f_3)
            elif miss[pp] is not None:                                       # PAR2 with masks :1249-1252
                for kk in range(_K_of(Z, pp)):
                    Mk = G['fac'][md[0]] @ np.diag(G['fac'][md[2]][kk, :]) @ G['fac'][md[1]][kk].T
                    fp[pp] += _fro(np.where(np.asarray(miss[pp][kk], dtype=bool), Z['object'][pp][kk] - Mk, 0.0)) ** 2
                fp[pp] = weights[pp] * fp[pp]
            else:                                                            # PAR2 :1254-1268
                if (not first) and last_m[pp] == 0:
                    f_2 = np.sum(last_mttkrp[pp] * G['fac'][md[0]])
                    f_3 = np.sum(last_had[pp] * G_transp_G[md[0]])
                    fp[pp] = Znorm_const[pp] - 2 * f_2 + f_3
                else:
                    for kk in range(_K_of(Z, pp)):
                        Mk = G['fac'][md[0]] @ np.diag(G['fac'][md[2]][kk, :]) @ G['fac'][md[1]][kk].T
                        fp[pp] += _fro(Z['object'][pp][kk] - Mk) ** 2
                fp[pp] = weights[pp] * fp[pp]
        f_tensors = float(np.sum(fp))
        reg = Z.get('reg_func')
        if reg is not None:                                                  # :1272-1288
            for n in range(nb_modes):
                if reg[n] is not None:
                    if isinstance(G['constraint_fac'][n], list):
                        if Z['constraints'][n][0] == 'tPARAFAC2':
                            f_tensors += float(reg[n](G['fac'][n]))
                        else:
                            for kk in range(len(G['constraint_fac'][n])):
                                f_tensors += float(reg[n](G['fac'][n][kk]))
                    else:
                        f_tensors += float(reg[n](G['fac'][n]))
        if has_ridge:                                                        # :1290-1300
            for n in range(nb_modes):
                if isinstance(G['fac'][n], list):
                    cf = G['constraint_fac'][n]
                    for kk in range(len(cf) if cf is not None else 0):       # QUIRK :1293
                        f_tensors += Z['ridge'][n] * _fro(G['fac'][n][kk]) ** 2
                else:
                    f_tensors += Z['ridge'][n] * _fro(G['fac'][n]) ** 2
        # couplings :1303-1329
        nb_couplings = max(lin) if lin else 0
        coupling_p = np.zeros(nb_couplings)
        for n in range(nb_couplings):
            ct = int(Z['coupling']['coupling_type'][n])
            cm = [i for i, v in enumerate(lin) if v == n + 1]
            D = G['coupling_fac'][n]
            for j in cm:
                F, H = G['fac'][j], ctm[j]
                if ct == 0:
                    coupling_p[n] += _fro(F - D) / _fro(F)
                elif ct == 1:
                    coupling_p[n] += _fro(H @ F - D) / _fro(H @ F)
                elif ct == 2:
                    coupling_p[n] += _fro(F @ H - D) / _fro(F @ H)
                elif ct == 3:
                    coupling_p[n] += _fro(F - H @ D) / _fro(F)
                elif ct == 4:
                    coupling_p[n] += _fro(F - D @ H) / _fro(F)
                else:
                    coupling_p[n] += _fro(H @ F - D @ ctm2[j]) / _fro(H @ F)
        f_couplings = float(np.sum(coupling_p))
        if f_couplings > 0:
            f_couplings /= np.count_nonzero(coupling_p)
        # constraints :1332-1348
        fc = np.zeros(nb_modes)
        for n in range(nb_modes):
            cf = G['constraint_fac'][n]
            if cf is None or (isinstance(cf, list) and len(cf) == 0):
                continue
            if isinstance(cf, list):
                for kk in range(len(cf)):
                    fc[n] += _fro(G['fac'][n][kk] - cf[kk]) / _fro(G['fac'][n][kk])
                fc[n] /= len(cf)
            else:
                fc[n] = _fro(G['fac'][n] - cf) / _fro(G['fac'][n])
        f_constraints = float(np.sum(fc))
        if f_constraints > 0:
            f_constraints /= np.count_nonzero(fc)
        # PAR2 internal couplings :1351-1362
        fpar = np.zeros(P)
        for pp in range(P):
            if Z['model'][pp] == 'PAR2':
                md = _modes0(Z, pp)
                for kk in range(_K_of(Z, pp)):
                    Bk = G['fac'][md[1]][kk]
                    fpar[pp] += _fro(Bk - G['P'][pp][kk] @ G['DeltaB'][pp]) / _fro(Bk)
        f_par2 = float(np.sum(fpar))
        if f_par2 > 0:
            f_par2 /= _K_of(Z, P - 1) if Z['model'][P - 1] == 'PAR2' else 1   # QUIRK :1361 uses last pp
        return f_tensors, f_couplings, f_constraints, f_par2

    # ================= main body =================
    f = func_eval(first=True)                                                # :32
    func_val = [f[0]]; func_coupl = [f[1]]; func_constr = [f[2]]; func_par2 = [f[3]]
    func_rel_missing = [f_rel_missing]                                       # :37-39
    tstart = time.perf_counter()
    time_at_it = [0.0]
    it = 1

    for m in range(nb_modes):                                                # :62-81
        p = which_p[m]
        if Z['model'][p] == 'CP':
            G_transp_G[m] = G['fac'][m].T @ G['fac'][m]
        else:
            pos = _modes0(Z, p).index(m)
            if pos == 0:
                G_transp_G[m] = G['fac'][m].T @ G['fac'][m]
            elif pos == 1:
                G_transp_G[m] = [Bk.T @ Bk for Bk in G['fac'][m]]

    stop = False
    while it <= options['MaxOuterIters'] and not stop:                        # :87
        for coupl_id in couplings:                                            # :89
            coupled_modes = [i for i, v in enumerate(lin) if v == coupl_id]
            for p in sorted(set(which_p[i] for i in coupled_modes)):          # :91
                md = _modes0(Z, p)
                my_modes = [m for m in coupled_modes if which_p[m] == p]
                w = weights[p]
                if Z['model'][p] == 'CP':
                    for m in my_modes:                                        # :93
                        pos = md.index(m)
                        obj = Z['object'][p]
                        if _is_tensor(obj):                                   # :96-103
                            A[m] = w * mttkrp(obj, [G['fac'][j] for j in md], pos)
                            C[m] = np.ones_like(G_transp_G[m])
                            for j in md:
                                if j != m:
                                    C[m] = C[m] * G_transp_G[j]
                        else:                                                 # :105-114
                            if pos == 0:
                                A[m] = w * (np.asarray(obj) @ G['fac'][md[1]])
                                C[m] = G_transp_G[md[1]]
                            else:
                                A[m] = w * (np.asarray(obj).T @ G['fac'][md[0]])
                                C[m] = G_transp_G[md[0]]
                        R = C[m].shape[0]
                        rho[m] = float(np.trace(C[m]) / R)                    # :115
                        B[m] = w * C[m]                                       # :116
                        if has_ridge:
                            B[m] = B[m] + Z['ridge'][m] * np.eye(R)           # :117-119
                        last_mttkrp[p] = A[m] * 1 / w                         # :121
                        last_had[p] = C[m]
                        last_m[p] = pos
                        if options['bsum']:                                   # :124-127
                            A[m] = A[m] + options['bsum_weight'] / 2 * G['fac'][m]
                            B[m] = B[m] + options['bsum_weight'] / 2 * np.eye(R)
                        if coupl_id == 0:                                     # :131-155
                            if not constrained[m]:
                                G['fac'][m] = _mrdivide(A[m], B[m])           # :134
                                inner_iters = 1
                            else:
                                B[m] = B[m] + rho[m] / 2 * np.eye(R)          # :141
                                L[m] = _chol_lower(B[m])                      # :142
                                inner_iters = ADMM_constrained_only(A[m], L[m], m, p)
                            innerIters[(m, it)] = inner_iters
                            G_transp_G[m] = G['fac'][m].T @ G['fac'][m]       # :148
                else:  # PAR2  :157-250
                    K = _K_of(Z, p)
                    mA, mB, mC = md
                    for m in my_modes:
                        pos = md.index(m)
                        if pos == 0:                                          # :159-190
                            R = G['fac'][m].shape[1]
                            A[m] = np.zeros_like(G['fac'][m])
                            C[m] = np.zeros((R, R))
                            for k in range(K):
                                Dk = np.diag(G['fac'][mC][k, :])
                                A[m] = A[m] + Z['object'][p][k] @ G['fac'][mB][k] @ Dk
                                C[m] = C[m] + Dk @ G_transp_G[mB][k] @ Dk
                            last_had[p] = C[m]
                            last_mttkrp[p] = A[m]
                            last_m[p] = 0
                            A[m] = w * A[m]
                            rho[m] = float(np.trace(C[m]) / R)
                            B[m] = w * C[m]
                            if has_ridge:
                                B[m] = B[m] + Z['ridge'][m] * np.eye(R)
                            if options['bsum']:
                                A[m] = A[m] + options['bsum_weight'] / 2 * G['fac'][m]
                                B[m] = B[m] + options['bsum_weight'] / 2 * np.eye(R)
                            if coupl_id == 0:
                                if not constrained[m]:
                                    G['fac'][m] = _mrdivide(A[m], B[m])
                                    inner_iters = 1
                                else:
                                    B[m] = B[m] + rho[m] / 2 * np.eye(R)
                                    L[m] = _chol_lower(B[m])
                                    inner_iters = ADMM_constrained_only(A[m], L[m], m, p)
                                innerIters[(m, it)] = inner_iters
                            G_transp_G[m] = G['fac'][m].T @ G['fac'][m]       # :190
                        elif pos == 1:                                        # :191-218
                            R = G['fac'][mA].shape[1]
                            A[m] = [None] * K; C[m] = [None] * K; B[m] = [None] * K; L[m] = [None] * K
                            rho[m] = np.zeros(K)
                            for k in range(K):
                                Dk = np.diag(G['fac'][mC][k, :])
                                A[m][k] = w * Z['object'][p][k].T @ G['fac'][mA] @ Dk          # :193
                                C[m][k] = Dk @ G_transp_G[mA] @ Dk                              # :194
                                rho[m][k] = np.trace(C[m][k]) / R
                                if 'increase_factor_rhoBk' in options:
                                    rho[m][k] = options['increase_factor_rhoBk'] * rho[m][k]
                                B[m][k] = w * C[m][k]
                                B[m][k] = B[m][k] + rho[m][k] / 2 * np.eye(R)                   # :200
                                if has_ridge:
                                    B[m][k] = B[m][k] + Z['ridge'][m] * np.eye(R)
                                if options['bsum']:
                                    A[m][k] = A[m][k] + options['bsum_weight'] / 2 * G['fac'][m][k]
                                    B[m][k] = B[m][k] + options['bsum_weight'] / 2 * np.eye(R)
                                last_m[p] = 1
                                if constrained[m] and it >= options['iter_start_PAR2Bkconstraint']:
                                    B[m][k] = B[m][k] + rho[m][k] / 2 * np.eye(R)               # :210
                                L[m][k] = np.linalg.cholesky(B[m][k])                           # :212
                            innerIters[(m, it)] = ADMM_B_Parafac2(A[m], L[m], m, p, rho[m], it)
                            G_transp_G[m] = [Bk.T @ Bk for Bk in G['fac'][m]]
                        else:                                                 # :219-248
                            R = G['fac'][mA].shape[1]
                            A[m] = [None] * K; C[m] = [None] * K; B[m] = [None] * K; L[m] = [None] * K
                            rho[m] = np.zeros(K)
                            for k in range(K):
                                A[m][k] = w * np.diag(G['fac'][mA].T @ Z['object'][p][k] @ G['fac'][mB][k])   # :221
                                C[m][k] = G_transp_G[mA] * G_transp_G[mB][k]
                                rho[m][k] = np.trace(C[m][k]) / R
                                B[m][k] = w * C[m][k]
                                if has_ridge:
                                    B[m][k] = B[m][k] + Z['ridge'][m] * np.eye(R)
                                last_m[p] = 2
                                if options['bsum']:
                                    A[m][k] = A[m][k] + options['bsum_weight'] / 2 * G['fac'][m][k, :]
                                    B[m][k] = B[m][k] + options['bsum_weight'] / 2 * np.eye(R)
                                if coupl_id == 0:
                                    if not constrained[m]:
                                        G['fac'][m][k, :] = np.linalg.solve(B[m][k], A[m][k])   # :236
                                        inner_iters = 1
                                    else:
                                        B[m][k] = B[m][k] + rho[m][k] / 2 * np.eye(R)
                                        L[m][k] = _chol_lower(B[m][k])
                            if constrained[m] and coupl_id == 0:
                                inner_iters = ADMM_constrained_only(A[m], L[m], m, p)
                            if coupl_id == 0:
                                innerIters[(m, it)] = inner_iters
            if coupl_id != 0:                                                 # :253-404
                ctype = int(Z['coupling']['coupling_type'][coupl_id - 1])
                for m in coupled_modes:
                    R_ = None
                    if is_par2_C(m) and ctype in (1, 5):                       # :282-297, :371-385
                        Kc = _K_of(Z, which_p[m])
                        R_ = B[m][0].shape[0]
                        HcI = np.kron(ctm[m], np.eye(R_))                                          # :283
                        B2[m] = float(np.mean(rho[m])) / 2 * (HcI.T @ HcI)                          # :284
                        B2[m] = sla.block_diag(*[B[m][k] for k in range(Kc)]) + B2[m]               # :286
                        if constrained[m]:
                            B2[m] = B2[m] + float(np.mean(rho[m])) / 2 * np.eye(B2[m].shape[0])    # :292
                        L[m] = np.linalg.cholesky(B2[m])                                           # :296
                    elif is_par2_C(m):
                        for k in range(_K_of(Z, which_p[m])):
                            R_ = B[m][k].shape[0]
                            if ctype == 2:
                                B[m][k] = B[m][k] + rho[m][k] / 2 * (ctm[m] @ ctm[m].T)          # :307
                            else:
                                B[m][k] = B[m][k] + rho[m][k] / 2 * np.eye(R_)                   # :262,:329,:351
                            if constrained[m]:
                                B[m][k] = B[m][k] + rho[m][k] / 2 * np.eye(R_)
                            L[m][k] = _chol_lower(B[m][k])
                    else:
                        R_ = B[m].shape[0]
                        if ctype in (0, 3, 4):
                            B[m] = B[m] + rho[m] / 2 * np.eye(R_)                                # :269,:336,:358
                            if constrained[m]:
                                B[m] = B[m] + rho[m] / 2 * np.eye(R_)
                            L[m] = _chol_lower(B[m])
                        elif ctype == 2:
                            B[m] = B[m] + rho[m] / 2 * (ctm[m] @ ctm[m].T)                       # :314
                            if constrained[m]:
                                B[m] = B[m] + rho[m] / 2 * np.eye(R_)
                            L[m] = _chol_lower(B[m])
                        else:  # 1, 5  (:288-293, :377-382)
                            B2[m] = rho[m] / 2 * (ctm[m].T @ ctm[m])
                            if constrained[m]:
                                B2[m] = B2[m] + np.mean(rho[m]) / 2 * np.eye(B2[m].shape[0])
                inner_iters = ADMM_coupled(ctype, coupled_modes, coupl_id - 1)
                for m in coupled_modes:
                    innerIters[(m, it)] = inner_iters                          # :392
                    G_transp_G[m] = G['fac'][m].T @ G['fac'][m]               # :396

        # EM imputation (:408-441): missing entries <- current model; relative change of the imputed values
        if has_missing:
            num_sq = den_sq = 0.0
            for p in range(P):
                if miss[p] is None:
                    continue
                md = _modes0(Z, p)
                if Z['model'][p] == 'CP':
                    M_full = full_ktensor([G['fac'][m] for m in md])
                    mm = ~np.asarray(miss[p], dtype=bool)
                    old = Z['object'][p][mm]
                    new = M_full[mm]
                    Z['object'][p][mm] = new
                    num_sq += float(np.sum((new - old) ** 2))
                    den_sq += float(np.sum(old ** 2))
                else:
                    for k in range(_K_of(Z, p)):
                        M_k = G['fac'][md[0]] @ np.diag(G['fac'][md[2]][k, :]) @ G['fac'][md[1]][k].T
                        mk = ~np.asarray(miss[p][k], dtype=bool)
                        old = Z['object'][p][k][mk]
                        new = M_k[mk]
                        num_sq += float(np.sum((new - old) ** 2))
                        den_sq += float(np.sum(old ** 2))
                        Z['object'][p][k][mk] = new
            f_rel_missing = float(np.sqrt(num_sq / den_sq)) if den_sq > 0 else float(np.sqrt(num_sq))   # :436-440
        f_old = f
        f = func_eval(first=False)                                            # :447
        func_val.append(f[0]); func_coupl.append(f[1]); func_constr.append(f[2]); func_par2.append(f[3])
        func_rel_missing.append(f_rel_missing)
        time_at_it.append(time.perf_counter() - tstart)
        stop = evaluate_stopping_conditions(f, f_old, options)                 # :456
        if has_missing:
            stop = stop and (f_rel_missing < options['OuterRelTol'])           # :457-459
        if trace is not None:
            trace.setdefault('fac', []).append(copy.deepcopy(G['fac']))
        it += 1

    out = {
        'f_tensors': f[0], 'f_couplings': f[1], 'f_constraints': f[2], 'f_PAR2_couplings': f[3],
        'f_rel_missing': f_rel_missing,
        'exit_flag': make_exit_flag(it, f, options),
        'OuterIterations': it - 1,
        'func_val_conv': np.array(func_val), 'func_coupl_conv': np.array(func_coupl),
        'func_constr_conv': np.array(func_constr), 'func_PAR2_coupl': np.array(func_par2),
        'time_at_it': np.array(time_at_it),
    }
    inner = np.zeros((nb_modes, max(it - 1, 1)))
    for (m, i), v in innerIters.items():
        inner[m, i - 1] = v
    out['innerIters'] = inner
    if has_missing:
        out['func_rel_missing'] = np.array(func_rel_missing)                  # :490-492
    return G, out


# --------------------------------------------------------------------------
# driver + initialiser
# --------------------------------------------------------------------------

def compute_Znorm_const(Z):
    """functions/cmtf_AOADMM.m:124-156 (Frobenius): ||X||^2, or ||miss .* X||^2 for a block with a mask."""
    out = []
    miss = Z.get('miss') if Z.get('miss') is not None else [None] * len(Z['object'])
    for p in range(len(Z['object'])):
        if Z['model'][p] == 'CP':
            if miss[p] is not None:
                out.append(float(np.sum(np.where(np.asarray(miss[p], dtype=bool), np.asarray(Z['object'][p]), 0.0) ** 2)))
            else:
                out.append(tensor_norm(Z['object'][p]) ** 2)
        else:
            if miss[p] is not None:
                out.append(float(sum(np.sum(np.where(np.asarray(mk, dtype=bool), Xk, 0.0) ** 2)
                                     for Xk, mk in zip(Z['object'][p], miss[p]))))
            else:
                out.append(float(sum(np.linalg.norm(Xk, 'fro') ** 2 for Xk in Z['object'][p])))
    return out


def cmtf_AOADMM(Z, alg_options=None, init='random', init_options=None, rng=None):
    """functions/cmtf_AOADMM.m:1-206.  Returns (Zhat, Fac, G, out)."""
    Z = dict(Z)
    which_p = _which_p(Z)
    prox_ops, reg = _prox.constraints_to_prox(Z['constrained_modes'], Z['constraints'], Z['size'])   # :30
    Z['prox_operators'] = prox_ops
    Z['reg_func'] = reg
    for m, c in enumerate(Z['constraints']):                                                      # :33-41
        if Z['constrained_modes'][m] and c[0] == 'tPARAFAC2':
            p = which_p[m]
            if Z['model'][p] != 'PAR2' or _modes0(Z, p).index(m) != 1:
                raise ValueError('The tPARAFAC2 constraint can only be impsed on the second mode of a PARAFAC2 model')
    if isinstance(init, dict):                                                                    # :44-53
        G = init
    elif isinstance(init, str) and init.lower() == 'random':
        if init_options is None:
            raise ValueError('init_options are missing as input in cmtf_AOADMM.')
        G = init_coupled_AOADMM_CMTF(Z, init_options=init_options, rng=rng)
    else:
        raise ValueError('Initialization type not supported')
    for p in range(len(Z['object'])):                                                             # :55-65
        if Z['model'][p] == 'PAR2':
            md = _modes0(Z, p)
            R = G['fac'][md[0]].shape[1]
            for k, jk in enumerate(Z['size'][md[1]]):
                if jk < R:
                    raise ValueError('Number of components for PARAFAC2 is larger than size of slice %d of data tensor %d.' % (k + 1, p + 1))
    Znorm_const = compute_Znorm_const(Z)
    Fac, out = cmtf_fun_AOADMM(Z, Znorm_const, G, alg_options)                                    # :193
    Zhat = []
    for p in range(len(Z['object'])):                                                             # :197-206
        md = _modes0(Z, p)
        if Z['model'][p] == 'CP':
            Zhat.append([Fac['fac'][m] for m in md])
        else:
            Zhat.append({'A': Fac['fac'][md[0]], 'Bk': Fac['fac'][md[1]], 'C': Fac['fac'][md[2]]})
    return Zhat, Fac, G, out


def _leading_eigvecs(Y, r):
    """`[U,~] = eigs(Y, r, 'LM')` (cmtf_nvecs.m:58): eigenvectors of the r largest-magnitude eigenvalues; signs are
    arbitrary (ARPACK's are too), so comparisons use the spanned subspace."""
    w, V = np.linalg.eigh((Y + Y.T) / 2)
    idx = np.argsort(-np.abs(w), kind='stable')[:r]
    return V[:, idx]


def cmtf_nvecs(Z, n, r):
    """functions/cmtf_nvecs.m:1-58 (dense data): A = mode-n unfolding of the data set owning mode n, Y = A*A'."""
    which_p = _which_p(Z)
    p = which_p[n]
    X = np.asarray(Z['object'][p], dtype=np.float64)
    i = _modes0(Z, p).index(n)
    A = np.moveaxis(X, i, 0).reshape(X.shape[i], -1)
    return _leading_eigvecs(A @ A.T, r)


def init_coupled_AOADMM_CMTF(Z, init_options, Delta=None, rng=None):
    """functions/init_coupled_AOADMM_CMTF.m:1-174 (random path, and the SVD-based path `nvecs = 1`, :50-73).

    `init_options['distr'][n]` is a callable `(rows, cols) -> ndarray`;
    MATLAB's `rand(...)` calls are drawn from `rng.random(...)` (numpy Generator).
    RNG streams cannot match MATLAB's; parity tests pass the resulting struct
    explicitly as `init` to both implementations.
    """
    if rng is None:
        rng = np.random.default_rng(0)
    nvecs = bool(init_options.get('nvecs', 0))
    sz = Z['size']
    lambdas = init_options['lambdas_init']
    distr = init_options['distr']
    normalize = init_options['normalize']
    nb_modes = len(sz)
    lin = [int(v) for v in Z['coupling']['lin_coupled_modes']]
    nb_couplings = max(lin) if lin else 0
    ctm = Z['coupling'].get('coupl_trafo_matrices', [None] * nb_modes)
    P = len(Z['modes'])
    A = {'fac': [None] * nb_modes, 'coupling_fac': [None] * nb_couplings,
         'constraint_fac': [None] * nb_modes, 'coupling_dual_fac': [None] * nb_modes,
         'constraint_dual_fac': [None] * nb_modes, 'DeltaB': {}, 'P': {}, 'mu_DeltaB': {}}

    def colnorm(M):
        return M / np.sqrt(np.sum(M * M, axis=0))[None, :]

    for p in range(P):                                                          # :48-97
        md = _modes0(Z, p)
        R = len(lambdas[p])
        for n in md:
            if nvecs:                                                           # :50-73
                if Z['model'][p] == 'CP':
                    A['fac'][n] = cmtf_nvecs(Z, n, R)
                elif md.index(n) == 0:
                    M = np.hstack([np.asarray(Xk, dtype=np.float64) for Xk in Z['object'][p]])
                    A['fac'][n] = _leading_eigvecs(M @ M.T, R)
                elif md.index(n) == 1:
                    A['DeltaB'][p] = rng.random((R, R))
                    A['fac'][n] = []; A['P'][p] = []; A['mu_DeltaB'][p] = []
                    for k in range(len(sz[n])):
                        Mk = np.asarray(Z['object'][p][k], dtype=np.float64).T
                        A['fac'][n].append(_leading_eigvecs(Mk @ Mk.T, R))
                        A['P'][p].append(np.eye(sz[n][k], R))
                        A['mu_DeltaB'][p].append(rng.random((sz[n][k], R)))
                else:
                    A['fac'][n] = np.ones((sz[n], R))
                continue
            if Z['model'][p] == 'PAR2' and md.index(n) == 1:                    # :75-86
                A['DeltaB'][p] = rng.random((R, R))
                A['fac'][n] = []; A['P'][p] = []; A['mu_DeltaB'][p] = []
                for k in range(len(sz[n])):
                    F = np.asarray(distr[n](sz[n][k], R), dtype=np.float64)
                    A['P'][p].append(np.eye(sz[n][k], R))
                    A['mu_DeltaB'][p].append(rng.random((sz[n][k], R)))
                    A['fac'][n].append(colnorm(F) if normalize else F)
            else:                                                               # :87-94
                F = np.asarray(distr[n](sz[n], R), dtype=np.float64)
                A['fac'][n] = colnorm(F) if normalize else F
    if any(Z['constrained_modes']):                                             # :99-129
        prox_ops, _ = _prox.constraints_to_prox(Z['constrained_modes'], Z['constraints'], sz)
        for p in range(P):
            md = _modes0(Z, p)
            for n in md:
                if not Z['constrained_modes'][n]:
                    continue
                if Z['model'][p] == 'PAR2' and md.index(n) == 1:
                    A['constraint_fac'][n] = []; A['constraint_dual_fac'][n] = []
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
                    A['constraint_fac'][n] = prox_ops[n](Zn, 1.0)               # :123
                    A['constraint_dual_fac'][n] = rng.random(A['fac'][n].shape)
    for n in range(nb_couplings):                                               # :133-169
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
