"""Thin object wrapper over the C ABI context plus the op-level calls.

Mirrors the L2->L1 calls of the reference solver
(`functions/cmtf_fun_AOADMM.m:97` mttkrp, `:66` Gram, `:142` chol,
`functions/constraints_to_prox.m` prox handles, `:591-623` ADMM inner loop).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi as capi

# constraint names of "List of constraints and regularizations.txt" -> ids of include/aoadmm_hip.h
CONSTRAINT_IDS = {
    'non-negativity': 1, 'box': 2, 'simplex column-wise': 3, 'simplex row-wise': 4, 'non-decreasing': 5,
    'non-increasing': 6, 'unimodality': 7, 'l1-ball': 8, 'l2-ball': 9, 'non-negative l2-ball': 10,
    'non-negative l2-sphere': 11, 'orthonormal': 12, 'l1 regularization': 13, 'l0 regularization': 14,
    'l2 regularization': 15, 'ridge': 16, 'quadratic regularization': 17, 'GL smoothness': 18,
    'TV regularization': 19, 'tPARAFAC2': 20,
}


def constraint_descriptor(c):
    """`Z.constraints{m}` cell -> (id, params, Lmat) (constraints_to_prox.m:13-91)."""
    if c is None or len(c) == 0:
        raise ValueError('No constraint provided')
    name = c[0]
    if name == 'custom':
        raise capi.UnsupportedOnDevice(capi.ERR_UNSUPPORTED,
                                       "'custom' prox handles cannot cross to the device (constraints_to_prox.m:86-90)")
    if name not in CONSTRAINT_IDS:
        raise ValueError('unknown constraint %r' % (name,))
    cid = CONSTRAINT_IDS[name]
    params = []
    Lmat = None
    if name == 'quadratic regularization':
        params = [float(c[1])]
        Lmat = capi.as_f(c[2])
    elif name == 'unimodality':
        params = [1.0 if c[1] else 0.0]
    else:
        params = [float(v) for v in c[1:]]
    return cid, np.asarray(params, dtype=np.float64), Lmat


class Engine:
    """One `aoadmm_ctx`: one GPU (`Engine(0)`), or several GPUs driven from this one process (`Engine([0, 1, 2, 3])`,
    aoadmm_create_multi: one engine + one host thread per device inside the library, RCCL between them).
    Raises when no GPU / library is available."""

    def __init__(self, device=0):
        self.lib = capi.load_library()
        self.h = C.c_void_p()
        if isinstance(device, (list, tuple)):
            devs = (C.c_int * len(device))(*[int(d) for d in device])
            capi.check(self.lib.aoadmm_create_multi(C.byref(self.h), len(device), devs))
        else:
            capi.check(self.lib.aoadmm_create(C.byref(self.h), int(device)))

    def close(self):
        if getattr(self, 'h', None) is not None and self.h:
            self.lib.aoadmm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def synchronize(self):
        capi.check(self.lib.aoadmm_synchronize(self.h))

    # ---- communicator ---------------------------------------------------------
    def comm_unique_id(self):
        buf = C.create_string_buffer(128)
        capi.check(self.lib.aoadmm_comm_unique_id(buf))
        return buf.raw

    def comm_init_rank(self, uid, rank, world):
        capi.check(self.lib.aoadmm_comm_init_rank(self.h, uid, int(rank), int(world)))

    def comm_init_rank_share(self, uid, rank, world):
        """Measurement hook: rank `rank` of `world` in every sharding decision, on a ONE-rank RCCL communicator."""
        capi.check(self.lib.aoadmm_comm_init_rank_share(self.h, uid, int(rank), int(world)))

    def comm_rank(self):
        """(rank, world) of this engine's communicator; (0, 1) without one."""
        r, w = C.c_int(0), C.c_int(1)
        capi.check(self.lib.aoadmm_comm_rank(self.h, C.byref(r), C.byref(w)))
        return r.value, w.value

    def comm_info(self):
        """{'rccl_version', 'comm_ranks', 'librccl'}: what the collectives of this context run on."""
        v, n = C.c_int(0), C.c_int(0)
        buf = C.create_string_buffer(1024)
        capi.check(self.lib.aoadmm_comm_info(self.h, C.byref(v), C.byref(n), buf, 1024))
        return {'rccl_version': v.value, 'comm_ranks': n.value, 'librccl': buf.value.decode('utf-8', 'replace')}

    def comm_init_local(self, key, rank, world):
        """Bring-up/test transport: engines driven by threads of this process form group `key` (see aoadmm_hip.h)."""
        capi.check(self.lib.aoadmm_comm_init_local(self.h, int(key), int(rank), int(world)))

    # ---- op level -----------------------------------------------------------------
    def mttkrp(self, X, U, n, precision='f64'):
        """`mttkrp(X,U,n)` with 0-based n (cmtf_fun_AOADMM.m:97)."""
        X = capi.as_f(X)
        dims = (C.c_int64 * X.ndim)(*X.shape)
        Us = [capi.as_f(u) for u in U]
        R = Us[0].shape[1]
        arr = (C.POINTER(C.c_double) * X.ndim)(*[capi.dptr(u) for u in Us])
        out = np.zeros((X.shape[n], R), order='F')
        prec = capi.PREC_F32 if precision == 'f32' else capi.PREC_F64
        capi.check(self.lib.aoadmm_op_mttkrp(self.h, capi.dptr(X), X.ndim, dims, arr, R, int(n), prec, capi.dptr(out)))
        return out

    def unfold_gram(self, X, n, precision='f64'):
        """`Y = A*A'` with A the mode-n unfolding of X (0-based n), cmtf_nvecs.m:40-56."""
        X = capi.as_f(X)
        dims = (C.c_int64 * X.ndim)(*X.shape)
        out = np.zeros((X.shape[n], X.shape[n]), order='F')
        prec = capi.PREC_F32 if precision == 'f32' else capi.PREC_F64
        capi.check(self.lib.aoadmm_op_unfold_gram(self.h, capi.dptr(X), X.ndim, dims, int(n), prec, capi.dptr(out)))
        return out

    def resident_unfold_gram(self, p, tensor_mode, n, slab=0):
        """The same Gram matrix (n x n) from the data of tensor p already on the device (`aoadmm_resident_unfold_gram`)."""
        out = np.zeros((n, n), order='F')
        capi.check(self.lib.aoadmm_resident_unfold_gram(self.h, int(p), int(tensor_mode), int(slab), capi.dptr(out)))
        return out

    def gram(self, F):
        F = capi.as_f(F)
        out = np.zeros((F.shape[1], F.shape[1]), order='F')
        capi.check(self.lib.aoadmm_op_gram(self.h, capi.dptr(F), F.shape[0], F.shape[1], capi.dptr(out)))
        return out

    def chol(self, B):
        B = capi.as_f(B)
        out = np.zeros_like(B, order='F')
        capi.check(self.lib.aoadmm_op_chol(self.h, capi.dptr(B), B.shape[0], capi.dptr(out)))
        return out

    def prox(self, constraint, X, rho):
        """Evaluate the prox handle `constraints_to_prox` builds for `constraint` at (X, rho)."""
        cid, params, Lmat = constraint_descriptor(constraint)
        X = capi.as_f(X)
        out = np.zeros_like(X, order='F')
        capi.check(self.lib.aoadmm_op_prox(self.h, cid, capi.dptr(params) if params.size else None, params.size,
                                           capi.dptr(Lmat) if Lmat is not None else None, capi.dptr(X),
                                           X.shape[0], X.shape[1], float(rho), capi.dptr(out)))
        return out

    def admm_constrained(self, A, Bsys, rho, constraint, fac, Z, mu, max_inner, tol_pr, tol_du):
        """ADMM_constrained_only (cmtf_fun_AOADMM.m:591-623); returns (fac, Z, mu, inner_iters)."""
        cid, params, Lmat = constraint_descriptor(constraint)
        A = capi.as_f(A)
        Bsys = capi.as_f(Bsys)
        fac = capi.as_f(fac).copy(order='F')
        Z = capi.as_f(Z).copy(order='F')
        mu = capi.as_f(mu).copy(order='F')
        it = C.c_int(0)
        capi.check(self.lib.aoadmm_op_admm_constrained(
            self.h, capi.dptr(A), capi.dptr(Bsys), float(rho), cid, capi.dptr(params) if params.size else None,
            params.size, capi.dptr(Lmat) if Lmat is not None else None, A.shape[0], A.shape[1], int(max_inner),
            float(tol_pr), float(tol_du), capi.dptr(fac), capi.dptr(Z), capi.dptr(mu), C.byref(it)))
        return fac, Z, mu, it.value


_default = None


def default_engine():
    """Process-wide engine on the GPU LOCAL_RANK points at (device 0 otherwise)."""
    global _default
    if _default is None:
        import os
        _default = Engine(int(os.environ.get('LOCAL_RANK', '0')))
    return _default
