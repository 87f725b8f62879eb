"""Oracle (test infrastructure): dense tensor primitives the reference takes from
the MATLAB Tensor Toolbox v3.1 (absent from /root/reference; "parity unpinned",
see oracle/__init__.py).  Restated from the published definitions
(Kolda & Bader, SIAM Review 51(3), 2009) and anchored on the call sites
`functions/cmtf_fun_AOADMM.m:97`, `functions/cp_func.m:47`,
`functions/cmtf_AOADMM.m:136,200`.

Tensors are numpy arrays in *Fortran* order semantics: element (i1,i2,...,iN)
of MATLAB's column-major array is `X[i1,i2,...,iN]`; unfolding uses
`order='F'` so that results match MATLAB bit-for-bit in layout.
"""
from __future__ import annotations

import numpy as np


def khatrirao(mats):
    """Column-wise Kronecker product, first matrix's row index varying FASTEST
    when `mats` is given in the order (U_1, U_2, ...): row index = i1 + I1*i2 + ...
    (this is khatrirao(U_N,...,U_1) in Tensor Toolbox notation)."""
    R = mats[0].shape[1]
    out = mats[0]
    for M in mats[1:]:
        # new row index = old + rows_old * i_new
        out = (M[:, None, :] * out[None, :, :]).reshape(-1, R)
    return out


def mttkrp(X, U, n):
    """`mttkrp(X,U,n)` (0-based n): X_(n) * khatrirao(U_N..U_{n+1},U_{n-1}..U_1).

    Same formulation as the toolbox (unfold -> Khatri-Rao -> one GEMM).
    """
    X = np.asarray(X, dtype=np.float64)
    N = X.ndim
    dims = X.shape
    others = [m for m in range(N) if m != n]
    Xn = np.reshape(np.moveaxis(X, n, 0), (dims[n], -1), order='F')
    KR = khatrirao([U[m] for m in others])
    return Xn @ KR


def mttkrp_bruteforce(X, U, n):
    """Independent definition check: explicit sum over all entries (einsum)."""
    X = np.asarray(X, dtype=np.float64)
    N = X.ndim
    letters = 'abcdefgh'[:N]
    ops = []
    subs = [letters]
    ops.append(X)
    for m in range(N):
        if m != n:
            subs.append(letters[m] + 'r')
            ops.append(U[m])
    return np.einsum(','.join(subs) + '->' + letters[n] + 'r', *ops)


def full_ktensor(U, lam=None):
    """`full(ktensor(U))`."""
    R = U[0].shape[1]
    dims = [u.shape[0] for u in U]
    KR = khatrirao(list(U))           # (prod dims) x R, first index fastest
    if lam is None:
        v = KR.sum(axis=1)
    else:
        v = KR @ np.asarray(lam, dtype=np.float64)
    return np.reshape(v, dims, order='F')


def tensor_norm(X):
    """`norm(tensor)` -- Frobenius norm."""
    return float(np.sqrt(np.sum(np.asarray(X, dtype=np.float64) ** 2)))
