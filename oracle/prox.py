"""Oracle (test infrastructure): proximal operators of the AO-ADMM framework.

numpy/fp64 restatement of what `functions/constraints_to_prox.m:13-91` wires
up.  In-repo operators follow the reference file line by line; third-party
ones (Proximity Operator Repository, TV_Condat_v2 -- absent from
/root/reference, see oracle/__init__.py) are restated from their mathematical
definition: each is the unique minimiser of a strictly convex problem, so any
exact algorithm gives the same answer up to rounding ("parity unpinned").

Every function takes and returns a 2-D float64 array (I_n x R); "column-wise"
means along axis 0, exactly like MATLAB's `dir = 1`.
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------
# third-party: Proximity Operator Repository (definitions)
# --------------------------------------------------------------------------


def project_box(x, lo, hi):
    """`project_box(x,l,u)` -- constraints_to_prox.m:14,18. Clamp."""
    return np.minimum(np.maximum(x, lo), hi)


def _simplex_vec(v, eta):
    """Euclidean projection of a vector onto {x>=0, sum x = eta} (sort based)."""
    n = v.size
    u = np.sort(v)[::-1]
    css = np.cumsum(u) - eta
    k = np.arange(1, n + 1)
    cond = u - css / k > 0
    rho = np.nonzero(cond)[0][-1]
    tau = css[rho] / (rho + 1.0)
    return np.maximum(v - tau, 0.0)


def project_simplex(x, eta, direction):
    """`project_simplex(x,eta,dir)` -- constraints_to_prox.m:21,24.

    dir=1: every column sums to eta; dir=2: every row sums to eta.
    """
    x = np.asarray(x, dtype=np.float64)
    out = np.empty_like(x)
    if direction == 1:
        for r in range(x.shape[1]):
            out[:, r] = _simplex_vec(x[:, r], eta)
    else:
        for i in range(x.shape[0]):
            out[i, :] = _simplex_vec(x[i, :], eta)
    return out


def _pava_nondecreasing(y):
    """Isotonic (non-decreasing) L2 regression, unit weights, pool-adjacent-violators."""
    n = y.size
    val = np.empty(n)
    wt = np.empty(n)
    length = np.empty(n, dtype=np.int64)
    nb = 0
    for i in range(n):
        val[nb] = y[i]
        wt[nb] = 1.0
        length[nb] = 1
        nb += 1
        while nb > 1 and val[nb - 2] > val[nb - 1]:
            w = wt[nb - 2] + wt[nb - 1]
            val[nb - 2] = (val[nb - 2] * wt[nb - 2] + val[nb - 1] * wt[nb - 1]) / w
            wt[nb - 2] = w
            length[nb - 2] += length[nb - 1]
            nb -= 1
    return np.repeat(val[:nb], length[:nb])


def project_monotone(x, direction=1):
    """`project_monotone(x,1)` -- constraints_to_prox.m:26,28. Column-wise non-decreasing."""
    x = np.asarray(x, dtype=np.float64)
    out = np.empty_like(x)
    for r in range(x.shape[1]):
        out[:, r] = _pava_nondecreasing(x[:, r])
    return out


def project_L1(x, eta, direction=1):
    """`project_L1(x,eta,1)` -- constraints_to_prox.m:34. Column-wise l1-ball ||x||_1<=eta."""
    x = np.asarray(x, dtype=np.float64)
    out = x.copy()
    for r in range(x.shape[1]):
        v = x[:, r]
        if np.sum(np.abs(v)) > eta:
            w = _simplex_vec(np.abs(v), eta)
            out[:, r] = np.sign(v) * w
    return out


def project_L2(x, eta, direction=1):
    """`project_L2(x,eta,1)` -- constraints_to_prox.m:37,40. Column-wise l2-ball."""
    x = np.asarray(x, dtype=np.float64)
    nrm = np.sqrt(np.sum(x * x, axis=0))
    scale = np.ones_like(nrm)
    big = nrm > eta
    scale[big] = eta / nrm[big]
    return x * scale[None, :]


def prox_abs(x, gamma):
    """`prox_abs(x,gamma)` -- constraints_to_prox.m:48. Soft threshold."""
    return np.sign(x) * np.maximum(np.abs(x) - gamma, 0.0)


def prox_zero(x, gamma):
    """`prox_zero(x,gamma)` -- constraints_to_prox.m:52. prox of gamma*||x||_0 (hard threshold)."""
    return np.where(x * x > 2.0 * gamma, x, 0.0)


def prox_L2(x, gamma, direction=1):
    """`prox_L2(x,gamma,1)` -- constraints_to_prox.m:56. Column-wise block soft threshold."""
    x = np.asarray(x, dtype=np.float64)
    nrm = np.sqrt(np.sum(x * x, axis=0))
    scale = np.zeros_like(nrm)
    big = nrm > gamma
    scale[big] = 1.0 - gamma / nrm[big]
    return x * scale[None, :]


# --------------------------------------------------------------------------
# third-party: TV_Condat_v2 (definition: exact 1-D TV prox)
# --------------------------------------------------------------------------


def tv1d_condat(y, lam):
    """Exact minimiser of 0.5*||x-y||^2 + lam*sum_i |x[i+1]-x[i]|.

    Direct (taut-string like) algorithm of L. Condat, "A direct algorithm for
    1-D total variation denoising", IEEE SPL 20(11), 2013 -- restated from the
    paper; `TV_Condat_v2` called at functions/prox_TV.m:7 returns the same
    unique minimiser.
    """
    y = np.asarray(y, dtype=np.float64)
    n = y.size
    x = np.empty(n)
    if n == 0:
        return x
    if lam <= 0:
        return y.copy()
    k = k0 = km = kp = 0
    vmin = y[0] - lam
    vmax = y[0] + lam
    umin = lam
    umax = -lam
    while True:
        if k == n - 1:
            if umin < 0.0:
                while k0 <= km:
                    x[k0] = vmin
                    k0 += 1
                k = km = kp = k0
                vmin = y[k]
                umin = lam
                umax = vmin + umin - vmax
            elif umax > 0.0:
                while k0 <= kp:
                    x[k0] = vmax
                    k0 += 1
                k = km = kp = k0
                vmax = y[k]
                umax = -lam
                umin = vmax + umax - vmin
            else:
                vmin += umin / (k - k0 + 1)
                while k0 <= k:
                    x[k0] = vmin
                    k0 += 1
                return x
        else:
            umin += y[k + 1] - vmin
            if umin < -lam:
                while k0 <= km:
                    x[k0] = vmin
                    k0 += 1
                k = km = kp = k0
                vmin = y[k]
                vmax = vmin + 2.0 * lam
                umin = lam
                umax = -lam
            else:
                umax += y[k + 1] - vmax
                if umax > lam:
                    while k0 <= kp:
                        x[k0] = vmax
                        k0 += 1
                    k = km = kp = k0
                    vmax = y[k]
                    vmin = vmax - 2.0 * lam
                    umin = lam
                    umax = -lam
                else:
                    k += 1
                    if umin >= lam:
                        km = k
                        vmin += (umin - lam) / (km - k0 + 1)
                        umin = lam
                    if umax <= -lam:
                        kp = k
                        vmax += (umax + lam) / (kp - k0 + 1)
                        umax = -lam


def prox_TV(X, lam):
    """functions/prox_TV.m:1-9 -- column-wise TV prox."""
    X = np.asarray(X, dtype=np.float64)
    out = np.zeros_like(X)
    for r in range(X.shape[1]):
        out[:, r] = tv1d_condat(X[:, r], lam)
    return out


# --------------------------------------------------------------------------
# in-repo operators (restated line by line)
# --------------------------------------------------------------------------


def _prefix_isotonic_regression(y, non_negativity):
    """functions/project_unimodal_vector.m:43-88 (Stout 2008 prefix isotonic regression).

    Returns (level_set[1:], index_range[1:]-1, error[1:]) with MATLAB's 1-based
    `index_range` kept 1-based relative to the *returned* arrays (i.e. the value
    `index_range(i)-1` of the reference, `:79`).
    """
    n = y.shape[0]
    # 1-based arrays of length n+1, slot 0 unused except sentinel at slot 1 -> we
    # keep MATLAB indices by allocating n+2 and ignoring slot 0.
    sumwy = np.zeros(n + 2)
    sumwy2 = np.zeros(n + 2)
    sumw = np.zeros(n + 2)
    sumwy[2:] = y                      # :45  sumwy = [0;y]
    sumwy2[2:] = y ** 2                # :46
    sumw[2:] = 1.0                     # :47
    level_set = np.zeros(n + 2)        # :49
    index_range = np.zeros(n + 2, dtype=np.int64)  # :50
    error = np.zeros(n + 2)            # :51
    level_set[1] = -np.inf             # :53
    if non_negativity:
        cumsumwy2 = np.zeros(n + 2)
        cumsumwy2[1:] = np.cumsum(sumwy2[1:])   # :56
        threshold = np.zeros(n + 2, dtype=bool)  # :57
    for i in range(2, n + 2):          # :60
        level_set[i] = y[i - 2]        # :61
        index_range[i] = i             # :62
        while level_set[i] <= level_set[index_range[i] - 1]:   # :63
            merger = index_range[i] - 1
            sumwy[i] += sumwy[merger]          # :83
            sumwy2[i] += sumwy2[merger]        # :84
            sumw[i] += sumw[merger]            # :85
            level_set[i] = sumwy[i] / sumw[i]  # :86
            index_range[i] = index_range[index_range[i] - 1]   # :65
        levelerror = sumwy2[i] - (sumwy[i] ** 2 / sumw[i])     # :67
        if non_negativity and level_set[i] < 0:                # :68
            threshold[i] = True
            error[i] = cumsumwy2[i - 1]                        # :70
        else:
            error[i] = levelerror + error[index_range[i] - 1]  # :72
    if non_negativity:
        level_set[threshold] = 0.0                             # :76
    return level_set[2:], index_range[2:] - 1, error[2:]       # :79-80


def _compute_isotonic_from_index(mode_idx, level_set, index_range):
    """functions/project_unimodal_vector.m:34-41 (1-based mode_idx, 1-based index_range)."""
    y_iso = np.full(mode_idx, np.nan)
    idx = mode_idx
    while idx >= 1:
        lo = index_range[idx - 1]
        y_iso[lo - 1:idx] = level_set[idx - 1]
        idx = lo - 1
    return y_iso


def project_unimodal_vector(x, non_negativity):
    """functions/project_unimodal_vector.m:1-19."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    lvl_l, rng_l, err_l = _prefix_isotonic_regression(x, non_negativity)       # :11
    lvl_r, rng_r, err_r = _prefix_isotonic_regression(x[::-1], non_negativity)  # :12
    # get_best_unimodality_index, :21-32 (1-based i)
    best_error = err_r[n - 1]
    best_idx = 1
    for i in range(2, n + 1):
        e = err_l[i - 1] + err_r[n - (i - 1) - 1]
        if e < best_error:
            best_error = e
            best_idx = i
    left = _compute_isotonic_from_index(best_idx, lvl_l, rng_l)       # :15
    right = _compute_isotonic_from_index(n - best_idx, lvl_r, rng_r)  # :16
    return np.concatenate([left, right[::-1]])                        # :18


def project_unimodal(X, non_negativity):
    """functions/project_unimodal.m:10-14."""
    X = np.asarray(X, dtype=np.float64)
    out = np.zeros_like(X)
    for r in range(X.shape[1]):
        out[:, r] = project_unimodal_vector(X[:, r], non_negativity)
    return out


def project_ortho(X):
    """functions/project_ortho.m:3-4 -- U*V' of the economy SVD."""
    U, _, Vt = np.linalg.svd(np.asarray(X, dtype=np.float64), full_matrices=False)
    return U @ Vt


def prox_normalized_nonneg(X):
    """functions/prox_normalized_nonneg.m:3-11."""
    X = np.asarray(X, dtype=np.float64)
    Y = project_box(X, 0.0, np.inf)
    for r in range(Y.shape[1]):
        nr = np.linalg.norm(Y[:, r])
        if nr == 0:
            Y[np.argmax(X[:, r]), r] = 1.0     # :6-7 (first maximum, like MATLAB max)
        else:
            Y[:, r] = Y[:, r] / nr
    return Y


def gl_laplacian(n):
    """constraints_to_prox.m:71-73 -- path-graph Laplacian."""
    L = 2.0 * np.eye(n) - np.eye(n, k=1) - np.eye(n, k=-1)
    L[0, 0] = 1.0
    L[-1, -1] = 1.0
    return L


def prox_quadratic(x, scale, L):
    """constraints_to_prox.m:66,76 -- (2*eta/rho*L + I) \\ x, `scale` = eta/rho."""
    n = L.shape[0]
    return np.linalg.solve(2.0 * scale * L + np.eye(n), x)


def t_smoothness_prox(factor_matrices, rho, smoothness_l):
    """functions/t_smoothness_prox.m:1-58 -- Thomas solve across the K slabs."""
    K = len(factor_matrices)
    rho = np.asarray(rho, dtype=np.float64).reshape(-1)
    rhs = [rho[i] * np.asarray(factor_matrices[i], dtype=np.float64) for i in range(K)]  # :8-10
    A = np.zeros((K, K))
    for i in range(K):
        for j in range(K):
            if i == j:
                A[i, j] = 4 * smoothness_l + rho[i]
            elif i == j - 1 or i == j + 1:
                A[i, j] = -2 * smoothness_l
    A[0, 0] -= 2 * smoothness_l       # :37
    A[-1, -1] -= 2 * smoothness_l     # :38
    for i in range(1, K):             # :42-46
        m = A[i, i - 1] / A[i - 1, i - 1]
        A[i, i] = A[i, i] - m * A[i - 1, i]
        rhs[i] = rhs[i] - m * rhs[i - 1]
    out = [None] * K
    out[-1] = rhs[-1] / A[-1, -1]     # :50
    q = out[-1]
    for k in range(K - 2, -1, -1):    # :53-56
        q = (rhs[k] - A[k, k + 1] * q) / A[k, k]
        out[k] = q
    return out


def t_smoothness_penalty(x, smoothness_l):
    """functions/t_smoothness_penalty.m:1-10."""
    loss = 0.0
    for i in range(1, len(x)):
        loss += np.linalg.norm(x[i] - x[i - 1], 'fro') ** 2
    return smoothness_l * loss


# --------------------------------------------------------------------------
# constraints_to_prox
# --------------------------------------------------------------------------


def constraints_to_prox(constrained_modes, constraints, sz):
    """functions/constraints_to_prox.m:1-94.

    `constraints[m]` is a tuple/list like ('non-negativity',) or
    ('TV regularization', 0.001); returns (prox_operators, reg_func) lists of
    callables `prox(x, rho)` / `reg(x)` (None where the reference leaves the
    cell empty).
    """
    n = len(constrained_modes)
    prox_operators = [None] * n
    reg_func = [None] * n
    for m in range(n):
        if not constrained_modes[m]:
            continue
        c = constraints[m]
        if c is None or len(c) == 0:
            raise ValueError('No constraint provided for mode %d.' % (m + 1))   # :11
        name = c[0]
        if name == 'non-negativity':                                           # :13
            prox_operators[m] = lambda x, rho: project_box(x, 0.0, np.inf)
        elif name == 'box':                                                    # :15
            lo, hi = c[1], c[2]
            prox_operators[m] = lambda x, rho, lo=lo, hi=hi: project_box(x, lo, hi)
        elif name == 'simplex column-wise':                                    # :19
            eta = c[1]
            prox_operators[m] = lambda x, rho, eta=eta: project_simplex(x, eta, 1)
        elif name == 'simplex row-wise':                                       # :22
            eta = c[1]
            prox_operators[m] = lambda x, rho, eta=eta: project_simplex(x, eta, 2)
        elif name == 'non-decreasing':                                         # :25
            prox_operators[m] = lambda x, rho: project_monotone(x, 1)
        elif name == 'non-increasing':                                         # :27
            prox_operators[m] = lambda x, rho: -project_monotone(-x, 1)
        elif name == 'unimodality':                                            # :29
            nn = bool(c[1])
            prox_operators[m] = lambda x, rho, nn=nn: project_unimodal(x, nn)
        elif name == 'l1-ball':                                                # :32
            eta = c[1]
            prox_operators[m] = lambda x, rho, eta=eta: project_L1(x, eta, 1)
        elif name == 'l2-ball':                                                # :35
            eta = c[1]
            prox_operators[m] = lambda x, rho, eta=eta: project_L2(x, eta, 1)
        elif name == 'non-negative l2-ball':                                   # :38
            eta = c[1]
            prox_operators[m] = lambda x, rho, eta=eta: project_L2(project_box(x, 0.0, np.inf), eta, 1)
        elif name == 'non-negative l2-sphere':                                 # :41 (eta ignored)
            prox_operators[m] = lambda x, rho: prox_normalized_nonneg(x)
        elif name == 'orthonormal':                                            # :44
            prox_operators[m] = lambda x, rho: project_ortho(x)
        elif name == 'l1 regularization':                                      # :46
            eta = c[1]
            prox_operators[m] = lambda x, rho, eta=eta: prox_abs(x, eta / rho)
            reg_func[m] = lambda x, eta=eta: eta * np.sum(np.sum(np.abs(x), axis=0))
        elif name == 'l0 regularization':                                      # :50
            eta = c[1]
            prox_operators[m] = lambda x, rho, eta=eta: prox_zero(x, eta / rho)
            reg_func[m] = lambda x, eta=eta: eta * float(np.count_nonzero(x))
        elif name == 'l2 regularization':                                      # :54
            eta = c[1]
            prox_operators[m] = lambda x, rho, eta=eta: prox_L2(x, eta / rho, 1)
            reg_func[m] = lambda x, eta=eta: eta * np.sum(np.sqrt(np.sum(x * x, axis=0)))
        elif name == 'ridge':                                                  # :58
            eta = c[1]
            prox_operators[m] = lambda x, rho, eta=eta: 1.0 / (2.0 * (eta / rho) + 1.0) * x
            reg_func[m] = lambda x, eta=eta: eta * np.linalg.norm(x, 'fro') ** 2
        elif name == 'quadratic regularization':                               # :62
            eta = c[1]
            L = np.asarray(c[2], dtype=np.float64)
            prox_operators[m] = lambda x, rho, eta=eta, L=L: prox_quadratic(x, eta / rho, L)
            reg_func[m] = lambda x, eta=eta, L=L: eta * np.trace(x.T @ L @ x)
        elif name == 'GL smoothness':                                          # :68
            eta = c[1]
            szm = sz[m][0] if isinstance(sz[m], (list, tuple, np.ndarray)) else sz[m]   # :70
            L = gl_laplacian(int(szm))
            prox_operators[m] = lambda x, rho, eta=eta, L=L: prox_quadratic(x, eta / rho, L)
            reg_func[m] = lambda x, eta=eta, L=L: eta * np.trace(x.T @ L @ x)
        elif name == 'TV regularization':                                      # :78
            eta = c[1]
            prox_operators[m] = lambda x, rho, eta=eta: prox_TV(x, eta / rho)
            # QUIRK kept on purpose (:81): no absolute value -> telescoping sum.
            reg_func[m] = lambda x, eta=eta: eta * np.sum(x[1:, :] - x[:-1, :])
        elif name == 'tPARAFAC2':                                              # :82
            eta = c[1]
            prox_operators[m] = lambda x, rho, eta=eta: t_smoothness_prox(x, rho, eta)
            reg_func[m] = lambda x, eta=eta: t_smoothness_penalty(x, eta)
        elif name == 'custom':                                                 # :86
            prox_operators[m] = c[1]
            if len(c) > 2:
                reg_func[m] = c[2]
        # unknown names fall through silently, exactly like the if/elseif chain
    return prox_operators, reg_func
