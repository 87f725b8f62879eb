"""CPU: pin the oracle's proximal operators by their defining properties (the reference has no tests;
third-party operators are 'parity unpinned', see oracle/__init__.py): feasibility, first-order
optimality / KKT conditions, idempotence of projections, and a second independent solver for TV."""
import numpy as np
import pytest
from scipy.optimize import lsq_linear

from oracle import prox as P

RNG = np.random.default_rng(0)


def test_box_and_nonneg():
    x = RNG.standard_normal((7, 3))
    assert np.array_equal(P.project_box(x, 0, np.inf), np.maximum(x, 0))
    y = P.project_box(x, -0.2, 0.3)
    assert y.min() >= -0.2 and y.max() <= 0.3 and np.array_equal(P.project_box(y, -0.2, 0.3), y)


@pytest.mark.parametrize('direction', [1, 2])
def test_simplex_projection_optimality(direction):
    x = RNG.standard_normal((40, 6)) * 2
    eta = 1.5
    y = P.project_simplex(x, eta, direction)
    s = y.sum(axis=0 if direction == 1 else 1)
    assert np.allclose(s, eta) and y.min() >= 0
    # KKT: x - y = tau on the support, <= tau... i.e. (x - y) is constant where y > 0 and x <= tau elsewhere
    V = (x - y) if direction == 1 else (x - y).T
    Y = y if direction == 1 else y.T
    X = x if direction == 1 else x.T
    for c in range(V.shape[1]):
        supp = Y[:, c] > 0
        tau = V[supp, c]
        assert np.allclose(tau, tau[0])
        assert np.all(X[~supp, c] <= tau[0] + 1e-12)


def test_monotone_is_isotonic_regression():
    x = RNG.standard_normal((50, 4))
    y = P.project_monotone(x)
    assert np.all(np.diff(y, axis=0) >= -1e-14)
    # optimality: block means equal data means on each constant block; projection is idempotent
    assert np.allclose(P.project_monotone(y), y)
    for c in range(4):
        # compare with a brute-force QP via cumulative parametrisation on a short prefix
        n = 8
        A = np.tril(np.ones((n, n)))
        lb = np.r_[-np.inf, np.zeros(n - 1)]
        res = lsq_linear(A, x[:n, c], bounds=(lb, np.inf), tol=1e-14)
        assert np.allclose(A @ res.x, P._pava_nondecreasing(x[:n, c]), atol=1e-7)


def test_l1_l2_balls():
    x = RNG.standard_normal((30, 5)) * 3
    y = P.project_L1(x, 2.0)
    assert np.all(np.abs(y).sum(axis=0) <= 2.0 + 1e-12)
    inside = RNG.standard_normal((30, 2)) * 1e-3
    assert np.array_equal(P.project_L1(inside, 2.0), inside)
    z = P.project_L2(x, 1.0)
    assert np.allclose(np.linalg.norm(z, axis=0), 1.0)
    assert np.allclose(z / np.linalg.norm(z, axis=0), x / np.linalg.norm(x, axis=0))


def test_soft_hard_block_thresholds():
    x = np.array([[-2.0, 0.1], [0.5, -0.05], [3.0, 0.0]])
    assert np.allclose(P.prox_abs(x, 1.0), [[-1, 0], [0, 0], [2, 0]])
    # prox of gamma*|x|_0: keep x iff x^2/2 > gamma
    assert np.array_equal(P.prox_zero(x, 0.125), np.where(np.abs(x) > 0.5, x, 0))
    y = P.prox_L2(x, 1.0)
    nrm = np.linalg.norm(x, axis=0)
    assert np.allclose(y[:, 0], x[:, 0] * (1 - 1 / nrm[0])) and np.all(y[:, 1] == 0)


def _tv_reference(y, lam):
    """Independent exact solver: dual box-constrained least squares  min ||D'u - y||^2, |u| <= lam."""
    n = y.size
    D = np.diff(np.eye(n), axis=0)          # (n-1) x n
    res = lsq_linear(D.T, y, bounds=(-lam, lam), tol=1e-15, max_iter=2000)
    return y - D.T @ res.x


@pytest.mark.parametrize('n,lam', [(2, 0.3), (17, 0.05), (40, 0.5), (40, 5.0), (25, 1e-9)])
def test_tv_condat_against_dual_qp(n, lam):
    y = np.cumsum(RNG.standard_normal(n)) * 0.3
    x = P.tv1d_condat(y, lam)
    assert np.allclose(x, _tv_reference(y, lam), atol=2e-6)
    # KKT: u = cumsum(y - x) satisfies |u| <= lam, u_end = 0, u_i = -lam*sign(x_{i+1}-x_i) on jumps
    u = np.cumsum(y - x)
    assert abs(u[-1]) < 1e-10 and np.all(np.abs(u[:-1]) <= lam * (1 + 1e-9) + 1e-12)
    d = np.diff(x)
    jump = np.abs(d) > 1e-10
    assert np.allclose(u[:-1][jump], -lam * np.sign(d[jump]), atol=1e-9)


def test_tv_edge_cases():
    assert np.array_equal(P.tv1d_condat(np.array([3.0]), 1.0), [3.0])
    y = RNG.standard_normal(12)
    assert np.array_equal(P.tv1d_condat(y, 0.0), y)
    assert np.allclose(P.tv1d_condat(y, 1e6), y.mean())
    assert P.tv1d_condat(np.zeros(0), 1.0).size == 0


@pytest.mark.parametrize('nonneg', [False, True])
def test_unimodal_is_best_unimodal_fit(nonneg):
    """project_unimodal_vector.m: result is unimodal (non-decreasing then non-increasing) and no worse
    than the best split found by brute force over isotonic fits."""
    for trial in range(5):
        y = RNG.standard_normal(14) + (0.8 if nonneg else 0.0)
        x = P.project_unimodal_vector(y, nonneg)
        k = int(np.argmax(x))
        assert np.all(np.diff(x[:k + 1]) >= -1e-12) and np.all(np.diff(x[k:]) <= 1e-12)
        if nonneg:
            assert x.min() >= 0
        best = np.inf
        for m in range(1, y.size + 1):
            left = P._pava_nondecreasing(y[:m])
            right = -P._pava_nondecreasing(-y[m:]) if m < y.size else np.zeros(0)
            cand = np.r_[left, right]
            if nonneg:
                cand = np.maximum(cand, 0)
            best = min(best, np.sum((cand - y) ** 2))
        assert np.sum((x - y) ** 2) <= best + 1e-9


def test_ortho_and_sphere():
    X = RNG.standard_normal((20, 4))
    Q = P.project_ortho(X)
    assert np.allclose(Q.T @ Q, np.eye(4))
    # polar factor: Q'X symmetric positive semi-definite
    S = Q.T @ X
    assert np.allclose(S, S.T) and np.all(np.linalg.eigvalsh(S) > -1e-12)
    Y = P.prox_normalized_nonneg(np.c_[X[:, :2], -np.abs(X[:, 2:3])])
    assert np.allclose(np.linalg.norm(Y, axis=0), 1.0) and Y.min() >= 0
    assert Y[:, 2].sum() == 1.0 and Y[np.argmax(-np.abs(X[:, 2])), 2] == 1.0   # prox_normalized_nonneg.m:5-7


def test_quadratic_and_tparafac2_solves():
    n = 9
    L = P.gl_laplacian(n)
    assert L[0, 0] == 1 and L[-1, -1] == 1 and L[3, 3] == 2 and L[3, 4] == -1      # constraints_to_prox.m:71-73
    x = RNG.standard_normal((n, 3))
    y = P.prox_quadratic(x, 0.35, L)
    assert np.allclose((2 * 0.35 * L + np.eye(n)) @ y, x)
    K = 5
    F = [RNG.standard_normal((6, 2)) for _ in range(K)]
    rho = RNG.random(K) + 0.5
    eta = 0.4
    out = P.t_smoothness_prox(F, rho, eta)
    # optimality of sum_k rho_k/2 ||Z_k - F_k||^2 + eta*sum ||Z_k - Z_{k-1}||^2  (t_smoothness_prox.m)
    for k in range(K):
        g = rho[k] * (out[k] - F[k])
        if k > 0:
            g = g + 2 * eta * (out[k] - out[k - 1])
        if k < K - 1:
            g = g + 2 * eta * (out[k] - out[k + 1])
        assert np.allclose(g, 0, atol=1e-10)


def test_constraints_to_prox_wiring():
    names = [('non-negativity',), ('box', 0, 1), ('simplex column-wise', 1), ('simplex row-wise', 1), ('non-decreasing',),
             ('non-increasing',), ('unimodality', True), ('l1-ball', 1), ('l2-ball', 1), ('non-negative l2-ball', 1),
             ('non-negative l2-sphere', 1), ('orthonormal',), ('l1 regularization', .1), ('l0 regularization', .1),
             ('l2 regularization', .1), ('ridge', .1), ('GL smoothness', .1), ('TV regularization', .1)]
    ops, reg = P.constraints_to_prox([1] * len(names), names, [12] * len(names))
    x = RNG.standard_normal((12, 3))
    for op in ops:
        assert op(x, 2.0).shape == x.shape
    # reg_func exists exactly for the regularisations (constraints_to_prox.m:49,53,57,61,77,81)
    assert [r is not None for r in reg] == [False] * 12 + [True] * 6
    # the TV value has no abs() in the reference (:81): it telescopes
    assert np.isclose(reg[-1](x), 0.1 * np.sum(x[-1] - x[0]))
    with pytest.raises(ValueError):
        P.constraints_to_prox([1], [None], [3])
