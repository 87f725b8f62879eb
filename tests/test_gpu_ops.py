"""GPU parity of the op-level C ABI entries against the CPU oracle (tolerances stated per test)."""
import zlib

import numpy as np
import pytest

from oracle import prox as OP
from oracle import aoadmm as OA
from oracle.tensor_ops import mttkrp as o_mttkrp
from helpers import rel_fro

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('dims,R', [((40, 50, 60), 3), ((20, 30, 40), 3), ((50, 30, 40), 4), ((33, 17, 29), 5),
                                    ((130, 7, 9), 10), ((7, 5, 300), 20), ((64, 64, 64), 20), ((3, 2, 2), 1),
                                    ((50, 70), 3), ((71, 33), 4), ((257, 129), 17), ((5, 6, 7, 8), 3) ])
@pytest.mark.parametrize('prec,tol', [('f64', 1e-12), ('f32', 2e-6)])
def test_mttkrp_matches_oracle(eng, dims, R, prec, tol):
    """mttkrp(X,U,n) (cmtf_fun_AOADMM.m:97).  fp64 path: 1e-12 relative Frobenius (summation order only);
    fp32 storage + f32 MFMA: 2e-6 (input rounding 6e-8 * sqrt(reduction length))."""
    rng = np.random.default_rng(sum(dims) + R)
    X = rng.standard_normal(dims)
    U = [rng.standard_normal((n, R)) for n in dims]
    for n in range(len(dims)):
        ref = o_mttkrp(X, U, n)
        got = eng.mttkrp(X, U, n, precision=prec)
        assert got.shape == ref.shape
        assert rel_fro(got, ref) < tol, (n, rel_fro(got, ref))


def test_mttkrp_asymmetric_layout(eng):
    """Exact-integer data with an asymmetric factor catches swapped MFMA fragment maps."""
    I, J, K, R = 131, 6, 5, 7
    X = np.arange(I * J * K, dtype=np.float64).reshape((I, J, K), order='F') % 17 - 8
    U = [np.arange(n * R, dtype=np.float64).reshape((n, R), order='F') % 5 - 2 for n in (I, J, K)]
    for prec in ('f64', 'f32'):
        for n in range(3):
            assert np.array_equal(eng.mttkrp(X, U, n, precision=prec), o_mttkrp(X, U, n))


def test_gram_and_chol(eng, pkg):
    rng = np.random.default_rng(3)
    F = rng.standard_normal((1000, 20))
    G = eng.gram(F)
    assert rel_fro(G, F.T @ F) < 1e-13
    B = F.T @ F + 3.0 * np.eye(20)
    L = eng.chol(B)
    assert rel_fro(L, np.linalg.cholesky(B)) < 1e-13
    with pytest.raises(pkg.NotPositiveDefinite):
        eng.chol(-np.eye(4))


CASES = [
    ('non-negativity',), ('box', -0.3, 0.4), ('simplex column-wise', 1.0), ('simplex row-wise', 2.0),
    ('non-decreasing',), ('non-increasing',), ('unimodality', True), ('unimodality', False), ('l1-ball', 3.0),
    ('l2-ball', 1.0), ('non-negative l2-ball', 1.0), ('non-negative l2-sphere', 1.0), ('orthonormal',),
    ('l1 regularization', 0.2), ('l0 regularization', 0.2), ('l2 regularization', 0.5), ('ridge', 0.3),
    ('GL smoothness', 0.7), ('TV regularization', 0.4),
]


@pytest.mark.parametrize('c', CASES, ids=[c[0] + str(c[1:]) for c in CASES])
@pytest.mark.parametrize('shape', [(60, 3), (257, 20), (1, 4), (2000, 20)])
def test_prox_matches_oracle(eng, c, shape):
    """Every device prox equals the oracle's operator to 1e-10 (they are exact algorithms;
    differences are summation order only)."""
    rng = np.random.default_rng(zlib.crc32(c[0].encode()) % 1000 + shape[0])   # not hash(): it changes per process
    X = rng.standard_normal(shape)
    if c[0] == 'orthonormal' and shape[0] < shape[1]:
        pytest.skip('needs rows >= cols')
    rho = 1.7
    ops, _ = OP.constraints_to_prox([1], [c], [shape[0]])
    ref = ops[0](X, rho)
    got = eng.prox(c, X, rho)
    assert rel_fro(got, ref) < 1e-10 or np.max(np.abs(got - ref)) < 1e-12, rel_fro(got, ref)


@pytest.mark.parametrize('kind', ['noise', 'steps', 'constant', 'ramp', 'strong'])
@pytest.mark.parametrize('rows', [4097, 6400, 6401, 9000, 20011])
def test_tv_prox_long_columns(eng, kind, rows):
    """TV on long columns: up to 6400 rows all working arrays in LDS (eight entries per thread), 6401-12000 rows the
    hybrid form (Pc / start / J in LDS, the rest in the prox workspace), beyond that everything in the workspace -- the
    same parallel split/merge active set instead of the one-thread scan; exact, like the short-column kernel."""
    rng = np.random.default_rng(rows + zlib.crc32(kind.encode()) % 1000)
    R = 3
    if kind == 'noise':
        X, eta = rng.standard_normal((rows, R)), 0.4
    elif kind == 'steps':
        X = np.repeat(rng.standard_normal((rows // 500 + 1, R)), 500, axis=0)[:rows] + 0.05 * rng.standard_normal((rows, R))
        eta = 0.3
    elif kind == 'constant':
        X, eta = np.full((rows, R), 1.25), 0.4
    elif kind == 'ramp':
        X, eta = np.linspace(-1, 1, rows)[:, None] * np.array([[1.0, -2.0, 0.5]]), 0.01
    else:                                               # one segment: the whole column merges to its mean
        X, eta = rng.standard_normal((rows, R)), 1e4
    c = ('TV regularization', eta)
    ops, _ = OP.constraints_to_prox([1], [c], [rows])
    ref = ops[0](X, 1.7)
    got = eng.prox(c, X, 1.7)
    assert rel_fro(got, ref) < 1e-10 or np.max(np.abs(got - ref)) < 1e-12, rel_fro(got, ref)


def _gl_banded(X, eta, rho):
    """(2*eta/rho*L + I) \\ X with the path-graph Laplacian L (constraints_to_prox.m:68-76) by a banded solve: the
    oracle builds the dense matrix (3 GB at 20 000 rows), so the long columns are checked against this and this
    against the oracle at 4097 rows."""
    from scipy.linalg import solve_banded
    n = X.shape[0]
    s2 = 2.0 * eta / rho
    ab = np.zeros((3, n))
    ab[0, 1:] = -s2
    ab[2, :-1] = -s2
    ab[1, :] = 2.0 * s2 + 1.0
    ab[1, 0] = ab[1, -1] = s2 + 1.0
    return solve_banded((1, 1), ab, X)


@pytest.mark.parametrize('rows', [4097, 9000, 20011])
@pytest.mark.parametrize('eta', [0.7, 40.0, 1e-3])
def test_gl_prox_long_columns(eng, rows, eta):
    """GL smoothness beyond the LDS-resident 4096 rows: the same cyclic reduction with its coefficient arrays in the
    prox workspace (prox_gl_pcr_long_k) instead of the one-thread Thomas solve."""
    rng = np.random.default_rng(rows)
    X = np.cumsum(rng.standard_normal((rows, 3)), axis=0) * 0.05 + rng.standard_normal((rows, 3))
    c = ('GL smoothness', eta)
    ref = _gl_banded(X, eta, 1.7)
    if rows == 4097:
        ops, _ = OP.constraints_to_prox([1], [c], [rows])
        assert rel_fro(ref, ops[0](X, 1.7)) < 1e-12
    got = eng.prox(c, X, 1.7)
    assert rel_fro(got, ref) < 1e-10, rel_fro(got, ref)


SHAPE_CASES = [('non-decreasing',), ('non-increasing',), ('unimodality', True), ('unimodality', False), ('GL smoothness', 0.7)]


@pytest.mark.parametrize('c', SHAPE_CASES, ids=[c[0] + str(c[1:]) for c in SHAPE_CASES])
@pytest.mark.parametrize('kind', ['long', 'verylong', 'staircase', 'ties', 'constant', 'negative', 'sorted', 'spike'])
def test_shape_constraints_hard_inputs(eng, c, kind):
    """The workgroup-parallel isotonic / unimodal projections and the cyclic-reduction GL solve (csrc/iso.hip) on the
    inputs that are hard for them: columns beyond the LDS-resident sizes (2500 rows: global scratch for the isotonic
    kernels; 5000 rows: sequential GL fallback), one outlier in front of sorted data (a pool that swallows the column
    one entry at a time in sequential PAVA), exact ties, constant and all-negative columns."""
    rng = np.random.default_rng(zlib.crc32(kind.encode()) % 1000)      # hash() of a str changes from process to process
    if kind == 'long':
        X = rng.standard_normal((5000 if c[0] == 'GL smoothness' else 2500, 2))
    elif kind == 'verylong':                            # beyond the LDS-resident prefix sums of iso_prev_k
        X = rng.standard_normal((6500, 1))
    elif kind == 'staircase':
        X = np.sort(rng.standard_normal((1500, 3)), axis=0)
        X[0, :] = 50.0
        X[-1, 1] = -50.0
    elif kind == 'ties':
        X = rng.integers(-2, 3, size=(700, 4)).astype(float)
    elif kind == 'constant':
        X = np.full((300, 3), -0.75)
        X[:, 1] = 2.0
    elif kind == 'negative':
        X = -np.abs(rng.standard_normal((400, 3))) - 0.1
    elif kind == 'sorted':
        X = np.sort(rng.standard_normal((1000, 2)), axis=0)
        X[:, 1] = X[::-1, 1]
    else:
        X = 0.01 * rng.standard_normal((1200, 3))
        X[600, :] = 5.0
        X[100, 1] = 4.0
    ops, _ = OP.constraints_to_prox([1], [c], [X.shape[0]])
    ref = ops[0](X, 1.3)
    got = eng.prox(c, X, 1.3)
    if kind == 'ties' and c[0] == 'unimodality' and not (rel_fro(got, ref) < 1e-10):
        # Integer data: many splits have exactly the same criterion value and the projection is not unique.  The
        # reference's pick among them is decided by the rounding of its error sums (project_unimodal_vector.m:67-72
        # accumulates them pool by pool); the device takes the first split within a few ulp of the minimum.  Every
        # column must then be feasible and exactly as close to the input as the oracle's.
        for r in range(X.shape[1]):
            g = got[:, r]
            pk = int(np.argmax(g))
            assert np.all(np.diff(g[:pk + 1]) >= 0) and np.all(np.diff(g[pk:]) <= 0)
            assert not c[1] or g.min() >= 0
            eg, er = np.sum((g - X[:, r]) ** 2), np.sum((ref[:, r] - X[:, r]) ** 2)
            assert abs(eg - er) <= 1e-12 * er, (r, eg, er)
        return
    assert rel_fro(got, ref) < 1e-10 or np.max(np.abs(got - ref)) < 1e-12, rel_fro(got, ref)


@pytest.mark.parametrize('seed', range(12))
def test_unimodal_integer_data(eng, seed):
    """Exactly tied data over many draws (the case above is one draw): where the pools, levels and thresholding flags are
    decided by exact comparisons the device must agree with the oracle; where only the split is tied it must return a
    projection that is feasible and exactly as close."""
    rng = np.random.default_rng(1000 + seed)
    X = rng.integers(-2, 3, size=(700, 4)).astype(float)
    for c in (('unimodality', True), ('unimodality', False)):
        ops, _ = OP.constraints_to_prox([1], [c], [X.shape[0]])
        ref = ops[0](X, 1.3)
        got = eng.prox(c, X, 1.3)
        for r in range(4):
            g = got[:, r]
            if np.max(np.abs(g - ref[:, r])) < 1e-12:
                continue
            pk = int(np.argmax(g))
            assert np.all(np.diff(g[:pk + 1]) >= 0) and np.all(np.diff(g[pk:]) <= 0)
            assert not c[1] or g.min() >= 0
            eg, er = np.sum((g - X[:, r]) ** 2), np.sum((ref[:, r] - X[:, r]) ** 2)
            assert abs(eg - er) <= 1e-12 * er, (c, r, eg, er)
    for c in (('non-decreasing',), ('non-increasing',)):
        ops, _ = OP.constraints_to_prox([1], [c], [X.shape[0]])
        assert np.max(np.abs(eng.prox(c, X, 1.3) - ops[0](X, 1.3))) < 1e-12


def test_sphere_zero_column(eng):
    X = -np.abs(np.random.default_rng(0).standard_normal((9, 3)))
    assert np.array_equal(eng.prox(('non-negative l2-sphere', 1), X, 1.0), OP.prox_normalized_nonneg(X))


@pytest.mark.parametrize('c', [('non-negativity',), ('TV regularization', 0.01), ('l2-ball', 1.0), ('simplex row-wise', 1.0),
                               ('unimodality', True)])
def test_admm_constrained_only(eng, c):
    """ADMM_constrained_only (cmtf_fun_AOADMM.m:591-623): 5 fixed inner iterations, 1e-11."""
    rng = np.random.default_rng(5)
    I, R = 300, 6
    F = rng.random((400, R))
    Bsys = F.T @ F
    rho = float(np.trace(Bsys) / R)
    A = rng.standard_normal((I, R))
    fac, Zc, mu = rng.random((I, R)), rng.random((I, R)), rng.random((I, R))
    ops, _ = OP.constraints_to_prox([1], [c], [I])
    L = np.linalg.cholesky(Bsys + rho / 2 * np.eye(R))
    f, z, m = fac.copy(), Zc.copy(), mu.copy()
    for _ in range(5):
        f = OA._solve_llt_right(A + rho / 2 * (z - m), L)
        z = ops[0](f + m, rho)
        m = m + f - z
    gf, gz, gm, its = eng.admm_constrained(A, Bsys, rho, c, fac, Zc, mu, 5, 0.0, 0.0)
    assert its == 5
    assert rel_fro(gf, f) < 1e-11 and rel_fro(gz, z) < 1e-11 and rel_fro(gm, m) < 1e-11


@pytest.mark.parametrize('dims', [(37, 22, 19), (64, 64, 5), (130, 9, 70), (41, 33)])
@pytest.mark.parametrize('prec,tol', [('f64', 1e-12), ('f32', 5e-6)])
def test_unfold_gram(pkg, eng, dims, prec, tol):
    """Y = X_(n) X_(n)' for every mode (cmtf_nvecs.m:40-56) against numpy."""
    rng = np.random.default_rng(sum(dims))
    X = rng.standard_normal(dims)
    for n in range(len(dims)):
        A = np.moveaxis(X, n, 0).reshape(dims[n], -1)
        Y = eng.unfold_gram(X, n, precision=prec)
        assert np.linalg.norm(Y - A @ A.T) / np.linalg.norm(A @ A.T) < tol


def test_nvecs_initialisation(pkg, eng):
    """init_options.nvecs = 1 (init_coupled_AOADMM_CMTF.m:50-73): the factors are the leading eigenvectors of the
    unfolding Gram matrices; eigenvector signs are arbitrary, so the spanned subspaces are compared."""
    from oracle import aoadmm as OA
    from helpers import script1_model
    rng = np.random.default_rng(71)
    Z, io = script1_model(rng, dims=(20, 30, 40), K=6, Jk=30, noise=0.05)
    io = dict(io, nvecs=1)
    Go = OA.init_coupled_AOADMM_CMTF({**Z, 'prox_operators': None}, io, rng=np.random.default_rng(3))
    Gg = pkg.init_coupled_AOADMM_CMTF(Z, io, rng=np.random.default_rng(3), engine=eng)

    def same_span(a, b):
        return np.linalg.norm(a @ (a.T @ b) - b) < 1e-8 * np.linalg.norm(b)
    for m, (fo, fg) in enumerate(zip(Go['fac'], Gg['fac'])):
        if isinstance(fo, list):
            assert all(same_span(x, y) for x, y in zip(fo, fg))
        elif np.allclose(fo, 1.0):
            assert np.allclose(fg, 1.0)                 # PARAFAC2 C mode: ones (:70-72)
        else:
            assert same_span(fo, fg), m


def test_nvecs_initialisation_from_resident_data(pkg):
    """The same initialisation when the model is built first: the Gram matrices come from the data already on the device
    (`aoadmm_resident_unfold_gram`; cmtf_nvecs.m:31-56 unfolds the array it already holds) and agree with the
    host-array form to 1e-12; the factors span the oracle's subspaces.  Also a synthetic (device-generated) block, which
    has no host array at all."""
    from oracle import aoadmm as OA
    from helpers import script1_model
    rng = np.random.default_rng(71)
    Z, io = script1_model(rng, dims=(20, 30, 40), K=6, Jk=30, noise=0.05)
    io = dict(io, nvecs=1)
    Go = OA.init_coupled_AOADMM_CMTF({**Z, 'prox_operators': None}, io, rng=np.random.default_rng(3))
    with pkg.Engine(0) as e:
        Z['_ranks'] = [3] * 6
        pkg.build_model(e, Z, 'f64')
        assert e._resident_model is Z
        # every Gram matrix both ways
        X1 = np.asarray(Z['object'][0])
        for n in range(3):
            a = e.resident_unfold_gram(0, n, X1.shape[n])
            b = e.unfold_gram(X1, n)
            assert np.linalg.norm(a - b) <= 1e-12 * np.linalg.norm(b), n
        M = np.hstack(Z['object'][1])
        a = e.resident_unfold_gram(1, 0, M.shape[0])
        assert np.linalg.norm(a - M @ M.T) <= 1e-12 * np.linalg.norm(M @ M.T)
        for k, Xk in enumerate(Z['object'][1]):
            a = e.resident_unfold_gram(1, 1, Xk.shape[1], k)
            assert np.linalg.norm(a - Xk.T @ Xk) <= 1e-12 * np.linalg.norm(Xk.T @ Xk), k
        calls = []
        orig = e.unfold_gram
        e.unfold_gram = lambda *a_, **k_: (calls.append(1), orig(*a_, **k_))[1]
        Gg = pkg.init_coupled_AOADMM_CMTF(Z, io, rng=np.random.default_rng(3), engine=e)
        assert not calls                                 # nothing went through the host-array form

    def same_span(a, b):
        return np.linalg.norm(a @ (a.T @ b) - b) < 1e-8 * np.linalg.norm(b)
    for m, (fo, fg) in enumerate(zip(Go['fac'], Gg['fac'])):
        if isinstance(fo, list):
            assert all(same_span(x, y) for x, y in zip(fo, fg))
        elif np.allclose(fo, 1.0):
            assert np.allclose(fg, 1.0)
        else:
            assert same_span(fo, fg), m
    # a block generated in HBM: nvecs = 1 works without any host copy of the tensor
    n, R = 48, 4
    Zs = dict(loss_function=['Frobenius'], model=['CP'], modes=[[1, 2, 3]], size=[n, n + 4, n + 8],
              coupling=dict(lin_coupled_modes=[0, 0, 0], coupling_type=[], coupl_trafo_matrices=[None] * 3),
              constrained_modes=[0, 0, 0], constraints=[None] * 3, weights=[1.0],
              object=[dict(synthetic=True, rank=R, seed=5, noise=0.0)], _ranks=[R] * 3)
    with pkg.Engine(0) as e:
        pkg.build_model(e, Zs, 'f64')
        io2 = dict(lambdas_init=[[1] * R], nvecs=1, distr=[lambda a, b: np.zeros((a, b))] * 3, normalize=0)
        Gs = pkg.init_coupled_AOADMM_CMTF(Zs, io2, rng=np.random.default_rng(1), engine=e)
        for m, f in enumerate(Gs['fac']):
            assert f.shape == (Zs['size'][m], R) and np.allclose(f.T @ f, np.eye(R), atol=1e-10)
        # noise-free rank-R data: the leading eigenvectors span the generating factor, so R eigenvalues carry everything
        Y = e.resident_unfold_gram(0, 0, n)
        w = np.sort(np.linalg.eigvalsh(Y))[::-1]
        assert w[R] < 1e-10 * w[0]


@pytest.mark.parametrize('dims', [(9, 7, 6, 5), (13, 4, 6, 3, 5), (34, 3, 2, 17)])
@pytest.mark.parametrize('prec,tol', [('f64', 1e-12), ('f32', 3e-6)])
def test_mttkrp_nway(pkg, eng, dims, prec, tol):
    """Tensors of order > 3 (the toolbox mttkrp is N-way): one matrix-core contraction + successive folds over T."""
    rng = np.random.default_rng(len(dims) + sum(dims))
    X = rng.standard_normal(dims)
    R = 4
    U = [rng.standard_normal((n, R)) for n in dims]
    for n in range(len(dims)):
        ref = o_mttkrp(X, U, n)
        got = eng.mttkrp(X, U, n, precision=prec)
        assert rel_fro(got, ref) < tol, (n, rel_fro(got, ref))


def test_cell_state_one_slab_at_a_time_equals_all_slabs(pkg, eng):
    """Cell-valued fields of G (PARAFAC2 B_k): per-slab transfers and the AOADMM_ALL_SLABS form address the same memory."""
    import ctypes as C
    from helpers import script4_model
    capi = __import__('importlib').import_module('matlab-code_amd._capi')
    rng = np.random.default_rng(3)
    Z, io = script4_model(rng, K=5)
    Z['_ranks'] = [3, 3, 3]
    pkg.build_model(eng, Z, 'f64')
    R = 3
    cells = [np.asfortranarray(rng.standard_normal((j, R))) for j in Z['size'][1]]
    for k, c in enumerate(cells):                                        # slab by slab in ...
        capi.check(eng.lib.aoadmm_state_set(eng.h, capi.F_FAC, 1, k, capi.dptr(c), c.shape[0], R))
    rows = sum(c.shape[0] for c in cells)
    packed = np.zeros(rows * R)                                          # ... all at once out
    capi.check(eng.lib.aoadmm_state_get(eng.h, capi.F_FAC, 1, capi.ALL_SLABS, capi.dptr(packed), rows, R))
    assert np.array_equal(packed, np.concatenate([c.ravel(order='F') for c in cells]))
    packed2 = rng.standard_normal(rows * R)                              # all at once in, slab by slab out
    capi.check(eng.lib.aoadmm_state_set(eng.h, capi.F_FAC, 1, capi.ALL_SLABS, capi.dptr(packed2), rows, R))
    o = 0
    for k, c in enumerate(cells):
        got = np.zeros(c.shape, order='F')
        capi.check(eng.lib.aoadmm_state_get(eng.h, capi.F_FAC, 1, k, capi.dptr(got), c.shape[0], R))
        assert np.array_equal(got.ravel(order='F'), packed2[o:o + c.size])
        o += c.size
    with pytest.raises(Exception):                                        # wrong total row count is rejected
        capi.check(eng.lib.aoadmm_state_set(eng.h, capi.F_FAC, 1, capi.ALL_SLABS, capi.dptr(packed2), rows - 1, R))


def test_plain_c_caller_runs(tmp_path):
    """examples/solve_cp.c built with gcc and run as its own process: the C ABI needs nothing from Python or C++."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / 'solve_cp'
    subprocess.run(['gcc', '-std=c99', '-O2', '-I', os.path.join(root, 'include'), os.path.join(root, 'examples', 'solve_cp.c'),
                    '-L', os.path.join(root, 'matlab-code_amd'), '-laoadmm_hip', '-lm', '-o', str(exe)], check=True)
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(root, 'matlab-code_amd') + ':' + os.environ.get('LD_LIBRARY_PATH', ''))
    out = subprocess.run([str(exe)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert 'RESULT ok' in out.stdout, out.stdout
