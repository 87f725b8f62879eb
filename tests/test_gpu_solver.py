"""GPU parity of the solver-level entry (replaces cmtf_fun_AOADMM, cmtf_AOADMM.m:193) against the oracle:
identical init struct, tolerances 0 => fixed iteration counts; bar = 1e-8 relative Frobenius
on every factor matrix (BASELINE.json north_star)."""
import copy
import os

import numpy as np
import pytest

from oracle import aoadmm as OA
from helpers import cp_cp_exact_model, cp_model, options, rel_fro, script3_model

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=['one-launch', 'tensor-pass'])
def mttkrp_path(request):
    """Every solver test runs twice: tiny blocks take the one-launch MTTKRP kernel by default, which would leave the
    tensor-pass kernels, the partial-contraction cache and the fused EM contraction untested at oracle-sized shapes."""
    if request.param == 'tensor-pass':
        os.environ['AOADMM_NO_SMALL_MTTKRP'] = '1'
    yield
    os.environ.pop('AOADMM_NO_SMALL_MTTKRP', None)
TOL = 1e-8


def run_both(pkg, eng, Z, io, opt, seed=7, precision='f64', Delta=None):
    rng = np.random.default_rng(seed)
    G = OA.init_coupled_AOADMM_CMTF({**Z, 'prox_operators': None}, io, Delta=Delta, rng=rng)
    _, Fo, _, oo = OA.cmtf_AOADMM(Z, alg_options=opt, init=copy.deepcopy(G))
    _, Fg, _, og = pkg.cmtf_AOADMM(Z, alg_options=opt, init=copy.deepcopy(G), engine=eng, precision=precision)
    return Fo, oo, Fg, og


def compare(Fo, oo, Fg, og, tol=TOL):
    for key in ('fac', 'constraint_fac', 'constraint_dual_fac', 'coupling_fac', 'coupling_dual_fac'):
        for a, b in zip(Fo[key], Fg[key]):
            if a is None:
                continue
            assert rel_fro(b, a) < tol, (key, rel_fro(b, a))
    assert og['OuterIterations'] == oo['OuterIterations']
    assert np.array_equal(og['innerIters'], oo['innerIters'])
    for k in ('func_val_conv', 'func_coupl_conv', 'func_constr_conv'):
        assert np.allclose(og[k], oo[k], rtol=1e-7, atol=1e-10), k


@pytest.mark.parametrize('dims,R', [((40, 50, 60), 3), ((20, 30, 40), 3), ((33, 17, 29), 5)])
def test_cp_nonneg(pkg, eng, dims, R):
    """config 1 (CP part, both readings of the shape) and config 2 at oracle-sized dims."""
    rng = np.random.default_rng(1)
    Z, io, _ = cp_model(dims, R, rng, [('non-negativity',)] * 3)
    compare(*run_both(pkg, eng, Z, io, options(MaxOuterIters=20)))


def test_cp_tv_nonneg(pkg, eng):
    """config 5 shape family: TV on mode 1 (example_script10_CP_TVreg.m:55), non-negativity on modes 2-3."""
    rng = np.random.default_rng(2)
    Z, io, _ = cp_model((60, 50, 70), 3, rng, [('TV regularization', 0.001), ('non-negativity',), ('non-negativity',)])
    compare(*run_both(pkg, eng, Z, io, options(MaxOuterIters=15)))


@pytest.mark.parametrize('rows', [4500, 7000])
def test_cp_tv_long_mode(pkg, eng, rows):
    """TV on a long mode inside the ADMM loop (prox + dual update + residual sums, warm-started from the previous Z):
    4500 rows with eight entries per thread in LDS, 7000 rows with part of the working arrays in the prox workspace."""
    rng = np.random.default_rng(12)
    Z, io, _ = cp_model((rows, 9, 8), 3, rng, [('TV regularization', 0.01), ('non-negativity',), ('non-negativity',)])
    compare(*run_both(pkg, eng, Z, io, options(MaxOuterIters=6)))


def test_cp_quadratic_nonsymmetric(pkg, eng):
    """{'quadratic regularization', eta, L} with a NON-symmetric L (constraints_to_prox.m:62-67 solves
    (2 eta/rho L + I) \\ x for any L): no eigenbasis, the library inverts the matrix on the device (Gauss-Jordan with row
    pivoting, rho read from device memory: no host round trip) once per outer iteration and the prox is one GEMM.  Solver parity at 1e-8, op-level at 1e-11, and a symmetric L right after it on the
    same engine (the two preparations must not leak into each other)."""
    rng = np.random.default_rng(52)
    n = 41
    L = np.triu(rng.random((n, n)), 1) * 0.6 + np.diag(1.0 + rng.random(n)) - 0.2 * np.tril(rng.random((n, n)), -1)
    assert np.linalg.norm(L - L.T) > 0.1 * np.linalg.norm(L)
    Z, io, _ = cp_model((n, 18, 15), 3, rng, [('quadratic regularization', 0.05, L), ('non-negativity',), None])
    compare(*run_both(pkg, eng, Z, io, options(MaxOuterIters=10)))
    x = rng.standard_normal((n, 5))
    for rho in (0.7, 0.7, 2.5):                          # same rho twice: the cached inverse is reused
        got = eng.prox(('quadratic regularization', 0.05, L), x, rho)
        ref = np.linalg.solve(2.0 * 0.05 / rho * L + np.eye(n), x)
        assert rel_fro(got, ref) < 1e-11
    Ls = L + L.T
    got = eng.prox(('quadratic regularization', 0.05, Ls), x, 0.7)
    assert rel_fro(got, np.linalg.solve(2.0 * 0.05 / 0.7 * Ls + np.eye(n), x)) < 1e-11
    # a coupled/permuted matrix that needs row exchanges: zero diagonal after scaling
    P = np.roll(np.eye(n), 1, axis=0) * 50.0 - np.eye(n) * (0.7 / (2.0 * 0.05))     # 2 eta/rho L + I has a zero diagonal
    got = eng.prox(('quadratic regularization', 0.05, P), x, 0.7)
    assert rel_fro(got, np.linalg.solve(2.0 * 0.05 / 0.7 * P + np.eye(n), x)) < 1e-11


def test_quadratic_nonsymmetric_larger_matrix(eng):
    """The device-side pivoted inverse beyond one workgroup's worth of rows (several row tiles, ragged column tiles):
    n = 700 rows, non-symmetric L, against numpy's LU solve at 1e-10; a second rho reuses nothing."""
    rng = np.random.default_rng(53)
    n = 700
    L = rng.standard_normal((n, n)) / np.sqrt(n) + np.diag(1.0 + rng.random(n))
    x = rng.standard_normal((n, 6))
    for rho in (1.3, 0.4):
        got = eng.prox(('quadratic regularization', 0.3, L), x, rho)
        ref = np.linalg.solve(2.0 * 0.3 / rho * L + np.eye(n), x)
        assert rel_fro(got, ref) < 1e-10, rel_fro(got, ref)


def test_natural_copy_released_after_the_pass_copies(pkg, monkeypatch):
    """Engine::maybe_release_natural: once the three pass copies exist the natural-layout array is freed (by default only when
    HBM is tight: a 2000^3 double tensor; forced here).  The solve is unchanged (1e-8 vs the oracle, and bit-identical to the
    run that keeps the array), a second solve on the same context works, HBM use drops by one copy of the tensor, the
    resident unfold-Gram entry answers 'unsupported' (the Python layer falls back to the host array) and Z.miss can no longer
    be attached without uploading the data again."""
    import ctypes as C
    capi = __import__('importlib').import_module('matlab-code_amd._capi')
    rng = np.random.default_rng(77)
    dims = (120, 90, 80)                               # 864 000 entries: beyond the one-launch MTTKRP of tiny blocks
    Z, io, _ = cp_model(dims, 4, rng, [('non-negativity',), ('TV regularization', 0.01), None])
    G = OA.init_coupled_AOADMM_CMTF({**Z, 'prox_operators': None}, io, rng=np.random.default_rng(5))
    opt = options(MaxOuterIters=6)
    _, Fo, _, oo = OA.cmtf_AOADMM(Z, alg_options=opt, init=copy.deepcopy(G))
    monkeypatch.setenv('AOADMM_RELEASE_NATURAL', '0')
    with pkg.Engine(0) as e:
        _, Fk, _, ok_ = pkg.cmtf_AOADMM(Z, alg_options=opt, init=copy.deepcopy(G), engine=e)
        Y = e.resident_unfold_gram(0, 1, dims[1])
    monkeypatch.setenv('AOADMM_RELEASE_NATURAL', '1')
    with pkg.Engine(0) as e:
        _, Fg, _, og = pkg.cmtf_AOADMM(Z, alg_options=opt, init=copy.deepcopy(G), engine=e)
        for a, b, c in zip(Fo['fac'], Fg['fac'], Fk['fac']):
            assert rel_fro(b, a) < 1e-8
            assert np.array_equal(b, c)
        with pytest.raises(capi.UnsupportedOnDevice):
            e.resident_unfold_gram(0, 1, dims[1])
        # the Python layer falls back to the host-array form
        Z2 = dict(Z); Z2['_ranks'] = [4] * 3
        assert e._resident_model is not None
        from importlib import import_module
        drv = import_module('matlab-code_amd.driver')
        Y2 = drv.cmtf_nvecs(e._resident_model, 1, 4, e)
        assert Y2.shape == (dims[1], 4)
        mk = np.ones(dims, dtype=np.uint8, order='F')
        st = e.lib.aoadmm_tensor_mask_upload(e.h, 0, mk.ctypes.data_as(C.POINTER(C.c_uint8)))
        assert st == capi.ERR_INVALID and 'released' in e.lib.aoadmm_last_error().decode()
        # a second solve on the same context (only the pass copies are resident)
        pkg.upload_state(e, e._resident_model, copy.deepcopy(G))
        out2 = pkg.run_solver(e, opt, 3)
        assert np.allclose(out2['func_val_conv'], oo['func_val_conv'], rtol=1e-7)
    assert Y.shape == (dims[1], dims[1])


def test_cp_mixed_constraints_and_ls(pkg, eng):
    rng = np.random.default_rng(3)
    Z, io, _ = cp_model((30, 25, 20), 4, rng, [None, ('l2-ball', 1.0), ('unimodality', True)])
    Z['ridge'] = [1e-3, 1e-3, 1e-3]
    compare(*run_both(pkg, eng, Z, io, options(MaxOuterIters=10, bsum=1, bsum_weight=1e-3)))


def test_script3_partial_coupling(pkg, eng):
    """config 3: example_script3 shapes, coupling type 4."""
    rng = np.random.default_rng(4)
    Z, io = script3_model(rng)
    compare(*run_both(pkg, eng, Z, io, options(MaxOuterIters=15)))


def test_cp_cp_exact_coupling(pkg, eng):
    rng = np.random.default_rng(5)
    Z, io = cp_cp_exact_model(rng)
    compare(*run_both(pkg, eng, Z, io, options(MaxOuterIters=15)))


@pytest.mark.parametrize('R', [7, 24, 32])
def test_row_loop_on_the_matrix_cores_rank_classes(pkg, eng, R):
    """admm_rows_mfma_k (modes longer than 256 rows, element-wise prox) in its smallest, a middle and its largest rank
    class (R <= 32: eight 4-column steps per row)."""
    rng = np.random.default_rng(70 + R)
    Z, io, _ = cp_model((300, 41, 37), R, rng, [('non-negativity',), ('box', 0.0, 0.5), ('l1 regularization', 0.001)])
    compare(*run_both(pkg, eng, Z, io, options(MaxOuterIters=5)))


@pytest.mark.parametrize('R', [9, 12, 16])
def test_short_modes_at_larger_ranks(pkg, eng, R):
    """The one-workgroup loop of a short mode (admm_loop_wg_k) and the one-launch MTTKRP of a tiny block (small_mttkrp_k)
    in their widest register class (9 <= R <= 16), with a row-wise, a column-norm and an element-wise prox."""
    rng = np.random.default_rng(40 + R)
    Z, io, _ = cp_model((60, 45, 35), R, rng, [('simplex row-wise', 1.0), ('non-negative l2-sphere', 1), ('box', 0.0, 0.6)])
    compare(*run_both(pkg, eng, Z, io, options(MaxOuterIters=6)))
    Fo, oo, Fg, og = run_both(pkg, eng, Z, io, options(MaxOuterIters=4), precision='f32')
    for a, b in zip(Fo['fac'], Fg['fac']):
        assert rel_fro(b, a) < 1e-4


@pytest.mark.parametrize('ctype,R', [(0, 6), (0, 8), (4, 7), (4, 8)])
def test_row_local_coupling_loop_register_classes(pkg, eng, ctype, R):
    """couple_loop_wg_regs_k with 5..8 rank columns (its second register class), exact coupling and C = Delta*H."""
    rng = np.random.default_rng(50 + R)
    if ctype == 0:
        Z, io = cp_cp_exact_model(rng, rows=57, R=R)
    else:
        Z, io = script3_model(rng, rows=57, R=R)
    compare(*run_both(pkg, eng, Z, io, options(MaxOuterIters=6)))


@pytest.mark.parametrize('rows', [40, 300])
def test_three_blocks_share_a_factor(pkg, eng, rows):
    """Three CP tensors coupled exactly in their first modes (type 0): the one-workgroup loops with three coupled modes
    (register form at 40 rows, memory form at 300)."""
    from oracle.tensor_ops import full_ktensor
    rng = np.random.default_rng(60 + rows)
    R = 3
    D = rng.random((rows, R))
    shapes = [(14, 11), (9, 16), (12, 10)]
    objs = []
    for a, b in shapes:
        X = full_ktensor([D, rng.random((a, R)), rng.random((b, R))])
        N = rng.standard_normal(X.shape)
        X += 0.05 * np.linalg.norm(X) / np.linalg.norm(N) * N
        objs.append(X / np.linalg.norm(X))
    size = []
    for a, b in shapes:
        size += [rows, a, b]
    Z = dict(loss_function=['Frobenius'] * 3, model=['CP'] * 3, modes=[[1, 2, 3], [4, 5, 6], [7, 8, 9]], size=size,
             coupling=dict(lin_coupled_modes=[1, 0, 0, 1, 0, 0, 1, 0, 0], coupling_type=[0], coupl_trafo_matrices=[None] * 9),
             constrained_modes=[1, 0, 1, 1, 0, 0, 0, 1, 0],
             constraints=[('non-negativity',), None, ('non-negativity',), ('box', 0.0, 2.0), None, None, None, ('non-negativity',), None],
             weights=[1 / 3] * 3, object=objs)
    io = dict(lambdas_init=[[1] * R] * 3, nvecs=0, distr=[lambda a, b: rng.random((a, b))] * 9, normalize=1)
    compare(*run_both(pkg, eng, Z, io, options(MaxOuterIters=8)))


@pytest.mark.parametrize('rows', [300, 2500])
@pytest.mark.parametrize('ctype', [0, 4])
def test_row_local_coupling_loop_forms(pkg, eng, rows, ctype):
    """The three forms of the row-local coupled loop (csrc/solver.hip): register-resident one-workgroup kernel (rows
    <= 256: the script-sized tests above), global-memory one-workgroup kernel (300 rows here), one launch per step
    (2500 rows), for the exact (type 0) and the partial (type 4) coupling."""
    rng = np.random.default_rng(40 + ctype)
    Z, io = (cp_cp_exact_model(rng, rows=rows) if ctype == 0 else script3_model(rng, rows=rows))
    compare(*run_both(pkg, eng, Z, io, options(MaxOuterIters=6)))


def test_coupled_loop_early_exit_matches(pkg, eng):
    """Non-zero inner tolerances on a coupled model: the one-workgroup loop stops after the same inner iteration as the
    oracle's ADMM_coupled_case4 (innerIters compared), and the state it leaves is the state of that iteration."""
    rng = np.random.default_rng(44)
    Z, io = script3_model(rng)
    opt = options(MaxOuterIters=25, MaxInnerIters=10, innerRelPrTol_coupl=1e-2, innerRelDualTol_coupl=1e-2,
                  innerRelPrTol_constr=1e-2, innerRelDualTol_constr=1e-2)
    Fo, oo, Fg, og = run_both(pkg, eng, Z, io, opt)
    assert np.array_equal(og['innerIters'], oo['innerIters'])
    compare(Fo, oo, Fg, og)


def test_early_stop_matches(pkg, eng):
    """Non-zero tolerances: inner/outer stopping decisions taken on the device agree with the oracle."""
    rng = np.random.default_rng(6)
    Z, io, _ = cp_model((20, 30, 40), 3, rng, [('non-negativity',)] * 3, noise=0.0)
    opt = options(MaxOuterIters=300, AbsFuncTol=1e-9, OuterRelTol=1e-8, innerRelPrTol_constr=1e-3, innerRelDualTol_constr=1e-3)
    Fo, oo, Fg, og = run_both(pkg, eng, Z, io, opt)
    assert og['OuterIterations'] == oo['OuterIterations']
    assert og['exit_flag'] == oo['exit_flag']
    assert np.array_equal(og['innerIters'], oo['innerIters'])
    for a, b in zip(Fo['fac'], Fg['fac']):
        assert rel_fro(b, a) < 1e-6


def test_fp32_tensor_mode(pkg, eng):
    """Throughput mode (fp32 tensor, f32 MFMA, fp64 elsewhere): stated tolerance 1e-4 on the factors
    after 10 outer iterations (input rounding 6e-8 amplified by the iteration)."""
    rng = np.random.default_rng(8)
    Z, io, _ = cp_model((40, 50, 60), 3, rng, [('non-negativity',)] * 3)
    Fo, oo, Fg, og = run_both(pkg, eng, Z, io, options(MaxOuterIters=10), precision='f32')
    for a, b in zip(Fo['fac'], Fg['fac']):
        assert rel_fro(b, a) < 1e-4


def compare_par2(Fo, oo, Fg, og, tol=TOL):
    def each(a, b, key):
        if a is None:
            return
        if isinstance(a, (list, tuple)):
            for x, y in zip(a, b):
                each(x, y, key)
        elif isinstance(a, dict):
            for k in a:
                each(a[k], b[k], key)
        else:
            assert rel_fro(b, a) < tol, (key, rel_fro(b, a))
    for key in ('fac', 'constraint_fac', 'constraint_dual_fac', 'coupling_fac', 'coupling_dual_fac', 'DeltaB', 'P', 'mu_DeltaB'):
        each(Fo[key], Fg[key], key)
    assert og['OuterIterations'] == oo['OuterIterations']
    assert np.array_equal(og['innerIters'], oo['innerIters'])
    for k in ('func_val_conv', 'func_coupl_conv', 'func_constr_conv', 'func_PAR2_coupl'):
        assert np.allclose(og[k], oo[k], rtol=1e-7, atol=1e-10), (k, og[k], oo[k])


def test_script4_irregular_parafac2(pkg, eng):
    """config 4 family (example_script4: I=40, ragged J_k in 61..120, R=3, C non-negative), K = 12 slabs."""
    from helpers import script4_model
    rng = np.random.default_rng(10)
    Z, io = script4_model(rng, K=12)
    compare_par2(*run_both(pkg, eng, Z, io, options(MaxOuterIters=12)))


@pytest.mark.parametrize('R,K', [(6, 9), (8, 70), (9, 5)])
def test_parafac2_larger_ranks(pkg, eng, R, K):
    """The B_k loop with its sums folded into the neighbouring kernels exists for R <= 8 in two register classes (R*R <= 16
    and <= 64 partial sums per lane; more slabs than lanes at K = 70); R = 9 takes the four-launch form."""
    from helpers import script4_model
    rng = np.random.default_rng(13)
    Z, io = script4_model(rng, K=K, R=R)
    compare_par2(*run_both(pkg, eng, Z, io, options(MaxOuterIters=6)))


def test_parafac2_constrained_Bk(pkg, eng):
    """B_k constrained (example_script9 family: unimodality on the B_k columns, delayed start, rho factor)."""
    from helpers import script4_model
    rng = np.random.default_rng(11)
    Z, io = script4_model(rng, K=6, constraints_B=('unimodality', False))
    opt = options(MaxOuterIters=8, iter_start_PAR2Bkconstraint=3, increase_factor_rhoBk=2.0)
    compare_par2(*run_both(pkg, eng, Z, io, opt))


@pytest.mark.parametrize('dims', [(20, 30, 40), (40, 50, 60)])
def test_script1_cp_parafac2_coupled(pkg, eng, dims):
    """config 1 (example_script1: CP + PARAFAC2, first modes exactly coupled, non-negativity), both shape readings."""
    from helpers import script1_model
    rng = np.random.default_rng(12)
    Z, io = script1_model(rng, dims=dims)
    compare_par2(*run_both(pkg, eng, Z, io, options(MaxOuterIters=10)))


def test_tparafac2_temporal_smoothness(pkg, eng):
    """example_script11 family: regular PARAFAC2 with {'tPARAFAC2', eta} on the B_k mode
    (t_smoothness_prox.m tridiagonal solve across slabs, t_smoothness_penalty.m in f_tensors)."""
    from helpers import par2_slabs
    rng = np.random.default_rng(13)
    I, R, K, J = 25, 3, 9, 31
    X, _ = par2_slabs(I, [J] * K, R, rng, noise=0.1)
    Z = dict(loss_function=['Frobenius'], model=['PAR2'], modes=[[1, 2, 3]], size=[I, [J] * K, K],
             coupling=dict(lin_coupled_modes=[0, 0, 0], coupling_type=[], coupl_trafo_matrices=[None] * 3),
             constrained_modes=[0, 1, 1], constraints=[None, ('tPARAFAC2', 0.5), ('non-negativity',)],
             weights=[1.0], object=[X])
    distr = [lambda a, b: rng.standard_normal((a, b)), lambda a, b: rng.standard_normal((a, b)),
             lambda a, b: rng.random((a, b)) + 0.1]
    io = dict(lambdas_init=[[1] * R], nvecs=0, distr=distr, normalize=1)
    opt = options(MaxOuterIters=10, iter_start_PAR2Bkconstraint=2, increase_factor_rhoBk=1.5)
    compare_par2(*run_both(pkg, eng, Z, io, opt))


def test_rccl_one_rank_communicator(pkg):
    """The N > 1 data path (zero-filled own-rows buffer + ncclAllReduce of every MTTKRP output on the
    library's stream, csrc/solver.hip block_mttkrp/allreduce) with a ONE-rank RCCL communicator: the
    only form of the RCCL path a one-GPU box can run.  Same factors as the oracle."""
    rng = np.random.default_rng(21)
    Z, io, _ = cp_model((37, 14, 12), 3, rng, [('TV regularization', 0.01), ('non-negativity',), ('non-negativity',)])
    with pkg.Engine(0) as e1:
        e1.comm_init_rank(e1.comm_unique_id(), 0, 1)
        compare(*run_both(pkg, e1, Z, io, options(MaxOuterIters=6)))
        Fo, oo, Fg, og = run_both(pkg, e1, Z, io, options(MaxOuterIters=4), precision='f32')
        for a, b in zip(Fo['fac'], Fg['fac']):
            assert rel_fro(b, a) < 1e-4


def _with_mask(Z, rng, frac=0.2):
    """~20 % of the entries of every block missing at random, initialised with 0 (example_script12_CP_PAR2_EM.m:115-147)."""
    Z = dict(Z)
    Z['object'] = list(Z['object'])
    miss = []
    for p, obj in enumerate(Z['object']):
        if Z['model'][p] == 'CP':
            X = np.array(obj, dtype=float)
            mask = np.ones(X.shape, dtype=bool)
            mask.flat[rng.choice(mask.size, int(frac * mask.size), replace=False)] = False
            X[~mask] = 0.0
            Z['object'][p] = X
            miss.append(mask)
        else:
            Xs, ms = [], []
            for Xk in obj:
                Xk = np.array(Xk, dtype=float)
                mk = np.ones(Xk.shape, dtype=bool)
                mk.flat[rng.choice(mk.size, int(frac * mk.size), replace=False)] = False
                Xk[~mk] = 0.0
                Xs.append(Xk); ms.append(mk)
            Z['object'][p] = Xs
            miss.append(ms)
    Z['miss'] = miss
    return Z


def _compare_em(oo, og):
    assert np.allclose(og['func_rel_missing'][1:], oo['func_rel_missing'][1:], rtol=1e-7, atol=1e-12)
    assert np.isnan(og['func_rel_missing'][0])
    assert abs(og['f_rel_missing'] - oo['f_rel_missing']) <= 1e-7 * abs(oo['f_rel_missing']) + 1e-12


def test_em_missing_cp(pkg, eng):
    """EM imputation (cmtf_fun_AOADMM.m:408-441) on a CP tensor, ragged first mode (padding rows), TV + non-negativity."""
    rng = np.random.default_rng(31)
    Z, io, _ = cp_model((37, 22, 19), 3, rng, [('non-negativity',), ('non-negativity',), ('TV regularization', 1e-3)])
    Z = _with_mask(Z, rng)
    Fo, oo, Fg, og = run_both(pkg, eng, Z, io, options(MaxOuterIters=12))
    compare(Fo, oo, Fg, og)
    _compare_em(oo, og)


@pytest.mark.parametrize('prec', ['f64', 'f32'])
def test_em_fused_contraction_of_the_second_mode(pkg, eng, prec):
    """The EM pass can leave the partial contraction of mode 3 (strip walks mode 3) or of mode 2 (walks mode 2); the
    update order 1-2-3 always picks mode 3.  With the test hook the other walk is taken: it serves the first mode only
    (the cache then misses for the second, correctly), so the factors must still match the oracle."""
    rng = np.random.default_rng(36)
    Z, io, _ = cp_model((45, 150, 21), 6, rng, [('non-negativity',)] * 3)
    Z = _with_mask(Z, rng)
    os.environ['AOADMM_EM_FUSE_SECOND_MODE'] = '1'
    try:
        Fo, oo, Fg, og = run_both(pkg, eng, Z, io, options(MaxOuterIters=5), precision=prec)
    finally:
        os.environ.pop('AOADMM_EM_FUSE_SECOND_MODE', None)
    for a, b in zip(Fo['fac'], Fg['fac']):
        assert rel_fro(b, a) < (1e-4 if prec == 'f32' else 1e-8)
    assert np.allclose(og['func_rel_missing'][1:], oo['func_rel_missing'][1:], rtol=1e-3 if prec == 'f32' else 1e-7)


def test_em_missing_matrix_and_stop_rule(pkg, eng):
    """Matrix block (transposed copy imputed as well) + the extra stopping rule f_rel_missing < OuterRelTol (:457-459)."""
    rng = np.random.default_rng(32)
    Z, io, _ = cp_model((30, 26), 3, rng, [('non-negativity',), None], noise=0.0)
    Z = _with_mask(Z, rng, frac=0.1)
    opt = options(MaxOuterIters=400, AbsFuncTol=1e-4, OuterRelTol=1e-3)
    Fo, oo, Fg, og = run_both(pkg, eng, Z, io, opt)
    assert og['OuterIterations'] == oo['OuterIterations'] and og['OuterIterations'] < 400
    for a, b in zip(Fo['fac'], Fg['fac']):
        assert rel_fro(b, a) < 1e-6
    _compare_em(oo, og)


def test_em_missing_cp_parafac2_script12(pkg, eng):
    """example_script12_CP_PAR2_EM.m: coupled CP + PARAFAC2, ~20 % missing in both blocks."""
    from helpers import script1_model
    rng = np.random.default_rng(33)
    Z, io = script1_model(rng, dims=(20, 30, 40), K=8, Jk=30, noise=0.05)
    Z = _with_mask(Z, rng)
    Fo, oo, Fg, og = run_both(pkg, eng, Z, io, options(MaxOuterIters=10))
    compare_par2(Fo, oo, Fg, og)
    _compare_em(oo, og)


def test_em_missing_fp32_tensor(pkg, eng):
    """fp32-resident tensor: imputation and statistics in fp32/fp64 mix, stated tolerance 1e-4."""
    rng = np.random.default_rng(34)
    Z, io, _ = cp_model((41, 33, 28), 3, rng, [('non-negativity',)] * 3)
    Z = _with_mask(Z, rng)
    Fo, oo, Fg, og = run_both(pkg, eng, Z, io, options(MaxOuterIters=8), precision='f32')
    for a, b in zip(Fo['fac'], Fg['fac']):
        assert rel_fro(b, a) < 1e-4
    assert np.allclose(og['func_rel_missing'][1:], oo['func_rel_missing'][1:], rtol=1e-3)


@pytest.mark.parametrize('dims,R,prec', [((70, 200, 30), 5, 'f64'), ((70, 200, 30), 5, 'f32'),
                                         ((9, 400, 500), 20, 'f32'), ((130, 300), 7, 'f64'),
                                         ((24, 70, 9), 33, 'f64'), ((24, 70, 9), 33, 'f32'),     # R > 32: one row per thread
                                         ((130, 70, 40), 3, 'f32'), ((130, 70, 40), 10, 'f64'), ((130, 70, 40), 14, 'f32'),
                                         ((130, 70, 40), 18, 'f64'), ((130, 70, 40), 24, 'f32'), ((130, 70, 40), 30, 'f64')])
def test_em_missing_column_pieces(pkg, eng, dims, R, prec):
    """The EM pass cuts the second mode into pieces of whole 64-column tiles (em.hip: em_chunking): several pieces per
    slab with a short last one, several tiles per piece, a ragged first mode, ranks in different register classes."""
    rng = np.random.default_rng(35)
    Z, io, _ = cp_model(dims, R, rng, [('non-negativity',)] + [None] * (len(dims) - 1))
    Z = _with_mask(Z, rng)
    Fo, oo, Fg, og = run_both(pkg, eng, Z, io, options(MaxOuterIters=4), precision=prec)
    tol = 1e-4 if prec == 'f32' else 1e-8
    for a, b in zip(Fo['fac'], Fg['fac']):
        assert rel_fro(b, a) < tol
    assert np.allclose(og['func_rel_missing'][1:], oo['func_rel_missing'][1:], rtol=1e-3 if prec == 'f32' else 1e-7)


@pytest.mark.parametrize('cB', [('GL smoothness', 1.0), ('TV regularization', 0.01), ('l1 regularization', 0.01),
                                ('ridge', 0.1), ('l2 regularization', 0.05)])
def test_parafac2_regularised_Bk(pkg, eng, cB):
    """example_script1a family: a regularisation-type constraint on the PARAFAC2 B_k mode ('GL smoothness' there);
    its value sum_k reg_func(B_k) enters f_tensors (cmtf_fun_AOADMM.m:1279-1281)."""
    from helpers import par2_slabs
    rng = np.random.default_rng(41)
    I, R, K, J = 30, 3, 5, 44          # regular PARAFAC2: the reference builds ONE Laplacian from the first slab size
    X, _ = par2_slabs(I, [J] * K, R, rng, noise=0.1)             # (constraints_to_prox.m:70), so GL needs equal J_k
    Z = dict(loss_function=['Frobenius'], model=['PAR2'], modes=[[1, 2, 3]], size=[I, [J] * K, K],
             coupling=dict(lin_coupled_modes=[0, 0, 0], coupling_type=[], coupl_trafo_matrices=[None] * 3),
             constrained_modes=[0, 1, 1], constraints=[None, cB, ('non-negativity',)], weights=[1.0], object=[X])
    distr = [lambda a, b: rng.standard_normal((a, b)), lambda a, b: rng.standard_normal((a, b)),
             lambda a, b: rng.random((a, b)) + 0.1]
    io = dict(lambdas_init=[[1] * R], nvecs=0, distr=distr, normalize=1)
    compare_par2(*run_both(pkg, eng, Z, io, options(MaxOuterIters=8)))


def test_cp_quadratic_regularization(pkg, eng):
    """{'quadratic regularization', eta, L} with a dense symmetric L (constraints_to_prox.m:62-67): the device prox
    uses L = U diag(w) U' (diagonalised once on the host), the oracle solves (2 eta/rho L + I) \\ x directly."""
    rng = np.random.default_rng(51)
    n = 35
    W = rng.random((n, n)); W = np.triu(W, 1); W = W + W.T
    L = np.diag(W.sum(axis=1)) - W                       # weighted graph Laplacian
    Z, io, _ = cp_model((n, 22, 19), 3, rng, [('quadratic regularization', 0.02, L), ('non-negativity',), None])
    compare(*run_both(pkg, eng, Z, io, options(MaxOuterIters=12)))
    x = rng.standard_normal((n, 4))
    got = eng.prox(('quadratic regularization', 0.02, L), x, 0.7)
    ref = np.linalg.solve(2 * 0.02 / 0.7 * L + np.eye(n), x)
    assert rel_fro(got, ref) < 1e-12


@pytest.mark.parametrize('ctype', [1, 2, 3, 5])
def test_transformed_couplings(pkg, eng, ctype):
    """Coupling types 1 (H*C = Delta, example_script5), 2 (C*H = Delta), 3 (C = H*Delta) and 5 (H*C = Delta*H2,
    example_script13): cmtf_fun_AOADMM.m:698-901, :986-1075.  Types 1/5 solve a Sylvester equation per mode (MATLAB
    `sylvester`; here through the eigenbases of H'H and of the R x R system)."""
    from helpers import transformed_coupling_model
    rng = np.random.default_rng(60 + ctype)
    Z, io = transformed_coupling_model(rng, ctype)
    # type 5: init_coupled_AOADMM_CMTF.m:160-164 sizes coupling_fac from the Delta it is handed
    Delta = [np.zeros((25, 4))] if ctype == 5 else None
    compare(*run_both(pkg, eng, Z, io, options(MaxOuterIters=12), Delta=Delta))


def test_cp_four_way(pkg, eng):
    """A 4-way CP block through the solver (no example script uses one, the reference API allows it)."""
    rng = np.random.default_rng(91)
    Z, io, _ = cp_model((12, 9, 8, 7), 3, rng, [('non-negativity',), None, ('l2-ball', 1.0), ('non-negativity',)])
    compare(*run_both(pkg, eng, Z, io, options(MaxOuterIters=10)))


@pytest.mark.parametrize('dims,prec,frac', [((1100, 9, 7), 'f32', 0.01), ((1099, 9, 7), 'f64', 0.01),
                                            ((1029, 11, 5), 'f32', 0.003), ((523, 6, 7), 'f64', 0.05),
                                            ((2051, 13), 'f32', 0.01)])
def test_em_sparse_missing_line_writeback(pkg, eng, dims, prec, frac):
    """Few missing entries: the EM pass writes back only the 128-byte lines that hold one (em.hip), found with a wave
    ballot over the one-bit-per-entry mask; several waves per strip, first modes whose padded length puts the columns
    at every 16-byte slot of a line (1100 -> 4400 bytes per column, 1029 -> 1032 floats), fp32 and fp64 vectors.
    The imputed entries enter every later iteration, so the factors compared after 4 iterations catch a line that was
    not written (or a clean one that was changed)."""
    rng = np.random.default_rng(int(1000 * frac) + dims[0])
    Z, io, _ = cp_model(dims, 3, rng, [('non-negativity',)] + [None] * (len(dims) - 1))
    Z = _with_mask(Z, rng, frac=frac)
    Fo, oo, Fg, og = run_both(pkg, eng, Z, io, options(MaxOuterIters=4), precision=prec)
    tol = 1e-4 if prec == 'f32' else 1e-8
    for a, b in zip(Fo['fac'], Fg['fac']):
        assert rel_fro(b, a) < tol
    assert np.allclose(og['func_rel_missing'][1:], oo['func_rel_missing'][1:], rtol=1e-3 if prec == 'f32' else 1e-7)


def test_parafac2_C_mode_coupling_dense_system(pkg, eng):
    """Type 1 with an H whose H'H is NOT diagonal (means of neighbouring rows): the (K*R) x (K*R) system stays dense --
    one-workgroup Cholesky + column-parallel inverse (csrc/small.hip dense_spd_inverse).  Selection matrices (script 14)
    take the per-row path instead, see test_parafac2_C_mode_coupling."""
    from helpers import par2_C_coupled_model
    for constr6 in (True, False):
        rng = np.random.default_rng(181)
        Z, io = par2_C_coupled_model(rng, 1, noise=0.05, average=True)
        if not constr6:
            Z['constrained_modes'][5] = 0
            Z['constraints'][5] = None
        compare_par2(*run_both(pkg, eng, Z, io, options(MaxOuterIters=10)))


@pytest.mark.parametrize('ctype,constr6', [(0, True), (1, True), (1, False)])
def test_parafac2_C_mode_coupling(pkg, eng, ctype, constr6):
    """example_script14 family: first mode of a CP tensor coupled to the C mode of a PARAFAC2 block.  Type 0 = per-row
    systems with row-wise Delta weights rho_k (cmtf_fun_AOADMM.m:260-267, :638-645, :666-675); type 1 = H*C = Delta
    through the (K*R) x (K*R) system (:282-297, :710-724), with and without a constraint on the C mode."""
    from helpers import par2_C_coupled_model
    rng = np.random.default_rng(150 + ctype)
    Z, io = par2_C_coupled_model(rng, ctype, noise=0.05)
    if not constr6:
        Z['constrained_modes'][5] = 0
        Z['constraints'][5] = None
    compare_par2(*run_both(pkg, eng, Z, io, options(MaxOuterIters=12)))


@pytest.mark.parametrize('ctype', [2, 3, 4, 5])
def test_parafac2_C_mode_transformed_coupling(pkg, eng, ctype):
    """A PARAFAC2 C mode in couplings of type 2 (C*H = Delta: + rho_k/2*H*H' in the row systems, :305-312, :785-792),
    3 (C = H*Delta: Delta from H'*diag(rho)*H, :327-334, :875-885), 4 (C = Delta*H: one q x q system per row of
    Delta, AA + rho_k*AAA, :349-356, :945-961) and 5 (H*C = Delta*H2: the (K*R)^2 system of type 1 for C, the per-row
    systems of type 4 for Delta, :371-385, :998-1052)."""
    from helpers import par2_C_transformed_model
    rng = np.random.default_rng(170 + ctype)
    Z, io = par2_C_transformed_model(rng, ctype)
    Delta = [np.zeros((14, 4))] if ctype == 4 else ([np.zeros((7, 4))] if ctype == 5 else None)
    compare_par2(*run_both(pkg, eng, Z, io, options(MaxOuterIters=12), Delta=Delta))


def test_display_iter_reports_live(pkg, eng, capsys):
    """options.Display = 'iter' (cmtf_fun_AOADMM.m:44-59, :462-468, :496-503): header, iteration 0, every DisplayIters-th
    iteration from inside the solve (aoadmm_set_progress), and the final row; the numbers are those of `out`."""
    rng = np.random.default_rng(3)
    Z, io, _ = cp_model((20, 12, 9), 3, rng, [('non-negativity',)] * 3)
    G = OA.init_coupled_AOADMM_CMTF({**Z, 'prox_operators': None}, io, rng=np.random.default_rng(7))
    opt = options(MaxOuterIters=7, Display='iter', DisplayIters=2)
    _, _, _, out = pkg.cmtf_AOADMM(Z, alg_options=opt, init=copy.deepcopy(G), engine=eng)
    lines = [ln for ln in capsys.readouterr().out.splitlines() if ln.strip()]
    assert lines[0].startswith(' Iter  f total') and lines[1].startswith('------')
    rows = [ln.split() for ln in lines[2:]]
    assert [int(r[0]) for r in rows] == [0, 2, 4, 6, 7]                 # 0, multiples of DisplayIters, final
    for r in rows:
        it = int(r[0])
        assert abs(float(r[2]) - out['func_val_conv'][it]) < 5e-7       # printed with 6 decimals
        assert abs(float(r[4]) - out['func_constr_conv'][it]) < 5e-7


@pytest.mark.parametrize('family', ['cp_tv_f32', 'script1_par2', 'script3_coupled'])
def test_run_to_run_bitwise_reproducible(pkg, eng, family):
    """Every reduction in the library has a fixed summation order (block partials added in order, no floating-point
    atomics), so two solves from the same struct must agree bit for bit -- the determinism SURVEY section 5 asks for."""
    from helpers import script1_model
    rng = np.random.default_rng(77)
    prec = 'f64'
    if family == 'cp_tv_f32':
        Z, io, _ = cp_model((131, 37, 29), 5, rng, [('TV regularization', 0.01), ('non-negativity',), ('simplex column-wise', 1.0)])
        prec = 'f32'
    elif family == 'script1_par2':
        Z, io = script1_model(rng, dims=(20, 30, 40))
    else:
        Z, io = script3_model(rng)
    G = OA.init_coupled_AOADMM_CMTF({**Z, 'prox_operators': None}, io, rng=np.random.default_rng(7))
    runs = []
    for _ in range(2):
        _, F, _, o = pkg.cmtf_AOADMM(Z, alg_options=options(MaxOuterIters=8), init=copy.deepcopy(G), engine=eng, precision=prec)
        runs.append((F, o))

    def same(a, b):
        if a is None:
            return True
        if isinstance(a, (list, tuple)):
            return all(same(x, y) for x, y in zip(a, b))
        if isinstance(a, dict):
            return all(same(a[k], b[k]) for k in a)
        return np.array_equal(a, b)
    for key in ('fac', 'constraint_fac', 'constraint_dual_fac', 'coupling_fac', 'coupling_dual_fac'):
        assert same(runs[0][0][key], runs[1][0][key]), key
    assert np.array_equal(runs[0][1]['func_val_conv'], runs[1][1]['func_val_conv'])
