"""GPU parity on randomly drawn models: every seed draws a model from the space the reference's interface allows --
one or two CP blocks of order 2-4, optionally a PARAFAC2 block, constraints drawn from the catalogue of
constraints_to_prox.m, first modes optionally coupled (type 0, or type 4 with a column-selecting H), random weights,
Z.ridge and bsum on or off, 3-6 inner iterations -- and compares the device solve with the oracle from the same
initial struct.  The named tests cover the example scripts; this one looks for interactions nobody wrote a test for."""
import copy

import numpy as np
import pytest

from oracle import aoadmm as OA
from oracle.tensor_ops import full_ktensor
from helpers import options, par2_slabs
from test_gpu_solver import compare_par2

pytestmark = pytest.mark.gpu

CATALOGUE = [('non-negativity',), ('box', 0.0, 0.8), ('simplex column-wise', 1.0), ('simplex row-wise', 1.0),
             ('non-decreasing',), ('non-increasing',), ('unimodality', True), ('unimodality', False), ('l1-ball', 1.5),
             ('l2-ball', 1.0), ('non-negative l2-ball', 1.0), ('non-negative l2-sphere', 1.0), ('orthonormal',),
             ('l1 regularization', 0.01), ('l0 regularization', 0.01), ('l2 regularization', 0.01), ('ridge', 0.01),
             ('GL smoothness', 0.05), ('TV regularization', 0.01)]
# constraints that make sense on a PARAFAC2 C mode (K x R, kept away from zero rows) and on B_k slabs
C_MODE = [('non-negativity',), ('box', 0.0, 2.0), ('l1 regularization', 0.001), ('ridge', 0.001)]
B_MODE = [None, None, ('non-negativity',), ('l1 regularization', 0.005), ('ridge', 0.01), ('unimodality', False)]


def draw_model(seed):
    rng = np.random.default_rng(1000 + seed)
    R = int(rng.integers(2, 5))
    n_cp = int(rng.integers(1, 3))
    with_par2 = bool(rng.integers(0, 2)) and seed % 3 != 0
    size, modes, model, objs, weights, cons = [], [], [], [], [], []
    first_modes = []
    shared_rows = int(rng.integers(8, 20))
    couple = bool(rng.integers(0, 2)) and (n_cp + with_par2) >= 2
    for b in range(n_cp):
        order = int(rng.integers(2, 5)) if b == 0 else int(rng.integers(2, 4))
        dims = [int(rng.integers(5, 16)) for _ in range(order)]
        if couple:
            dims[0] = shared_rows
        fac = [rng.random((d, R)) + 0.05 for d in dims]
        X = full_ktensor(fac) if order > 2 else fac[0] @ fac[1].T
        X = X + 0.05 * np.linalg.norm(X) / np.sqrt(X.size) * rng.standard_normal(X.shape)
        X /= np.linalg.norm(X)
        first_modes.append(len(size))
        modes.append([len(size) + i + 1 for i in range(order)])
        size += dims
        model.append('CP')
        objs.append(X)
        weights.append(float(rng.choice([0.5, 1.0, 2.0])))
        for d in dims:
            c = CATALOGUE[int(rng.integers(0, len(CATALOGUE)))] if rng.random() < 0.7 else None
            if c is not None and c[0] == 'orthonormal' and d < R:
                c = None
            cons.append(c)
    if with_par2:
        K = int(rng.integers(4, 9))
        I = shared_rows if couple else int(rng.integers(6, 14))
        Jk = [int(rng.integers(R + 2, 14)) for _ in range(K)]
        X, _ = par2_slabs(I, Jk, R, rng, 0.1)
        first_modes.append(len(size))
        modes.append([len(size) + 1, len(size) + 2, len(size) + 3])
        size += [I, Jk, K]
        model.append('PAR2')
        objs.append(X)
        weights.append(float(rng.choice([0.5, 1.0])))
        cons += [CATALOGUE[int(rng.integers(0, 3))] if rng.random() < 0.6 else None,
                 B_MODE[int(rng.integers(0, len(B_MODE)))],
                 C_MODE[int(rng.integers(0, len(C_MODE)))] if rng.random() < 0.7 else None]
    nm = len(size)
    lin = [0] * nm
    ctype, H = [], [None] * nm
    if couple:
        for fm in first_modes:
            lin[fm] = 1
        if rng.random() < 0.5:
            ctype = [0]
        else:                                   # type 4: C_m = Delta * H_m, Delta has R + 1 columns, each mode keeps R of them
            ctype = [4]
            for fm in first_modes:
                sel = np.sort(rng.choice(R + 1, R, replace=False))
                Hm = np.zeros((R + 1, R))
                Hm[sel, np.arange(R)] = 1.0
                H[fm] = Hm
        same = cons[first_modes[0]]             # "put the same for coupled modes" (example scripts)
        if same is not None and same[0] in ('orthonormal', 'simplex column-wise', 'non-negative l2-sphere', 'l2-ball'):
            same = ('non-negativity',)
        for fm in first_modes:
            cons[fm] = same
    Z = dict(loss_function=['Frobenius'] * len(model), model=model, modes=modes, size=size,
             coupling=dict(lin_coupled_modes=lin, coupling_type=ctype, coupl_trafo_matrices=H),
             constrained_modes=[0 if c is None else 1 for c in cons], constraints=cons, weights=weights, object=objs)
    if rng.random() < 0.4:
        Z['ridge'] = [float(rng.choice([0.0, 1e-3, 1e-2])) for _ in range(nm)]
    distr = []
    for i in range(nm):
        distr.append((lambda a, b: rng.random((a, b)) + 0.05))
    if with_par2:
        distr[modes[-1][1] - 1] = lambda a, b: rng.standard_normal((a, b))
    io = dict(lambdas_init=[[1] * R] * len(model), nvecs=0, distr=distr, normalize=1)
    opt = options(MaxOuterIters=int(rng.integers(4, 9)), MaxInnerIters=int(rng.integers(3, 7)))
    if rng.random() < 0.35:
        opt['bsum'] = 1
        opt['bsum_weight'] = float(rng.choice([1e-3, 1e-2]))
    if seed >= 1000:                            # second family: ~20 % of the entries missing (Z.miss, EM imputation)
        miss = []
        for p_, obj in enumerate(objs):
            if model[p_] == 'CP':
                mk = rng.random(obj.shape) > 0.2
                objs[p_] = np.where(mk, obj, 0.0)
                miss.append(mk)
            elif rng.random() < 0.5:
                mks = [rng.random(x.shape) > 0.2 for x in obj]
                objs[p_] = [np.where(m, x, 0.0) for m, x in zip(mks, obj)]
                miss.append(mks)
            else:
                miss.append(None)
        Z['miss'] = miss
    return Z, io, opt


def draw_transformed(seed):
    """Third family (seeds 2000+): two CP blocks whose first modes are coupled with a transformation of type 1, 2, 3 or 5
    (random sizes, random row-sampling / column-selecting matrices); fourth family (seeds 3000+): a CP block coupled to
    the C mode of a PARAFAC2 block with type 0 or 1 (example_script14 family)."""
    from helpers import par2_C_coupled_model, transformed_coupling_model
    rng = np.random.default_rng(seed)
    if seed >= 3000 and seed % 2 == 1:
        from helpers import par2_C_transformed_model
        ctype = int(rng.choice([2, 3, 4, 5]))
        K = int(rng.integers(3, 9)) * 2
        Z, io = par2_C_transformed_model(rng, ctype, K=K, I2=int(rng.integers(6, 14)), Jk=int(rng.integers(6, 15)))
        if rng.random() < 0.4:
            Z['constrained_modes'][5] = 0
            Z['constraints'][5] = None
        Delta = [np.zeros((K, 4))] if ctype == 4 else ([np.zeros((K // 2, 4))] if ctype == 5 else None)
        opt = options(MaxOuterIters=int(rng.integers(4, 9)), MaxInnerIters=int(rng.integers(3, 7)))
        return Z, io, opt, Delta
    if seed >= 3000:
        ctype = int(rng.integers(0, 2))
        K = int(rng.integers(3, 9)) * 2
        Z, io = par2_C_coupled_model(rng, ctype, noise=0.05, K=K, I2=int(rng.integers(6, 14)), Jk=int(rng.integers(6, 15)))
        if rng.random() < 0.4:
            Z['constrained_modes'][5] = 0
            Z['constraints'][5] = None
        if rng.random() < 0.4:
            Z['ridge'] = [float(rng.choice([0.0, 1e-3])) for _ in range(6)]
        Delta = None
    else:
        ctype = int(rng.choice([1, 2, 3, 5]))
        Z, io = transformed_coupling_model(rng, ctype, noise=0.05)
        if rng.random() < 0.5:                      # other constraints on the uncoupled modes
            for m in (1, 2, 4, 5):
                c = CATALOGUE[int(rng.integers(0, len(CATALOGUE)))] if rng.random() < 0.6 else None
                Z['constraints'][m] = c
                Z['constrained_modes'][m] = 0 if c is None else 1
        Delta = [np.zeros((25, 4))] if ctype == 5 else None
    opt = options(MaxOuterIters=int(rng.integers(4, 9)), MaxInnerIters=int(rng.integers(3, 7)))
    if rng.random() < 0.3:
        opt['bsum'] = 1
        opt['bsum_weight'] = 1e-3
    return Z, io, opt, Delta


@pytest.mark.parametrize('seed', list(range(2000, 2030)) + list(range(3000, 3030)))
def test_random_coupled_model(pkg, eng, seed):
    Z, io, opt, Delta = draw_transformed(seed)
    G = OA.init_coupled_AOADMM_CMTF({**Z, 'prox_operators': None}, io, Delta=Delta, rng=np.random.default_rng(seed))
    try:
        _, Fo, _, oo = OA.cmtf_AOADMM(Z, alg_options=opt, init=copy.deepcopy(G))
    except np.linalg.LinAlgError:
        pytest.skip('the drawn model hits a singular system in the reference algorithm itself')
    _, Fg, _, og = pkg.cmtf_AOADMM(Z, alg_options=opt, init=copy.deepcopy(G), engine=eng)
    for key in ('DeltaB', 'P', 'mu_DeltaB'):
        Fo.setdefault(key, {}); Fg.setdefault(key, {})
    compare_par2(Fo, oo, Fg, og, tol=1e-7)


@pytest.mark.parametrize('seed', list(range(120)) + list(range(1000, 1040)))
def test_random_model(pkg, eng, seed):
    Z, io, opt = draw_model(seed)
    Delta = None
    if Z['coupling']['coupling_type'] == [4]:
        fm = Z['coupling']['lin_coupled_modes'].index(1)
        R = Z['coupling']['coupl_trafo_matrices'][fm].shape[1]
        Delta = [np.random.default_rng(seed).random((Z['size'][fm], R + 1))]
    G = OA.init_coupled_AOADMM_CMTF({**Z, 'prox_operators': None}, io, Delta=Delta, rng=np.random.default_rng(seed))
    try:
        _, Fo, _, oo = OA.cmtf_AOADMM(Z, alg_options=opt, init=copy.deepcopy(G))
    except np.linalg.LinAlgError:
        pytest.skip('the drawn model hits a singular system in the reference algorithm itself')
    _, Fg, _, og = pkg.cmtf_AOADMM(Z, alg_options=opt, init=copy.deepcopy(G), engine=eng)
    for key in ('DeltaB', 'P', 'mu_DeltaB'):
        Fo.setdefault(key, {}); Fg.setdefault(key, {})
    # f_constraints / f_couplings are means over the per-mode gaps that are NOT EXACTLY ZERO (cmtf_fun_AOADMM.m:1328,
    # :1347).  A gap that is zero in exact arithmetic (e.g. a factor that already satisfies its constraint) comes out
    # as 0 or as 1e-17 depending on summation order, so the divisor -- and nothing else -- may differ between two
    # correct implementations: accept values that agree up to a ratio of two such counts.
    nmodes = len(Z['size'])
    for k in ('func_constr_conv', 'func_coupl_conv'):
        a, b = np.asarray(og[k], dtype=float), np.asarray(oo[k], dtype=float)
        for i in range(len(a)):
            if np.isclose(a[i], b[i], rtol=1e-7, atol=1e-10):
                continue
            ratios = [n1 / n2 for n1 in range(1, nmodes + 1) for n2 in range(1, nmodes + 1)]
            assert any(np.isclose(a[i], b[i] * q, rtol=1e-6, atol=1e-12) for q in ratios), (k, i, a[i], b[i])
            og[k][i] = oo[k][i]
    compare_par2(Fo, oo, Fg, og, tol=1e-7)


@pytest.fixture(scope='module')
def eng2(pkg):
    """Two engines behind one context (aoadmm_create_multi, device 0 listed twice): every model is row-sharded."""
    with pkg.Engine([0, 0]) as e:
        yield e


@pytest.mark.parametrize('seed', list(range(0, 40)) + list(range(1000, 1020)) + list(range(2000, 2010)) + list(range(3000, 3010)))
def test_random_model_sharded_over_two_engines(pkg, eng2, seed):
    """The same draws through the N = 2 data path: CP blocks row-sharded, PARAFAC2 blocks slab-sharded where allowed."""
    Delta = None
    if seed >= 2000:
        Z, io, opt, Delta = draw_transformed(seed)
    else:
        Z, io, opt = draw_model(seed)
        if Z['coupling']['coupling_type'] == [4]:
            fm = Z['coupling']['lin_coupled_modes'].index(1)
            R = Z['coupling']['coupl_trafo_matrices'][fm].shape[1]
            Delta = [np.random.default_rng(seed).random((Z['size'][fm], R + 1))]
    G = OA.init_coupled_AOADMM_CMTF({**Z, 'prox_operators': None}, io, Delta=Delta, rng=np.random.default_rng(seed))
    try:
        _, Fo, _, oo = OA.cmtf_AOADMM(Z, alg_options=opt, init=copy.deepcopy(G))
    except np.linalg.LinAlgError:
        pytest.skip('the drawn model hits a singular system in the reference algorithm itself')
    opt = dict(opt, hip=dict(par2_slab_sharding=1))
    _, Fg, _, og = pkg.cmtf_AOADMM(Z, alg_options=opt, init=copy.deepcopy(G), engine=eng2)
    for key in ('DeltaB', 'P', 'mu_DeltaB'):
        Fo.setdefault(key, {}); Fg.setdefault(key, {})
    nmodes = len(Z['size'])
    for k in ('func_constr_conv', 'func_coupl_conv'):          # the non-zero-count quirk, see test_random_model
        a, b = np.asarray(og[k], dtype=float), np.asarray(oo[k], dtype=float)
        for i in range(len(a)):
            if not np.isclose(a[i], b[i], rtol=1e-7, atol=1e-10):
                ratios = [n1 / n2 for n1 in range(1, nmodes + 1) for n2 in range(1, nmodes + 1)]
                assert any(np.isclose(a[i], b[i] * q, rtol=1e-6, atol=1e-12) for q in ratios), (k, i, a[i], b[i])
                og[k][i] = oo[k][i]
    compare_par2(Fo, oo, Fg, og, tol=1e-7)
