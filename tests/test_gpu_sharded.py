"""The N > 1 data path with world > 1 on ONE GPU: every rank is a thread of this process with its own engine and
stream, joined by the library's process-local group (aoadmm_comm_init_local: host-staged, rank-ordered sums in
place of ncclAllReduce -- RCCL refuses two ranks on one device).  Everything else is the code the RCCL ranks run:
mode-1 row blocks (csrc/solver.hip tensor_upload), zero-filled own-rows MTTKRP buffers, all-reduced partial
objective sums, replicated ADMM.  Bars: all ranks bit-identical; factors within 1e-8 of the oracle (fp64)."""
import copy
import itertools
import threading

import numpy as np
import pytest

from oracle import aoadmm as OA
from helpers import cp_model, options, rel_fro, script1_model, script3_model
from test_gpu_solver import compare, compare_par2, _with_mask, _compare_em

pytestmark = pytest.mark.gpu
_keys = itertools.count(1000)


def run_sharded(pkg, Z, io, opt, world, seed=7, precision='f64'):
    rng = np.random.default_rng(seed)
    G = OA.init_coupled_AOADMM_CMTF({**Z, 'prox_operators': None}, io, rng=rng)
    _, Fo, _, oo = OA.cmtf_AOADMM(Z, alg_options=opt, init=copy.deepcopy(G))
    key = next(_keys)
    res, err = [None] * world, [None] * world

    def rank_main(r):
        try:
            with pkg.Engine(0) as e:
                e.comm_init_local(key, r, world)
                _, Fg, _, og = pkg.cmtf_AOADMM(Z, alg_options=opt, init=copy.deepcopy(G), engine=e, precision=precision)
                res[r] = (Fg, og)
        except BaseException as ex:   # noqa: BLE001 -- reported by the main thread
            err[r] = ex

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for r, ex in enumerate(err):
        assert ex is None, 'rank %d: %r' % (r, ex)
    F0, o0 = res[0]
    for r in range(1, world):                       # replicated state must be the same bits on every rank
        Fr, orr = res[r]
        for key_ in ('fac', 'constraint_fac', 'constraint_dual_fac'):
            for a, b in zip(F0[key_], Fr[key_]):
                if a is None:
                    continue
                if isinstance(a, (list, tuple)):
                    assert all(np.array_equal(x, y) for x, y in zip(a, b)), (key_, r)
                else:
                    assert np.array_equal(a, b), (key_, r)
        assert np.array_equal(o0['func_val_conv'], orr['func_val_conv'])
    return Fo, oo, F0, o0


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_cp_tv_nonneg(pkg, world):
    """config 5's model at oracle size; 37 rows over 2 or 3 ranks gives ragged blocks (19+18, 13+13+11)."""
    rng = np.random.default_rng(21)
    Z, io, _ = cp_model((37, 14, 12), 3, rng, [('TV regularization', 0.01), ('non-negativity',), ('non-negativity',)])
    compare(*run_sharded(pkg, Z, io, options(MaxOuterIters=8), world))


def test_sharded_fp32_tensor(pkg):
    rng = np.random.default_rng(22)
    Z, io, _ = cp_model((70, 33, 20), 4, rng, [('TV regularization', 0.01), ('non-negativity',), ('non-negativity',)])
    Fo, oo, Fg, og = run_sharded(pkg, Z, io, options(MaxOuterIters=5), 2, precision='f32')
    for a, b in zip(Fo['fac'], Fg['fac']):
        assert rel_fro(b, a) < 1e-4


def test_sharded_script3_partial_coupling(pkg):
    """config 3: matrix + CP tensor, coupling type 4; both blocks row-sharded along their first mode."""
    rng = np.random.default_rng(4)
    Z, io = script3_model(rng)
    compare(*run_sharded(pkg, Z, io, options(MaxOuterIters=10), 2))


def test_sharded_cp_with_replicated_parafac2(pkg):
    """config 1: the CP block is row-sharded, the PARAFAC2 block coupled to it is repeated on every rank."""
    rng = np.random.default_rng(12)
    Z, io = script1_model(rng, dims=(20, 30, 40))
    compare_par2(*run_sharded(pkg, Z, io, options(MaxOuterIters=6), 2))


def test_sharded_em_missing(pkg):
    rng = np.random.default_rng(31)
    Z, io, _ = cp_model((37, 22, 19), 3, rng, [('non-negativity',), ('non-negativity',), ('TV regularization', 1e-3)])
    Z = _with_mask(Z, rng)
    Fo, oo, Fg, og = run_sharded(pkg, Z, io, options(MaxOuterIters=8), 2)
    compare(Fo, oo, Fg, og)
    _compare_em(oo, og)


def test_sharded_four_way(pkg):
    rng = np.random.default_rng(91)
    Z, io, _ = cp_model((12, 9, 8, 7), 3, rng, [('non-negativity',), None, ('l2-ball', 1.0), ('non-negativity',)])
    compare(*run_sharded(pkg, Z, io, options(MaxOuterIters=6), 2))


# ---- PARAFAC2 slabs sharded over the ranks (aoadmm_options.par2_slab_sharding = 1) ---------------------------------
def run_sharded_par2(pkg, Z, io, opt, world, seed=7):
    opt = dict(opt)
    opt['hip'] = dict(par2_slab_sharding=1)
    rng = np.random.default_rng(seed)
    G = OA.init_coupled_AOADMM_CMTF({**Z, 'prox_operators': None}, io, rng=rng)
    oopt = {k: v for k, v in opt.items() if k != 'hip'}
    _, Fo, _, oo = OA.cmtf_AOADMM(Z, alg_options=oopt, init=copy.deepcopy(G))
    key = next(_keys)
    res, err = [None] * world, [None] * world

    def rank_main(r):
        try:
            with pkg.Engine(0) as e:
                e.comm_init_local(key, r, world)
                _, Fg, _, og = pkg.cmtf_AOADMM(Z, alg_options=opt, init=copy.deepcopy(G), engine=e)
                res[r] = (Fg, og)
        except BaseException as ex:   # noqa: BLE001
            err[r] = ex

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for r, ex in enumerate(err):
        assert ex is None, 'rank %d: %r' % (r, ex)

    def same(a, b, what):
        if a is None:
            return
        if isinstance(a, (list, tuple)):
            for x, y in zip(a, b):
                same(x, y, what)
        elif isinstance(a, dict):
            for k in a:
                same(a[k], b[k], what)
        else:
            assert np.array_equal(a, b), what
    for r in range(1, world):       # after the final gather every rank returns the same struct, bit for bit
        for key_ in ('fac', 'constraint_fac', 'constraint_dual_fac', 'DeltaB', 'P', 'mu_DeltaB'):
            same(res[0][0][key_], res[r][0][key_], (key_, r))
        assert np.array_equal(res[0][1]['func_val_conv'], res[r][1]['func_val_conv'])
    return Fo, oo, res[0][0], res[0][1]


@pytest.mark.parametrize('world', [2, 3])
def test_slab_sharded_irregular_parafac2(pkg, world):
    """config 4 family, K = 13 ragged slabs over 2 (7+6) or 3 (5+5+3) ranks: per-slab kernels on the own range,
    all-reduced sums for mode A, DeltaB, the residual means and the objective, gathered C-mode row systems."""
    from helpers import script4_model
    rng = np.random.default_rng(10)
    Z, io = script4_model(rng, K=13)
    compare_par2(*run_sharded_par2(pkg, Z, io, options(MaxOuterIters=10), world))


def test_config4_as_written_256_slabs_over_4_ranks(pkg):
    """BASELINE config 4 as written: irregular PARAFAC2 (example_script4 shapes: I = 40, R = 3, J_k cycled over 61..120,
    C non-negative), K = 256 slabs sharded over 4 ranks (64 slabs each) with par2_slab_sharding = 1; the collectives go
    through the process-local group (host-staged sums: RCCL refuses four ranks on one device).  Every field of G within
    1e-8 of the oracle, all ranks bit-identical.  Sums over k: cmtf_fun_AOADMM.m:163-164 (mode A), :537-547 (DeltaB
    per inner iteration), :582-585 (residual means)."""
    from helpers import script4_model
    rng = np.random.default_rng(21)
    Z, io = script4_model(rng, K=256)
    compare_par2(*run_sharded_par2(pkg, Z, io, options(MaxOuterIters=5), 4))


def test_slab_sharded_constrained_and_regularised_Bk(pkg):
    from helpers import script4_model
    rng = np.random.default_rng(11)
    Z, io = script4_model(rng, K=6, constraints_B=('unimodality', False))
    opt = options(MaxOuterIters=8, iter_start_PAR2Bkconstraint=3, increase_factor_rhoBk=2.0)
    compare_par2(*run_sharded_par2(pkg, Z, io, opt, 2))
    Z, io = script4_model(rng, K=7, constraints_B=('l1 regularization', 0.01))
    compare_par2(*run_sharded_par2(pkg, Z, io, options(MaxOuterIters=6), 2))


def test_slab_sharded_parafac2_coupled_to_row_sharded_cp(pkg):
    """config 1: CP block row-sharded along mode 1 and PARAFAC2 block slab-sharded, first modes exactly coupled."""
    rng = np.random.default_rng(12)
    Z, io = script1_model(rng, dims=(20, 30, 40))
    compare_par2(*run_sharded_par2(pkg, Z, io, options(MaxOuterIters=6), 2))


def test_slab_sharding_with_early_inner_exit(pkg):
    """Non-zero inner tolerances: the device-side loop control must stay in step on all ranks (same all-reduced
    residual means), with the collectives still issued after the loop went inactive."""
    from helpers import script4_model
    rng = np.random.default_rng(13)
    Z, io = script4_model(rng, K=9)
    opt = options(MaxOuterIters=15, MaxInnerIters=8, innerRelPrTol_coupl=1e-2, innerRelDualTol_coupl=1e-2,
                  innerRelPrTol_constr=1e-2, innerRelDualTol_constr=1e-2)
    compare_par2(*run_sharded_par2(pkg, Z, io, opt, 2))


# ---- one process, several engines behind ONE context (aoadmm_create_multi): the shape a MATLAB session uses ---------
@pytest.mark.parametrize('ndev', [2, 3])
def test_multi_device_context_single_caller(pkg, ndev):
    """`Engine([0, 0])`: one context, one caller thread, the library fans every call out to its per-device worker
    threads (device 0 listed several times selects the host-staged transport, RCCL would refuse it).  Same factors as
    the oracle, same objective history, for a row-sharded CP block with TV + non-negativity."""
    rng = np.random.default_rng(21)
    Z, io, _ = cp_model((37, 14, 12), 3, rng, [('TV regularization', 0.01), ('non-negativity',), ('non-negativity',)])
    G = OA.init_coupled_AOADMM_CMTF({**Z, 'prox_operators': None}, io, rng=np.random.default_rng(7))
    opt = options(MaxOuterIters=8)
    _, Fo, _, oo = OA.cmtf_AOADMM(Z, alg_options=opt, init=copy.deepcopy(G))
    with pkg.Engine([0] * ndev) as e:
        assert e.comm_rank() == (0, ndev)
        _, Fg, _, og = pkg.cmtf_AOADMM(Z, alg_options=opt, init=copy.deepcopy(G), engine=e)
        compare(Fo, oo, Fg, og)
        # the same context again with another model (coupled CP + PARAFAC2, slabs sharded over the engines)
        Z2, io2 = script1_model(np.random.default_rng(12), dims=(20, 30, 40))
        G2 = OA.init_coupled_AOADMM_CMTF({**Z2, 'prox_operators': None}, io2, rng=np.random.default_rng(7))
        opt2 = options(MaxOuterIters=5)
        _, Fo2, _, oo2 = OA.cmtf_AOADMM(Z2, alg_options=opt2, init=copy.deepcopy(G2))
        opt2h = dict(opt2, hip=dict(par2_slab_sharding=1))
        _, Fg2, _, og2 = pkg.cmtf_AOADMM(Z2, alg_options=opt2h, init=copy.deepcopy(G2), engine=e)
        compare_par2(Fo2, oo2, Fg2, og2)


def test_multi_device_context_reports_errors(pkg):
    """A failure inside the workers comes back as a status + message naming the rank, not as a hang or a crash."""
    capi = __import__('importlib').import_module('matlab-code_amd._capi')
    with pkg.Engine([0, 0]) as e:
        with pytest.raises(capi.AoadmmError) as ei:
            capi.check(e.lib.aoadmm_model_begin(e.h, 0, 0, 0))
        assert 'rank 0' in str(ei.value)
        with pytest.raises(capi.AoadmmError):
            e.comm_init_local(1, 0, 2)                 # the context owns its communicator


def test_multi_device_progress_runs_on_the_callers_thread(pkg):
    """options.Display = 'iter' on a multi-device context: rank 0's worker thread produces the rows, the CALLING thread
    delivers them (include/aoadmm_hip.h: a MEX callback may only touch MATLAB from the interpreter's thread)."""
    capi = __import__('importlib').import_module('matlab-code_amd._capi')
    rng = np.random.default_rng(5)
    Z, io, _ = cp_model((21, 9, 8), 2, rng, [('non-negativity',)] * 3)
    G = OA.init_coupled_AOADMM_CMTF({**Z, 'prox_operators': None}, io, rng=np.random.default_rng(7))
    seen = []
    cb = capi.PROGRESS_FN(lambda user, it, f, frm: seen.append((threading.get_ident(), it, f[0])))
    with pkg.Engine([0, 0]) as e:
        capi.check(e.lib.aoadmm_set_progress(e.h, cb, None, 2))
        _, _, _, og = pkg.cmtf_AOADMM(Z, alg_options=options(MaxOuterIters=6), init=copy.deepcopy(G), engine=e)
        capi.check(e.lib.aoadmm_set_progress(e.h, capi.PROGRESS_FN(0), None, 0))
    assert [s[1] for s in seen] == [0, 2, 4, 6]
    assert all(s[0] == threading.get_ident() for s in seen), 'progress callback left the calling thread'
    assert seen[-1][2] == og['func_val_conv'][6]


def test_rank_uniform_ownership_checks(pkg):
    """Decisions that depend on the rank are the same on every rank: 9 rows over 8 engines is refused by ALL of them
    (ranks 5-7 would own nothing) instead of three ranks throwing while five wait in the next collective; 5 PARAFAC2
    slabs over 4 engines with sharding forced fall back to the repeated block everywhere."""
    from helpers import script4_model
    capi = __import__('importlib').import_module('matlab-code_amd._capi')
    rng = np.random.default_rng(3)
    Z, io, _ = cp_model((9, 6, 5), 2, rng, [('non-negativity',)] * 3)
    G = OA.init_coupled_AOADMM_CMTF({**Z, 'prox_operators': None}, io, rng=np.random.default_rng(7))
    with pkg.Engine([0] * 8) as e:
        with pytest.raises(capi.AoadmmError) as ei:
            pkg.cmtf_AOADMM(Z, alg_options=options(MaxOuterIters=2), init=copy.deepcopy(G), engine=e)
        assert 'cannot be split over 8 ranks' in str(ei.value)
    Z4, io4 = script4_model(np.random.default_rng(10), K=5)
    G4 = OA.init_coupled_AOADMM_CMTF({**Z4, 'prox_operators': None}, io4, rng=np.random.default_rng(7))
    opt = options(MaxOuterIters=4)
    _, Fo, _, oo = OA.cmtf_AOADMM(Z4, alg_options=opt, init=copy.deepcopy(G4))
    with pkg.Engine([0] * 4) as e:
        _, Fg, _, og = pkg.cmtf_AOADMM(Z4, alg_options=dict(opt, hip=dict(par2_slab_sharding=1)), init=copy.deepcopy(G4), engine=e)
    compare_par2(Fo, oo, Fg, og)


def test_multi_device_lone_failure_does_not_hang(pkg, monkeypatch):
    """One rank fails alone in front of a collective (injected): the caller gets that rank's error after the grace
    period instead of waiting for ever on the peers that sit in the collective."""
    import time
    capi = __import__('importlib').import_module('matlab-code_amd._capi')
    rng = np.random.default_rng(5)
    Z, io, _ = cp_model((21, 9, 8), 2, rng, [('non-negativity',)] * 3)
    G = OA.init_coupled_AOADMM_CMTF({**Z, 'prox_operators': None}, io, rng=np.random.default_rng(7))
    monkeypatch.setenv('AOADMM_FAULT_INJECT', 'normsq:1')
    t0 = time.time()
    with pkg.Engine([0, 0, 0]) as e:
        with pytest.raises(capi.AoadmmError) as ei:
            pkg.cmtf_AOADMM(Z, alg_options=options(MaxOuterIters=3), init=copy.deepcopy(G), engine=e)
        assert 'rank 1' in str(ei.value) and 'injected fault' in str(ei.value)
        assert time.time() - t0 < 60
        # the context is poisoned for good: the aborted ranks would skip collectives the failed rank still enters, so
        # every later call fails at once (no second grace period, no silent unsharded work)
        monkeypatch.delenv('AOADMM_FAULT_INJECT')
        t1 = time.time()
        with pytest.raises(capi.AoadmmError) as ej:
            pkg.cmtf_AOADMM(Z, alg_options=options(MaxOuterIters=1), init=copy.deepcopy(G), engine=e)
        assert ej.value.code == capi.ERR_RCCL and 'unusable' in str(ej.value)
        assert time.time() - t1 < 2


def test_op_level_calls_on_a_multi_device_context(pkg):
    """Host-in/host-out op-level entries run on one engine without a collective, whatever communicator it is in;
    aoadmm_tensor_upload_rows (one block for every rank) is refused."""
    import ctypes as C
    capi = __import__('importlib').import_module('matlab-code_amd._capi')
    from oracle.tensor_ops import mttkrp as mttkrp_ref
    rng = np.random.default_rng(8)
    X = rng.standard_normal((11, 7, 5))
    U = [rng.standard_normal((n, 3)) for n in X.shape]
    with pkg.Engine([0, 0]) as e:
        for n in range(3):
            assert rel_fro(e.mttkrp(X, U, n), mttkrp_ref(X, U, n)) < 1e-12
        blk = capi.as_f(X[:5].reshape(5, -1, order='F'))
        st = e.lib.aoadmm_tensor_upload_rows(e.h, 0, capi.dptr(blk), 0, 5, capi.PREC_F64)
        assert st == capi.ERR_INVALID
