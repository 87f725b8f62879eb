"""CPU, world_size 2 over gloo: the N > 1 path's algorithm.

The multi-GPU engine row-shards the tensor's first mode, keeps factor matrices replicated and
all-reduces each MTTKRP output (csrc/solver.hip block_mttkrp + Engine::allreduce); the copy for the
pass that contracts mode 1 is sharded along mode 3 instead (CpBlock::xp_ksharded), so MTTKRPs of
modes 2 and 3 come from either partition depending on which pass serves them.  Here the same
partitions (`row_block`, product code) and the same collective pattern are run with the oracle as
the per-rank compute: the sharded AO-ADMM must give the same factors as the unsharded one, and every
rank must hold identical factors (replicated ADMM stays in lock-step).
"""
import copy
import importlib
import os
import sys
import tempfile

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, initfile, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import torch
    from oracle import aoadmm as OA
    from oracle import tensor_ops as OT
    from helpers import cp_model, options
    pkg = importlib.import_module('matlab-code_amd')
    dist.init_process_group('gloo', init_method='file://' + initfile, rank=rank, world_size=world)
    rng = np.random.default_rng(11)                      # same model on every rank
    Z, io, _ = cp_model((37, 14, 12), 3, rng, [('TV regularization', 0.01), ('non-negativity',), ('non-negativity',)])
    G = OA.init_coupled_AOADMM_CMTF(Z, io, rng=np.random.default_rng(5))
    opt = options(MaxOuterIters=6)
    X = Z['object'][0]
    r0, nloc = pkg.row_block(X.shape[0], world, rank)
    Xloc = X[r0:r0 + nloc]
    k0, kloc = pkg.row_block(X.shape[2], world, rank)    # the mode-3 slab of the mode-1 pass's copy
    Xk = X[:, :, k0:k0 + kloc]

    calls = {'n': 0}

    def sharded_mttkrp(Xfull, U, n):
        assert Xfull is X
        calls['n'] += 1
        if n != 0 and calls['n'] % 2 == 1:               # served by the pass on the mode-3 slab (all of mode 1)
            Uk = [U[0], U[1], U[2][k0:k0 + kloc]]
            part = OT.mttkrp(Xk, Uk, n)
            if n == 2:                                   # own rows of mode 3 into a zero buffer
                full = np.zeros((X.shape[2], part.shape[1]))
                full[k0:k0 + kloc] = part
            else:                                        # mode 2: partial sums over this rank's k
                full = part
        else:
            Uloc = [U[0][r0:r0 + nloc]] + list(U[1:])
            part = OT.mttkrp(Xloc, Uloc, n)
            if n == 0:                                   # own rows into a zero buffer: all-reduce == all-gather
                full = np.zeros((X.shape[0], part.shape[1]))
                full[r0:r0 + nloc] = part
            else:
                full = part
        t = torch.from_numpy(np.ascontiguousarray(full))
        dist.all_reduce(t)
        return t.numpy()

    ref_mttkrp = OA.mttkrp
    OA.mttkrp = sharded_mttkrp
    try:
        _, Fs, _, outs = OA.cmtf_AOADMM(Z, alg_options=opt, init=copy.deepcopy(G))
    finally:
        OA.mttkrp = ref_mttkrp
    _, Fr, _, outr = OA.cmtf_AOADMM(Z, alg_options=opt, init=copy.deepcopy(G))
    err = max(np.linalg.norm(a - b) / np.linalg.norm(b) for a, b in zip(Fs['fac'], Fr['fac']))
    # replicas identical: compare a checksum of the factors across ranks
    chk = torch.tensor([float(sum(np.sum(f) for f in Fs['fac']))], dtype=torch.float64)
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    np.save(os.path.join(outdir, 'r%d.npy' % rank),
            np.array([err, float(hi - lo), calls['n'], abs(outs['f_tensors'] - outr['f_tensors'])]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_aoadmm_matches_unsharded_world2():
    world = 2
    with tempfile.TemporaryDirectory() as d:
        initfile = os.path.join(d, 'init')
        mp.spawn(_worker, args=(world, initfile, d), nprocs=world, join=True)
        for r in range(world):
            err, spread, ncalls, df = np.load(os.path.join(d, 'r%d.npy' % r))
            assert err < 1e-12, err            # same factors as the single-process run (summation order only)
            assert spread == 0.0               # bit-identical replicas
            assert ncalls == 6 * 3 + 1         # one collective per mode update (+ initial objective)
            assert df < 1e-12


def test_row_block_partition_is_exact():
    pkg = importlib.import_module('matlab-code_amd')
    for rows in (1, 7, 250, 2000, 2001):
        for world in (1, 2, 3, 4, 8):
            blocks = [pkg.row_block(rows, world, r) for r in range(world)]
            assert sum(n for _, n in blocks) == rows
            pos = 0
            for r0, n in blocks:
                if n:
                    assert r0 == pos
                pos += n
