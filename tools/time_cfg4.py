#!/usr/bin/env python3
"""Development timing of BASELINE config 4 (irregular PARAFAC2, K = 256 slabs, I = 40, J_k in 61..120, R = 3).

  time_cfg4.py               one engine, per outer iteration (slope between two solve lengths)
  time_cfg4.py --sharded N   N ranks (threads, one engine each on device 0) joined by the process-local group, once with
                             the slabs sharded (par2_slab_sharding = 1) and once replicated (-1): what the policy
                             `auto` has to choose between.  The local group stages every collective through the host
                             (two PCIe crossings + two thread barriers each) and the N ranks share ONE GPU, so the sharded
                             figure is an upper bound for the sharded form on N GPUs, and the replicated figure is N
                             copies of the work on one GPU (divide the kernel time by N for N GPUs)."""
import copy, importlib, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
pkg = importlib.import_module('matlab-code_amd')
from oracle import aoadmm as OA
from helpers import script4_model, options
rng = np.random.default_rng(4)
Z, io = script4_model(rng, K=256)
G = OA.init_coupled_AOADMM_CMTF({**Z, 'prox_operators': None}, io, rng=np.random.default_rng(7))


def single():
    eng = pkg.Engine(0)
    times = {}
    for n in (5, 50, 250):
        t = time.perf_counter()
        _, F, _, out = pkg.cmtf_AOADMM(Z, alg_options=options(MaxOuterIters=n), init=copy.deepcopy(G), engine=eng)
        times[n] = time.perf_counter() - t
        print('cfg4 K=256: %d outer iterations in %.3f s (incl. upload and download of the %d slabs)' % (n, times[n], 256), flush=True)
    print('cfg4 K=256: %.3f ms per outer iteration (slope between 50 and 250 iterations)'
          % ((times[250] - times[50]) / 200 * 1e3), flush=True)
    t = time.perf_counter()
    OA.cmtf_AOADMM(Z, alg_options=options(MaxOuterIters=5), init=copy.deepcopy(G))
    print('oracle (numpy): %.1f ms/iteration' % ((time.perf_counter() - t) / 5 * 1e3))
    eng.close()


def ranks(world, policy, key):
    """ms per outer iteration (slope) of `world` thread-ranks with par2_slab_sharding = policy"""
    out = {}
    for n in (20, 120):
        bar = threading.Barrier(world)
        dt = [0.0] * world
        err = [None] * world

        def main(r, n=n):
            try:
                with pkg.Engine(0) as e:
                    e.comm_init_local(key + n, r, world)
                    opt = dict(options(MaxOuterIters=n), hip=dict(par2_slab_sharding=policy))
                    bar.wait()
                    t = time.perf_counter()
                    pkg.cmtf_AOADMM(Z, alg_options=opt, init=copy.deepcopy(G), engine=e)
                    dt[r] = time.perf_counter() - t
            except BaseException as ex:  # noqa: BLE001
                err[r] = ex
                try:
                    bar.abort()
                except Exception:
                    pass
        th = [threading.Thread(target=main, args=(r,)) for r in range(world)]
        [t.start() for t in th]
        [t.join() for t in th]
        for ex in err:
            if ex is not None:
                raise ex
        out[n] = max(dt)
    return (out[120] - out[20]) / 100 * 1e3


def share(rank, world):
    """one rank's share of the slab-sharded job on a one-rank RCCL communicator (aoadmm_comm_init_rank_share): the
    per-slab kernels on K/world slabs and every collective issued, no peers -- a per-rank step without xGMI latency"""
    out = {}
    with pkg.Engine(0) as e:
        e.comm_init_rank_share(e.comm_unique_id(), rank, world)
        for n in (20, 220):
            opt = dict(options(MaxOuterIters=n), hip=dict(par2_slab_sharding=1))
            t = time.perf_counter()
            pkg.cmtf_AOADMM(Z, alg_options=opt, init=copy.deepcopy(G), engine=e)
            out[n] = time.perf_counter() - t
    return (out[220] - out[20]) / 200 * 1e3


if '--as-rank' in sys.argv:
    r = int(sys.argv[sys.argv.index('--as-rank') + 1]); w = int(sys.argv[sys.argv.index('--of') + 1])
    print('cfg4 K=256, rank %d of %d, slabs sharded, one-rank RCCL communicator: %.3f ms per outer iteration '
          '(collectives issued without peers)' % (r, w, share(r, w)))
elif '--sharded' in sys.argv:
    w = int(sys.argv[sys.argv.index('--sharded') + 1])
    sh = ranks(w, 1, 5000)
    rp = ranks(w, -1, 6000)
    print('cfg4 K=256 over %d thread-ranks on ONE GPU, process-local group (host-staged collectives):' % w)
    print('  slabs sharded   (par2_slab_sharding = 1):  %.3f ms per outer iteration' % sh)
    print('  block repeated  (par2_slab_sharding = -1): %.3f ms per outer iteration (%d copies of the work on one GPU)' % (rp, w))
else:
    single()
