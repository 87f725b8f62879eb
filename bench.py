#!/usr/bin/env python3
"""Headline benchmark: AO-ADMM outer iterations / second and mode-1 MTTKRP GFLOP/s.

Workload (BASELINE.json configs[4], the config the metric is quoted on; it fits one
GPU): one 3-way CP block 2000 x 2000 x 2000, rank 20, mode 1 `{'TV regularization',
0.001}`, modes 2-3 `{'non-negativity'}`, tensor stored fp32 and contracted with
v_mfma_f32_16x16x4_f32 (+ packed-fp32 VALU for columns 17-20), everything else fp64, MaxInnerIters = 5, all tolerances 0
(fixed work).  Data: synthetic, generated in HBM; init: seeded rand,
column-normalised (init_coupled_AOADMM_CMTF.m:88-93,119-124).

One "step" = one outer AO-ADMM iteration (cmtf_fun_AOADMM.m:87-476) through the
solver-level C ABI entry `aoadmm_solve`.  N > 1 (launched by torchrun, one rank per
GPU): the tensor's first mode is row-sharded, factor matrices are replicated and
only MTTKRP outputs are all-reduced over RCCL (strong scaling: total work fixed).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak (spec); ~6.3 TB/s measured achievable
F32_MFMA_PEAK_TF = 157.3   # MI355X_MICROARCH.md: f32-input MFMA peak


def build_Z(I, J, K, R, seed, noise):
    return dict(
        loss_function=['Frobenius'], model=['CP'], modes=[[1, 2, 3]], size=[I, J, K],
        coupling=dict(lin_coupled_modes=[0, 0, 0], coupling_type=[], coupl_trafo_matrices=[None] * 3),
        constrained_modes=[1, 1, 1],
        constraints=[('TV regularization', 0.001), ('non-negativity',), ('non-negativity',)],
        weights=[1.0], object=[dict(synthetic=True, rank=R, seed=seed, noise=noise)])


def cpu_baseline(I, J, K, R, budget_s=25.0, rows=None):
    """CPU restatement (NOT MATLAB) timed beside the GPU number: the C/OpenMP port of the same outer iteration
    (oracle/c/aoadmm_cpu.c: three MTTKRPs over the fp32 tensor, Gram/Hadamard/Cholesky, ADMM inner loops with
    non-negativity and Condat's TV) on the box's host cores, on the same synthetic workload generated on the host.
    A 64-row probe sizes the sample: the whole tensor when one generation + three iterations fit `budget_s`, else the
    tallest mode-1 slab that does (then `extrapolated` is true and the slab size is reported)."""
    from oracle import c_port
    c_port.set_threads(c_port.usable_cpus())              # affinity mask capped by the cgroup CPU quota
    threads = c_port.threads()
    cons = [('TV regularization', 0.001), ('non-negativity',), ('non-negativity',)]

    def run(nrows, outer):
        rng = np.random.default_rng(1)
        t0 = time.perf_counter()
        X, _ = c_port.synth(nrows, J, K, R, 0.05, 0)
        t_gen = time.perf_counter() - t0
        fac = [rng.random((n, R)) for n in (nrows, J, K)]
        fac = [f / np.linalg.norm(f, axis=0) for f in fac]
        Zc = [rng.random((n, R)) for n in (nrows, J, K)]
        mu = [rng.random((n, R)) for n in (nrows, J, K)]
        ts = []
        for k in outer:
            t0 = time.perf_counter()
            c_port.solve_cp3(X, cons, fac, Zc, mu, k, 5, normsq=1.0)
            ts.append(time.perf_counter() - t0)
        return t_gen, ts

    if rows is None:
        g, ts = run(min(64, I), (1,))
        per_row = (g + 2.5 * ts[0]) / min(64, I)          # generation + (1 + 3) iterations incl. the initial objective pass
        rows = int(min(I, max(64, budget_s / max(per_row, 1e-9))))
        if rows >= 0.9 * I:
            rows = I
    t_gen, ts = run(rows, (1, 3))
    per_iter = (ts[1] - ts[0]) / 2.0                      # slope: the initial objective pass drops out
    scale = I / rows
    return {
        'value': 1.0 / (per_iter * scale), 'unit': 'iters/s', 'cores': int(threads), 'kind': 'port',
        'extrapolated': bool(rows != I), 'slab_rows': int(rows), 'scale_to_full': float(scale),
        'sample': 'C/OpenMP CPU restatement (oracle/c/aoadmm_cpu.c, not MATLAB), %d threads: %dx%dx%d fp32 tensor '
                  'generated on the host in %.1f s; 1 and 3 outer iterations (MaxInnerIters 5, TV + non-negativity) took '
                  '%.2f s and %.2f s -> %.3f s per iteration%s'
                  % (threads, rows, J, K, t_gen, ts[0], ts[1], per_iter,
                     '' if rows == I else '; scaled x%.2f to %d rows' % (scale, I)),
    }


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--size', type=int, default=2000)
    ap.add_argument('--rank', type=int, default=20)
    ap.add_argument('--prec', default='f32', choices=['f32', 'f64'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-rows', type=int, default=0, help='mode-1 rows of the CPU baseline sample (0: sized by a probe)')
    ap.add_argument('--as-rank', type=int, default=-1,
                    help="with --of N: time rank R's share of an N-GPU job on ONE GPU (its row block of mode 1, its mode-3 slab "
                         'of the mode-1 pass, own-rows buffers, every collective issued on a one-rank RCCL communicator); '
                         'a measurement of the per-rank step, not a solve of the problem')
    ap.add_argument('--of', type=int, default=0)
    ap.add_argument('--no-drift', action='store_true',
                    help='skip the fp64 repeat of the same iterations (N = 1, fp32 runs only) that measures the factor drift')
    return ap.parse_args(argv)


def launch_plan(gpus, environ, n_devices, argv, port=None):
    """How this invocation runs (pure function, tested on CPU by tests/test_bench_launch.py).

    * inside a launcher (WORLD_SIZE set): this process is one rank; WORLD_SIZE must equal --gpus;
    * --gpus 1 without a launcher: this process is the only rank;
    * --gpus N > 1 without a launcher: this process becomes a PARENT that never touches the GPU and starts
      `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>`;
      its exit code is the children's.
    Fewer than N visible devices is an error in every case (never a silent 1-GPU number)."""
    if gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    ws = environ.get('WORLD_SIZE')
    if ws is not None:
        world = int(ws)
        if world != gpus:
            raise SystemExit('bench.py: WORLD_SIZE=%d but --gpus %d' % (world, gpus))
        if n_devices is not None and n_devices < int(environ.get('LOCAL_RANK', '0')) + 1:
            raise SystemExit('bench.py: rank with LOCAL_RANK=%s but only %d GPU(s) visible'
                             % (environ.get('LOCAL_RANK', '0'), n_devices))
        return {'mode': 'rank', 'world': world, 'rank': int(environ.get('RANK', '0')),
                'local_rank': int(environ.get('LOCAL_RANK', '0'))}
    if n_devices is not None and n_devices < gpus:
        raise SystemExit('bench.py: --gpus %d but only %d GPU(s) visible: refusing to report a smaller job under that name'
                         % (gpus, n_devices))
    if gpus == 1:
        return {'mode': 'rank', 'world': 1, 'rank': 0, 'local_rank': 0}
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    return {'mode': 'spawn', 'world': gpus, 'cmd': cmd}


def visible_devices():
    """Device count from a CHILD process (the parent of a self-launched job must not initialise the GPU)."""
    import subprocess
    code = ('import ctypes, sys; sys.path.insert(0, %r); import importlib; '
            'lib = importlib.import_module("matlab-code_amd").load_library(); n = ctypes.c_int(0); '
            'lib.aoadmm_device_count(ctypes.byref(n)); print(n.value)' % ROOT)
    try:
        out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
        return int(out.stdout.strip().splitlines()[-1])
    except Exception:
        return None


def main():
    args = parse_args()
    in_launcher = 'WORLD_SIZE' in os.environ
    # a lone `--gpus 1` run needs no probe either: Engine(0) fails loudly without a device
    n_dev = visible_devices() if (not in_launcher and args.gpus > 1) else None
    plan = launch_plan(args.gpus, os.environ, n_dev, sys.argv[1:])
    if plan['mode'] == 'spawn':
        import subprocess
        env = dict(os.environ)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        env.setdefault('OMP_NUM_THREADS', '4')
        raise SystemExit(subprocess.call(plan['cmd'], env=env))
    rank, world, local_rank = plan['rank'], plan['world'], plan['local_rank']
    # before anything touches the GPU: RCCL between processes needs dmabuf IPC on this driver (hipIpcGetMemHandle fails
    # otherwise); the launcher normally exports it already
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import torch
    dist = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # control plane only; a rank that dies early must not leave the others waiting for the default half hour
        dist.init_process_group(backend='gloo', rank=rank, world_size=world, timeout=datetime.timedelta(minutes=10))
    torch.cuda.set_device(local_rank)

    pkg = importlib.import_module('matlab-code_amd')
    capi = importlib.import_module('matlab-code_amd._capi')
    eng = pkg.Engine(local_rank)
    if world > 1:
        pkg.init_engine_comm(eng, dist)               # data plane: RCCL inside the library
    elif args.as_rank >= 0:
        if not (args.of > 1 and args.as_rank < args.of):
            raise SystemExit('bench.py: --as-rank R needs --of N with 0 <= R < N, N > 1')
        eng.comm_init_rank_share(eng.comm_unique_id(), args.as_rank, args.of)
    elif os.environ.get('AOADMM_BENCH_ONE_RANK_COMM'):
        # development switch: run the N > 1 data path (zero-filled own-rows buffer + ncclAllReduce of every MTTKRP
        # output) with a one-rank RCCL communicator -- the only form a one-GPU box can exercise at full size
        eng.comm_init_rank(eng.comm_unique_id(), 0, 1)
    comm = eng.comm_info()
    if world > 1 and comm['comm_ranks'] != world:
        raise SystemExit('bench.py: communicator has %d ranks, expected %d' % (comm['comm_ranks'], world))

    I = J = K = args.size
    R = args.rank
    Z = build_Z(I, J, K, R, seed=0, noise=0.05)
    Z['_ranks'] = [R, R, R]
    rng = np.random.default_rng(1)                     # identical on every rank: replicated factors
    io = dict(lambdas_init=[[1] * R], nvecs=0, distr=[lambda a, b: rng.random((a, b))] * 3, normalize=1)
    t_gen = time.perf_counter()
    pkg.build_model(eng, Z, args.prec)                 # generates the tensor in HBM
    G = pkg.init_coupled_AOADMM_CMTF(Z, io, rng=rng, engine=eng)
    import copy
    G0 = copy.deepcopy(G)
    pkg.upload_state(eng, Z, G)
    eng.synchronize()
    t_gen = time.perf_counter() - t_gen

    def opts(n):
        return dict(MaxOuterIters=n, MaxInnerIters=5, AbsFuncTol=0.0, OuterRelTol=0.0, innerRelPrTol_coupl=0.0,
                    innerRelPrTol_constr=0.0, innerRelDualTol_coupl=0.0, innerRelDualTol_constr=0.0, bsum=0)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        pkg.run_solver(eng, opts(args.warmup), 3)
    ms = C.c_double(); nl = C.c_int64(); by = C.c_double(); fl = C.c_double()
    ms1 = C.c_double(); nl1 = C.c_int64(); by1 = C.c_double(); fl1 = C.c_double()
    capi.check(eng.lib.aoadmm_kernel_stats(eng.h, 0, 1, C.byref(ms), C.byref(nl), C.byref(by), C.byref(fl)))
    capi.check(eng.lib.aoadmm_kernel_stats(eng.h, 1, 1, C.byref(ms1), C.byref(nl1), C.byref(by1), C.byref(fl1)))
    barrier()
    t0 = time.perf_counter()
    out = pkg.run_solver(eng, opts(args.steps), 3)
    barrier()
    dt = time.perf_counter() - t0
    capi.check(eng.lib.aoadmm_kernel_stats(eng.h, 0, 1, C.byref(ms), C.byref(nl), C.byref(by), C.byref(fl)))
    capi.check(eng.lib.aoadmm_kernel_stats(eng.h, 1, 1, C.byref(ms1), C.byref(nl1), C.byref(by1), C.byref(fl1)))
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # factors after exactly warmup + steps iterations, for the fp32-vs-fp64 comparison below (the runs that follow
    # iterate further)
    fac32 = None
    share = args.of if args.as_rank >= 0 else 1            # > 1: this process did 1/share of the job (--as-rank)
    if world == 1 and share == 1 and args.prec == 'f32' and not args.no_drift:
        fac32 = pkg.download_state(eng, Z, G0)['fac']

    # where the time outside the tensor passes goes: a few more iterations with the reductions over T timed as well
    # (HIP events around them, switched on by the first kernel_stats(2) call; outside the timed region)
    tail = None
    try:
        msr = C.c_double(); nlr = C.c_int64(); byr = C.c_double()
        capi.check(eng.lib.aoadmm_kernel_stats(eng.h, 2, 1, None, None, None, None))
        capi.check(eng.lib.aoadmm_kernel_stats(eng.h, 0, 1, None, None, None, None))
        capi.check(eng.lib.aoadmm_kernel_stats(eng.h, 1, 1, None, None, None, None))
        nb = max(args.steps, 1)           # same length as the timed run: the solve's fixed cost weighs the same
        eng.synchronize()
        tb = time.perf_counter()
        pkg.run_solver(eng, opts(nb), 3)
        eng.synchronize()
        tb = (time.perf_counter() - tb) / nb * 1e3
        msp = C.c_double(); msp1 = C.c_double()
        capi.check(eng.lib.aoadmm_kernel_stats(eng.h, 2, 1, C.byref(msr), C.byref(nlr), C.byref(byr), None))
        capi.check(eng.lib.aoadmm_kernel_stats(eng.h, 0, 1, C.byref(msp), None, None, None))
        capi.check(eng.lib.aoadmm_kernel_stats(eng.h, 1, 1, C.byref(msp1), None, None, None))
        red_ms = msr.value / nb
        pass_ms = (msp.value + msp1.value) / nb
        tail = {'iterations': nb, 'ms_per_step': tb, 'tensor_passes_ms': pass_ms, 't_reductions_ms': red_ms,
                't_reductions_per_iter': nlr.value / nb,
                't_reductions_GBps': (byr.value / (msr.value * 1e-3) / 1e9) if msr.value > 0 else None,
                'replicated_small_kernels_ms': tb - pass_ms - red_ms,
                'note': 'per rank; the reductions read the partial contraction T once each and shard with the passes; '
                        'replicated_small_kernels_ms (system builds, ADMM inner loops, Gram matrices, objective) is what '
                        'every rank repeats'}
    except Exception as e:   # a report, never a reason to lose the headline number
        tail = {'error': repr(e)}

    # bare mode-1 MTTKRP on the resident tensor (contraction + reduction), a few repetitions
    el = C.c_float()
    reps = []
    for _ in range(3):
        capi.check(eng.lib.aoadmm_resident_mttkrp(eng.h, 0, 0, None, C.byref(el)))
        reps.append(el.value)
    mttkrp_ms = float(np.median(reps))
    if dist is not None:
        tt = torch.tensor([mttkrp_ms], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        mttkrp_ms = float(tt.item())
    capi.check(eng.lib.aoadmm_kernel_stats(eng.h, 0, 1, None, None, None, None))

    # fp32-tensor mode against the fp64 parity mode on the SAME synthetic tensor (same seed, generated in HBM), same init,
    # same iterations: relative Frobenius gap of the factors.  Outside the timed region; the engines run one after the
    # other (96 GB, then 192 GB of HBM at 2000^3).
    drift = None
    parity_mode = None
    if fac32 is not None:
        try:
            eng.close()
            eng = pkg.Engine(local_rank)
            Z64 = build_Z(I, J, K, R, seed=0, noise=0.05)
            Z64['_ranks'] = [R, R, R]
            pkg.build_model(eng, Z64, 'f64')
            pkg.upload_state(eng, Z64, copy.deepcopy(G0))
            if args.warmup > 0:
                pkg.run_solver(eng, opts(args.warmup), 3)
            capi.check(eng.lib.aoadmm_kernel_stats(eng.h, 0, 1, None, None, None, None))
            eng.synchronize()
            t1 = time.perf_counter()
            pkg.run_solver(eng, opts(args.steps), 3)
            eng.synchronize()
            dt64 = time.perf_counter() - t1
            ms64 = C.c_double(); nl64 = C.c_int64(); by64 = C.c_double()
            capi.check(eng.lib.aoadmm_kernel_stats(eng.h, 0, 1, C.byref(ms64), C.byref(nl64), C.byref(by64), None))
            if nl64.value > 0 and ms64.value > 0:
                a64 = ms64.value / nl64.value
                ach64 = by64.value / nl64.value / (a64 * 1e-3) / 1e9
                tr64 = None
                import glob as _glob
                c64 = sorted(_glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_contract_f64.json')))
                if c64 and args.size == 2000 and R == 20:
                    try:
                        tr64 = float(json.load(open(c64[-1]))['hbm_traffic_bytes_per_launch'])
                    except Exception:
                        tr64 = None
                parity_mode = {
                    'what': 'the same workload with the tensor stored and contracted in IEEE double (the reference is double '
                            'throughout): the mode that is compared with the oracle at 1e-8',
                    'iters_s': args.steps / dt64, 'ms_per_step': dt64 / args.steps * 1e3, 'dtype': 'f64',
                    'roofline': {'bound': 'hbm', 'kernel': 'contract_f64', 'avg_launch_ms': a64, 'launches': int(nl64.value),
                                 'algorithmic_bytes_per_launch': by64.value / nl64.value, 'achieved': ach64,
                                 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach64 / HBM_PEAK_GBS, 'traffic': tr64}}
            fac64 = pkg.download_state(eng, Z64, G0)['fac']
            drift = {'factor_rel_fro_f32_vs_f64': [float(np.linalg.norm(a - b) / np.linalg.norm(b)) for a, b in zip(fac32, fac64)],
                     'outer_iterations': args.warmup + args.steps, 'f64_ms_per_step': dt64 / args.steps * 1e3,
                     'note': 'same synthetic tensor (seed 0) stored fp32 vs fp64, same init, same iteration counts'}
        except Exception as e:   # a report, never a reason to lose the headline number
            drift = {'error': repr(e)}

    if rank == 0:
        sx = 4.0 if args.prec == 'f32' else 8.0
        launches = max(int(nl.value), 1)
        avg_ms = ms.value / launches
        bytes_per_launch = by.value / launches
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        flops_mttkrp = 2.0 * I * J * K * R / share
        # HBM bytes per launch from the PMC pass committed under profiles/ (separate rocprofv3 --pmc runs,
        # FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950); only valid for this exact workload
        traffic = None
        import glob
        kname = 'contract16_f32' if args.prec == 'f32' else 'contract_f64'
        cands = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_%s.json' % kname)))   # newest round last
        pmc = cands[-1] if cands else ''
        if world == 1 and share == 1 and args.size == 2000 and R == 20 and pmc:
            try:
                traffic = float(json.load(open(pmc))['hbm_traffic_bytes_per_launch'])
            except Exception:
                traffic = None
        line = {
            'metric': 'AO-ADMM outer iters/sec (+ mode-1 MTTKRP GFLOP/s), %d^3 rank-%d CP' % (args.size, R),
            'value': args.steps / dt, 'unit': 'iters/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
            # everything that is not a tensor pass: T reductions, system builds, ADMM inner loops, Gram matrices
            'replicated_tail_ms': dt / args.steps * 1e3 - (launches + int(nl1.value)) / args.steps * avg_ms,
            'tail_breakdown': tail,
            'scaling': 'strong', 'vs_baseline': None, 'dtype': args.prec, 'data': 'synthetic',
            'config': {'workload': 'cfg5: %dx%dx%d R=%d CP, mode1 TV(0.001), modes2-3 nonneg, %s tensor + fp64 solve, '
                                   'MaxInnerIters=5, tol=0' % (I, J, K, R, args.prec),
                       'sharding': ('mode-1 rows over %d GPU(s), factors replicated' % world) if share == 1 else
                                   ('rank %d of %d on ONE GPU (--as-rank): this rank\'s mode-1 row block and mode-3 slab, every '
                                    'collective issued on a one-rank RCCL communicator; a per-rank timing, not a solve'
                                    % (args.as_rank, args.of)),
                       'tensor_passes_per_iter': round((launches + int(nl1.value)) / args.steps, 2),
                       'resident_copies': 'X(i,j,k) + mode-permuted X(j,k,i) and X(k,i,j): 3 x %.0f GB per node, '
                                          'every pass contracts a trailing mode'
                                          % (I * J * K * (4.0 if args.prec == 'f32' else 8.0) / 1e9)},
            'mttkrp_mode1_gflops': flops_mttkrp / (mttkrp_ms * 1e-3) / 1e9,
            'mttkrp_mode1_ms': mttkrp_ms,
            'mttkrp_mfma_frac_f32_peak': flops_mttkrp / (mttkrp_ms * 1e-3) / 1e12 / (F32_MFMA_PEAK_TF * world),
            'f_tensors_last': out['f_tensors'],
            'rank_share': None if share == 1 else {'rank': args.as_rank, 'of': args.of},
            'fp32_drift': drift,
            'parity_mode': parity_mode,
            'collectives': {'backend': 'rccl' if world > 1 else 'none', 'ranks': comm['comm_ranks'],
                            'rccl_version': comm['rccl_version'], 'librccl': comm['librccl']},
            'datagen_s': t_gen,
            'roofline': {'bound': 'hbm', 'kernel': '%s (tensor x factor partial contraction, trailing modes)'
                                                   % ('contract16_f32' if args.prec == 'f32' else 'contract_f64'),
                         'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': traffic, 'avg_launch_ms': avg_ms, 'launches': launches,
                         'algorithmic_bytes_per_launch': bytes_per_launch,
                         'timed_launches': 'every 4th pass of the timed region is bracketed by HIP events on the library\'s stream '
                                           '(AOADMM_PASS_EVENT_EVERY=1: every pass); avg_launch_ms is their mean',
                         'note': 'per rank; algorithmic bytes = local tensor block read once (s_X per entry) + T written once '
                                 '(s_X*R per unfolding row); traffic = FETCH_SIZE x2 + WRITE_SIZE of a separate rocprofv3 --pmc '
                                 'run (%s)' % (os.path.relpath(pmc, ROOT) if pmc else 'none for this workload')},
            'second_kernel': {'kernel': 'contract_lead16_f32 (leading-mode contraction, LDS-transposed): only used when the '
                                        'mode-permuted second copy of the tensor is switched off or does not fit',
                              'launches': int(nl1.value),
                              'avg_launch_ms': (ms1.value / nl1.value) if nl1.value else None,
                              'achieved_GBps': (by1.value / nl1.value / (ms1.value / nl1.value * 1e-3) / 1e9) if nl1.value and ms1.value > 0 else None},
        }
        if not args.no_cpu_baseline and world == 1 and share == 1:      # reported on rank 0 at N = 1 only
            try:
                line['cpu_baseline'] = cpu_baseline(I, J, K, R, rows=(min(args.cpu_rows, I) if args.cpu_rows > 0 else None))
            except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
                line['cpu_baseline'] = {'value': None, 'unit': 'iters/s', 'cores': 0, 'kind': 'port',
                                        'sample': 'failed: %r' % (e,)}
        print(json.dumps(line), flush=True)
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
