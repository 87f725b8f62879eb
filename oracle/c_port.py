"""ctypes binding of the C/OpenMP CPU restatement (oracle/c/aoadmm_cpu.c).

TEST INFRASTRUCTURE: imported only by tests/ and bench.py's cpu_baseline leg.  The library covers one dense 3-way CP
block with per-mode constraints none / non-negativity / TV (BASELINE configs 2 and 5); everything else is the numpy
oracle's (oracle/aoadmm.py), against which tests/test_oracle_c.py pins this port."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, '_cbuild', 'libaoadmm_cpu.so')
CTYPES = {None: 0, 'non-negativity': 1, 'TV regularization': 19}
_lib = None


def build():
    """gcc build of the library (oracle/Makefile); no-op when it is up to date."""
    subprocess.run(['make', '-C', HERE, '-s'], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        dp = C.POINTER(C.c_double)
        L.aoadmm_cpu_threads.restype = C.c_int
        L.aoadmm_cpu_set_threads.argtypes = [C.c_int]
        L.aoadmm_cpu_normsq.restype = C.c_double
        L.aoadmm_cpu_normsq.argtypes = [C.c_void_p, C.c_int, C.c_int64]
        L.aoadmm_cpu_mttkrp.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int, dp, dp, dp, C.c_int, dp]
        L.aoadmm_cpu_solve_cp3.restype = C.c_int
        L.aoadmm_cpu_solve_cp3.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_double,
                                           C.POINTER(C.c_int), dp, C.POINTER(dp), C.POINTER(dp), C.POINTER(dp), C.c_int,
                                           C.c_int, C.c_double, C.c_double, C.c_double, dp, C.POINTER(C.c_int)]
        L.aoadmm_cpu_synth.argtypes = [C.POINTER(C.c_float), C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_double,
                                       C.c_uint64, dp, dp, dp]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def threads():
    return int(lib().aoadmm_cpu_threads())


def usable_cpus():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (a container with 256
    visible CPUs and a 16-CPU quota is throttled, not helped, by 128 threads)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except Exception:
            continue
    return n


def set_threads(n):
    lib().aoadmm_cpu_set_threads(int(n))


def _tensor(X):
    X = np.asarray(X)
    if X.dtype not in (np.float32, np.float64):
        X = X.astype(np.float64)
    return np.asfortranarray(X)


def mttkrp(X, facs, mode):
    X = _tensor(X)
    I, J, K = X.shape
    F = [np.asfortranarray(f, dtype=np.float64) for f in facs]
    R = F[0].shape[1]
    out = np.zeros((X.shape[mode], R), order='F')
    lib().aoadmm_cpu_mttkrp(X.ctypes.data, int(X.dtype == np.float32), I, J, K, R, _dp(F[0]), _dp(F[1]), _dp(F[2]), mode, _dp(out))
    return out


def solve_cp3(X, constraints, fac, Zc, mu, max_outer, max_inner, tol_pr=0.0, tol_du=0.0, weight=1.0, normsq=None):
    """Fixed number of outer iterations on one 3-way CP block.  constraints: per mode None | ('non-negativity',) |
    ('TV regularization', eta).  fac / Zc / mu: lists of three I_n x R arrays (copied).  Returns the new
    (fac, Zc, mu), f_tensors[0..max_outer] and innerIters (3 x max_outer)."""
    X = _tensor(X)
    I, J, K = X.shape
    R = fac[0].shape[1]
    f32 = int(X.dtype == np.float32)
    ct = (C.c_int * 3)(*[CTYPES[c[0] if c else None] for c in constraints])
    cp = np.array([float(c[1]) if c and len(c) > 1 else 0.0 for c in constraints])
    F = [np.array(f, dtype=np.float64, order='F') for f in fac]
    Zs = [np.array(z if z is not None else np.zeros_like(f), dtype=np.float64, order='F') for z, f in zip(Zc, F)]
    Ms = [np.array(m if m is not None else np.zeros_like(f), dtype=np.float64, order='F') for m, f in zip(mu, F)]
    dp = C.POINTER(C.c_double)
    arr = lambda xs: (dp * 3)(*[_dp(x) for x in xs])
    if normsq is None:
        normsq = lib().aoadmm_cpu_normsq(X.ctypes.data, f32, X.size)
    ft = np.zeros(max_outer + 1)
    inner = np.zeros((3, max(max_outer, 1)), dtype=np.int32, order='F')
    rc = lib().aoadmm_cpu_solve_cp3(X.ctypes.data, f32, I, J, K, R, float(weight), ct, _dp(cp), arr(F), arr(Zs), arr(Ms),
                                    int(max_outer), int(max_inner), float(tol_pr), float(tol_du), float(normsq), _dp(ft),
                                    inner.ctypes.data_as(C.POINTER(C.c_int)))
    if rc:
        raise np.linalg.LinAlgError('system matrix not positive definite')
    return F, Zs, Ms, ft, inner


def synth(I, J, K, R, noise=0.05, seed=0):
    """bench.py's synthetic workload on the host (fp32 tensor, unit Frobenius norm) and its ground-truth factors."""
    X = np.empty((I, J, K), dtype=np.float32, order='F')
    A, B, Cc = (np.empty((n, R), order='F') for n in (I, J, K))
    lib().aoadmm_cpu_synth(X.ctypes.data_as(C.POINTER(C.c_float)), I, J, K, R, float(noise), int(seed), _dp(A), _dp(B), _dp(Cc))
    return X, [A, B, Cc]
