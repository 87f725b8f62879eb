"""ctypes binding of libaoadmm_hip.so (C ABI: include/aoadmm_hip.h).

This is the same marshalling a MEX gateway does; there is
no CPU fallback: if the library is missing or no GPU is visible, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('AOADMM_LIB_PATH') or os.path.join(_HERE, 'libaoadmm_hip.so')   # the override is for A/B timing of builds

# status codes (include/aoadmm_hip.h)
OK, ERR_INVALID, ERR_HIP, ERR_NOT_PD, ERR_RCCL, ERR_UNSUPPORTED, ERR_NOMEM = range(7)
PREC_F64, PREC_F32 = 0, 1
(F_FAC, F_CONSTRAINT_FAC, F_CONSTRAINT_DUAL, F_COUPLING_FAC, F_COUPLING_DUAL, F_DELTAB, F_P,
 F_MU_DELTAB) = range(8)
ALL_SLABS = -1            # AOADMM_ALL_SLABS
PROGRESS_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_double)   # aoadmm_progress_fn

# every symbol include/aoadmm_hip.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    'aoadmm_abi_version', 'aoadmm_last_error', 'aoadmm_device_count', 'aoadmm_create', 'aoadmm_create_multi', 'aoadmm_destroy',
    'aoadmm_synchronize', 'aoadmm_set_progress', 'aoadmm_comm_unique_id', 'aoadmm_comm_init_rank', 'aoadmm_comm_init_rank_share', 'aoadmm_comm_init_local', 'aoadmm_comm_rank', 'aoadmm_comm_info',
    'aoadmm_model_begin', 'aoadmm_model_set_mode', 'aoadmm_model_set_mode_slabs', 'aoadmm_model_add_cp',
    'aoadmm_model_add_par2', 'aoadmm_model_set_constraint', 'aoadmm_model_set_coupling',
    'aoadmm_model_set_coupling_type', 'aoadmm_model_set_ridge', 'aoadmm_model_end', 'aoadmm_tensor_upload',
    'aoadmm_tensor_upload_rows', 'aoadmm_par2_slab_upload', 'aoadmm_tensor_mask_upload', 'aoadmm_par2_slab_mask_upload', 'aoadmm_tensor_synth', 'aoadmm_tensor_normsq',
    'aoadmm_state_set', 'aoadmm_state_get', 'aoadmm_solve', 'aoadmm_resident_mttkrp', 'aoadmm_kernel_stats',
    'aoadmm_op_mttkrp', 'aoadmm_op_unfold_gram', 'aoadmm_resident_unfold_gram', 'aoadmm_op_gram', 'aoadmm_op_chol', 'aoadmm_op_prox', 'aoadmm_op_admm_constrained',
]


class AoadmmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__('[aoadmm status %d] %s' % (code, msg))
        self.code = code


class NotPositiveDefinite(AoadmmError):
    """chol() failure -- MATLAB raises 'Matrix must be positive definite' (cmtf_fun_AOADMM.m:142)."""


class UnsupportedOnDevice(AoadmmError):
    """Model feature that the host layer must route to the original MATLAB path."""


class Options(C.Structure):
    _fields_ = [
        ('MaxOuterIters', C.c_int32), ('MaxInnerIters', C.c_int32),
        ('AbsFuncTol', C.c_double), ('OuterRelTol', C.c_double),
        ('innerRelPrTol_coupl', C.c_double), ('innerRelPrTol_constr', C.c_double),
        ('innerRelDualTol_coupl', C.c_double), ('innerRelDualTol_constr', C.c_double),
        ('bsum', C.c_int32), ('bsum_weight', C.c_double),
        ('iter_start_PAR2Bkconstraint', C.c_int32), ('has_increase_factor_rhoBk', C.c_int32),
        ('increase_factor_rhoBk', C.c_double), ('use_dimtree', C.c_int32), ('no_permuted_copy', C.c_int32),
        ('par2_slab_sharding', C.c_int32), ('reserved', C.c_int32 * 5),
    ]


class Result(C.Structure):
    _fields_ = [
        ('f_tensors', C.c_double), ('f_couplings', C.c_double), ('f_constraints', C.c_double),
        ('f_PAR2_couplings', C.c_double), ('OuterIterations', C.c_int32), ('exit_code', C.c_int32),
        ('exit_abs', C.c_int32 * 4),
        ('func_val_conv', C.POINTER(C.c_double)), ('func_coupl_conv', C.POINTER(C.c_double)),
        ('func_constr_conv', C.POINTER(C.c_double)), ('func_PAR2_coupl', C.POINTER(C.c_double)),
        ('time_at_it', C.POINTER(C.c_double)), ('innerIters', C.POINTER(C.c_double)),
        ('f_rel_missing', C.c_double), ('func_rel_missing', C.POINTER(C.c_double)),
    ]


_lib = None


def load_library():
    """Load libaoadmm_hip.so; fail loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'libaoadmm_hip.so is missing (%s). Build it with `python -c "import __graft_entry__ as g; g.build()"` '
            'or `make -C matlab-code_amd/csrc`. There is no CPU fallback.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    lib.aoadmm_last_error.restype = C.c_char_p
    dp = C.POINTER(C.c_double)
    i64 = C.c_int64
    vp = C.c_void_p
    lib.aoadmm_create.argtypes = [C.POINTER(vp), C.c_int]
    lib.aoadmm_create_multi.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(C.c_int)]
    lib.aoadmm_destroy.argtypes = [vp]
    lib.aoadmm_synchronize.argtypes = [vp]
    lib.aoadmm_device_count.argtypes = [C.POINTER(C.c_int)]
    lib.aoadmm_set_progress.argtypes = [vp, PROGRESS_FN, vp, C.c_int]
    lib.aoadmm_comm_unique_id.argtypes = [C.c_char_p]
    lib.aoadmm_comm_init_rank.argtypes = [vp, C.c_char_p, C.c_int, C.c_int]
    lib.aoadmm_comm_init_rank_share.argtypes = [vp, C.c_char_p, C.c_int, C.c_int]
    lib.aoadmm_comm_init_local.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    lib.aoadmm_comm_rank.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.aoadmm_comm_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int]
    lib.aoadmm_model_begin.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    lib.aoadmm_model_set_mode.argtypes = [vp, C.c_int, i64, C.c_int]
    lib.aoadmm_model_set_mode_slabs.argtypes = [vp, C.c_int, C.c_int, C.POINTER(i64), C.c_int]
    lib.aoadmm_model_add_cp.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_double]
    lib.aoadmm_model_add_par2.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.c_double]
    lib.aoadmm_model_set_constraint.argtypes = [vp, C.c_int, C.c_int, dp, C.c_int, dp]
    lib.aoadmm_model_set_coupling.argtypes = [vp, C.c_int, C.c_int, dp, i64, i64, dp, i64, i64]
    lib.aoadmm_model_set_coupling_type.argtypes = [vp, C.c_int, C.c_int]
    lib.aoadmm_model_set_ridge.argtypes = [vp, dp]
    lib.aoadmm_model_end.argtypes = [vp]
    lib.aoadmm_tensor_upload.argtypes = [vp, C.c_int, dp, C.c_int]
    lib.aoadmm_tensor_upload_rows.argtypes = [vp, C.c_int, dp, i64, i64, C.c_int]
    lib.aoadmm_par2_slab_upload.argtypes = [vp, C.c_int, C.c_int, dp]
    lib.aoadmm_tensor_mask_upload.argtypes = [vp, C.c_int, C.POINTER(C.c_uint8)]
    lib.aoadmm_par2_slab_mask_upload.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_uint8)]
    lib.aoadmm_tensor_synth.argtypes = [vp, C.c_int, C.c_int, C.c_uint64, C.c_double, C.c_int]
    lib.aoadmm_tensor_normsq.argtypes = [vp, C.c_int, dp]
    lib.aoadmm_state_set.argtypes = [vp, C.c_int, C.c_int, C.c_int, dp, i64, i64]
    lib.aoadmm_state_get.argtypes = [vp, C.c_int, C.c_int, C.c_int, dp, i64, i64]
    lib.aoadmm_solve.argtypes = [vp, C.POINTER(Options), C.POINTER(Result)]
    lib.aoadmm_resident_mttkrp.argtypes = [vp, C.c_int, C.c_int, dp, C.POINTER(C.c_float)]
    lib.aoadmm_kernel_stats.argtypes = [vp, C.c_int, C.c_int, dp, C.POINTER(i64), dp, dp]
    lib.aoadmm_op_mttkrp.argtypes = [vp, dp, C.c_int, C.POINTER(i64), C.POINTER(dp), C.c_int, C.c_int, C.c_int, dp]
    lib.aoadmm_op_unfold_gram.argtypes = [vp, dp, C.c_int, C.POINTER(i64), C.c_int, C.c_int, dp]
    lib.aoadmm_resident_unfold_gram.argtypes = [vp, C.c_int, C.c_int, C.c_int, dp]
    lib.aoadmm_op_gram.argtypes = [vp, dp, i64, C.c_int, dp]
    lib.aoadmm_op_chol.argtypes = [vp, dp, C.c_int, dp]
    lib.aoadmm_op_prox.argtypes = [vp, C.c_int, dp, C.c_int, dp, dp, i64, C.c_int, C.c_double, dp]
    lib.aoadmm_op_admm_constrained.argtypes = [vp, dp, dp, C.c_double, C.c_int, dp, C.c_int, dp, i64, C.c_int,
                                               C.c_int, C.c_double, C.c_double, dp, dp, dp, C.POINTER(C.c_int)]
    for name in SYMBOLS:
        fn = getattr(lib, name)
        if name != 'aoadmm_last_error':
            fn.restype = C.c_int
    _lib = lib
    return lib


def check(status):
    if status == OK:
        return
    msg = load_library().aoadmm_last_error().decode('utf-8', 'replace')
    if status == ERR_NOT_PD:
        raise NotPositiveDefinite(status, msg)
    if status == ERR_UNSUPPORTED:
        raise UnsupportedOnDevice(status, msg)
    raise AoadmmError(status, msg)


def as_f(a):
    """Column-major (MATLAB layout) float64 copy/view."""
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None
