"""MI355X-native AO-ADMM engine: host-side mirror of the reference interface.

The directory name follows the reference repository (`Matlab-Code` + `_amd`) and
is not a valid Python identifier; import it with
`importlib.import_module('matlab-code_amd')` (see `__graft_entry__.py`).

Public names mirror the reference (`functions/cmtf_AOADMM.m`,
`functions/init_coupled_AOADMM_CMTF.m`, `functions/constraints_to_prox.m`);
compute goes through `libaoadmm_hip.so` (hand-written gfx950 kernels behind the
C ABI of `include/aoadmm_hip.h`).  No CPU fallback exists: importing works
without a GPU (so the C ABI can be inspected), creating an `Engine` does not.
"""
from ._capi import (AoadmmError, NotPositiveDefinite, UnsupportedOnDevice, LIB_PATH, SYMBOLS, load_library)
from .engine import CONSTRAINT_IDS, Engine, constraint_descriptor, default_engine
from .dist import init_engine_comm, row_block
from .driver import (build_model, cmtf_AOADMM, cmtf_nvecs, constraints_to_prox, download_state, init_coupled_AOADMM_CMTF,
                     run_solver, upload_state)

__all__ = ['AoadmmError', 'NotPositiveDefinite', 'UnsupportedOnDevice', 'LIB_PATH', 'SYMBOLS', 'load_library',
           'CONSTRAINT_IDS', 'Engine', 'constraint_descriptor', 'default_engine', 'build_model', 'cmtf_AOADMM',
           'cmtf_nvecs', 'constraints_to_prox', 'download_state', 'init_coupled_AOADMM_CMTF', 'run_solver', 'upload_state', 'init_engine_comm', 'row_block']
