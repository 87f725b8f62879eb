"""CPU oracle for the AO-ADMM hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This package is a numpy/fp64 restatement of the reference MATLAB algorithm
(`/root/reference/functions/*.m`), written from the source text.  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it, and only as the checker / reported CPU baseline.  The product path
(`matlab-code_amd/`) never imports it and fails loudly when its HIP library
is missing.

PARITY STATUS: **parity unpinned** for everything that lives in the
reference's un-vendored third-party dependencies (the reference ships no
tests, golden vectors or recorded outputs, and MATLAB/Octave are absent):

  * MATLAB Tensor Toolbox v3.1 (`README.md:7`): `mttkrp`, `ktensor`, `full`,
    `norm`.  Restated from the published definition (Kolda & Bader 2009) and
    anchored on the call sites `functions/cmtf_fun_AOADMM.m:97`,
    `functions/cp_func.m:47`; pinned by a brute-force triple sum.
  * Proximity Operator Repository (`README.md:8`, version unpinned):
    `project_box/simplex/monotone/L1/L2`, `prox_abs/zero/L2`
    (`functions/constraints_to_prox.m:14-56`).  Each is a unique Euclidean
    projection / proximity operator; pinned by feasibility + first-order
    optimality tests.
  * `TV_Condat_v2` (`functions/prox_TV.m:3-7`): unique minimiser of
    0.5||x-y||^2 + lam*sum|x[i+1]-x[i]|; restated from Condat (2013) and pinned
    by KKT conditions and a second independent solver.

Everything that *is* in the reference (`cmtf_fun_AOADMM.m`,
`constraints_to_prox.m`, `project_unimodal_vector.m`, ...) is restated line by
line with `file:line` citations; the only reference-held data fixtures
(`functions_for_example_scripts/noisy_dataset.mat`, `gnd_factors.mat`) are
inputs, not expected outputs, and are used as inputs in `tests/`.

`oracle/c/aoadmm_cpu.c` (+ `oracle/c_port.py`, `oracle/Makefile`) is a compiled C/OpenMP restatement of the same outer
iteration for one dense 3-way CP block with constraints none / non-negativity / TV: the CPU baseline `bench.py` times
beside the GPU number.  Same status (test infrastructure, parity unpinned); `tests/test_oracle_c.py` pins it to this
package.
"""
