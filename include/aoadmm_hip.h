/*
 * aoadmm_hip.h -- C ABI of the MI355X-native AO-ADMM engine (libaoadmm_hip.so).
 *
 * This is the drop-in boundary for the hot path of
 * AOADMM-DataFusionFramework/Matlab-Code.  Citations are relative to the
 * reference repository root.
 *
 *   solver level : replaces the call
 *       [Fac,out] = cmtf_fun_AOADMM(Z,Znorm_const,G,fh,gh,lscalar,uscalar,options)
 *       at functions/cmtf_AOADMM.m:193 (signature functions/cmtf_fun_AOADMM.m:1).
 *   op level     : the L2->L1 calls inside that function (mttkrp, Gram,
 *       Cholesky system, ADMM inner loops, prox operators, objective), exported
 *       for unit parity against the CPU oracle.
 *
 * Conventions
 *   - every matrix/tensor crossing the boundary is IEEE double, column-major
 *     (MATLAB layout), passed as plain pointer + 64-bit sizes;
 *   - mode numbers, tensor numbers and coupling ids are 0-based here (the MEX /
 *     ctypes host layer subtracts 1 from MATLAB's numbers); "no coupling" is -1;
 *   - every function returns an int status (AOADMM_OK == 0); the message of the
 *     last failure on the calling thread is returned by aoadmm_last_error();
 *   - no C++ exception crosses this boundary; the library owns all device
 *     memory behind the opaque handle and keeps no pointer into caller memory
 *     after a call returns (functions/cmtf_AOADMM.m value semantics).
 *   - there is NO CPU fallback: without a usable gfx950 device every compute
 *     entry point returns AOADMM_ERR_HIP.
 */
#ifndef AOADMM_HIP_H
#define AOADMM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AOADMM_ABI_VERSION 3

/* ---- status codes ------------------------------------------------------ */
enum {
  AOADMM_OK = 0,
  AOADMM_ERR_INVALID = 1,     /* bad argument / inconsistent model (check_data_input.m) */
  AOADMM_ERR_HIP = 2,         /* HIP runtime failure or no device */
  AOADMM_ERR_NOT_PD = 3,      /* chol() failed: system matrix not positive definite
                                 (cmtf_fun_AOADMM.m:142,185,273,362 throw in MATLAB) */
  AOADMM_ERR_RCCL = 4,        /* collective failure */
  AOADMM_ERR_UNSUPPORTED = 5, /* feature routed back to the MATLAB path:
                                 non-Frobenius loss, 'custom' prox, sptensor */
  AOADMM_ERR_NOMEM = 6
};

/* ---- constraint catalogue: functions/constraints_to_prox.m:13-91 -------- */
enum {
  AOADMM_C_NONE = 0,
  AOADMM_C_NONNEG = 1,          /* :13  {'non-negativity'} */
  AOADMM_C_BOX = 2,             /* :15  {'box',l,u}                 params: l,u */
  AOADMM_C_SIMPLEX_COL = 3,     /* :19  {'simplex column-wise',eta} params: eta */
  AOADMM_C_SIMPLEX_ROW = 4,     /* :22  {'simplex row-wise',eta} */
  AOADMM_C_NONDECREASING = 5,   /* :25 */
  AOADMM_C_NONINCREASING = 6,   /* :27 */
  AOADMM_C_UNIMODAL = 7,        /* :29  {'unimodality',nn}          params: nn (0/1) */
  AOADMM_C_L1_BALL = 8,         /* :32  params: eta */
  AOADMM_C_L2_BALL = 9,         /* :35  params: eta */
  AOADMM_C_NONNEG_L2_BALL = 10, /* :38  params: eta */
  AOADMM_C_NONNEG_L2_SPHERE = 11, /* :41 params: eta (ignored, as in the reference) */
  AOADMM_C_ORTHONORMAL = 12,    /* :44 */
  AOADMM_C_L1_REG = 13,         /* :46  params: eta */
  AOADMM_C_L0_REG = 14,         /* :50 */
  AOADMM_C_L2_REG = 15,         /* :54 */
  AOADMM_C_RIDGE = 16,          /* :58 */
  AOADMM_C_QUADRATIC = 17,      /* :62  params: eta ; matrix L passed separately */
  AOADMM_C_GL_SMOOTH = 18,      /* :68  params: eta */
  AOADMM_C_TV = 19,             /* :78  params: eta */
  AOADMM_C_TPARAFAC2 = 20       /* :82  params: eta (PARAFAC2 B_k mode only) */
};

/* ---- state fields of the struct G: init_coupled_AOADMM_CMTF.m:41-45 ----- */
enum {
  AOADMM_F_FAC = 0,              /* G.fac{m} (or G.fac{m}{k})            index = mode     */
  AOADMM_F_CONSTRAINT_FAC = 1,   /* G.constraint_fac{m}                  index = mode     */
  AOADMM_F_CONSTRAINT_DUAL = 2,  /* G.constraint_dual_fac{m}             index = mode     */
  AOADMM_F_COUPLING_FAC = 3,     /* G.coupling_fac{c}                    index = coupling */
  AOADMM_F_COUPLING_DUAL = 4,    /* G.coupling_dual_fac{m}               index = mode     */
  AOADMM_F_DELTAB = 5,           /* G.DeltaB{p}                          index = tensor   */
  AOADMM_F_P = 6,                /* G.P{p}{k}                            index = tensor   */
  AOADMM_F_MU_DELTAB = 7         /* G.mu_DeltaB{p}{k}                    index = tensor   */
};

/* storage / arithmetic of the big tensor passes */
enum {
  AOADMM_PREC_F64 = 0, /* tensor stored fp64, v_mfma_f64_16x16x4_f64 (parity mode)        */
  AOADMM_PREC_F32 = 1  /* tensor stored fp32, v_mfma_f32_16x16x4_f32 (+ packed-fp32 VALU for 1-4 leftover columns),
                          fp64 everywhere else */
};

typedef struct aoadmm_ctx aoadmm_ctx;

/* options struct: example_script1_CP_PAR2_nonneg.m:110-123, cmtf_fun_AOADMM.m:4-9 */
typedef struct aoadmm_options {
  int32_t MaxOuterIters;
  int32_t MaxInnerIters;
  double AbsFuncTol;
  double OuterRelTol;
  double innerRelPrTol_coupl;
  double innerRelPrTol_constr;
  double innerRelDualTol_coupl;
  double innerRelDualTol_constr;
  int32_t bsum;
  double bsum_weight;
  int32_t iter_start_PAR2Bkconstraint;   /* default 0 (cmtf_fun_AOADMM.m:7-9) */
  int32_t has_increase_factor_rhoBk;     /* isfield(options,'increase_factor_rhoBk') */
  double increase_factor_rhoBk;
  int32_t use_dimtree;                   /* engine option (options.hip.*): reuse partial
                                            contractions between modes; 1 = default */
  int32_t no_permuted_copy;              /* engine option: 1 = do not keep the two mode-permuted resident copies of
                                            3-way tensors (saves twice the tensor's size in HBM; mode-1 contractions then
                                            use the LDS-transposed kernel, mode-2 ones run batched); 0 = default */
  int32_t par2_slab_sharding;            /* engine option, with a communicator: 0 = auto (a PARAFAC2 block is repeated
                                            on every rank unless it has >= 1024 slabs per rank), 1 = shard the slabs
                                            over the ranks, -1 = never.  Blocks with Z.miss or the tPARAFAC2 constraint
                                            are always repeated */
  int32_t reserved[5];
} aoadmm_options;

/* `out` struct of cmtf_fun_AOADMM.m:480-494.  Arrays are caller-allocated with
 * MaxOuterIters+1 entries (innerIters: n_modes*MaxOuterIters, column-major
 * n_modes x MaxOuterIters like out.innerIters). */
typedef struct aoadmm_result {
  double f_tensors, f_couplings, f_constraints, f_PAR2_couplings;
  int32_t OuterIterations;
  int32_t exit_code;            /* 0 = 'maxIterations', 1 = stopping rule met (make_exit_flag.m) */
  int32_t exit_abs[4];          /* per quantity: 1 = 'AbsFuncTol', 0 = 'RelFuncTol' */
  double *func_val_conv, *func_coupl_conv, *func_constr_conv, *func_PAR2_coupl, *time_at_it;
  double *innerIters;
  /* EM missing data (cmtf_fun_AOADMM.m:408-441, :485, :490-492): NaN / untouched without Z.miss */
  double f_rel_missing;
  double *func_rel_missing;     /* MaxOuterIters+1 entries, [0] = NaN; may be NULL */
} aoadmm_result;

/* Progress report of options.Display = 'iter' (cmtf_fun_AOADMM.m:44-59, :462-468): called on the caller's thread
 * from inside aoadmm_solve after the initial evaluation (iter = 0) and after every `every`-th outer iteration with
 * f = {f_tensors, f_couplings, f_constraints, f_PAR2_couplings} and f_rel_missing (NaN without Z.miss).
 * On a multi-device context the rows are produced by rank 0's worker thread and handed to the calling thread,
 * which delivers them while it waits inside aoadmm_solve (a MEX callback may use mexPrintf / drawnow).
 * The callback must not throw and must not call back into the library. */
typedef void (*aoadmm_progress_fn)(void* user, int iter, const double f[4], double f_rel_missing);

/* ---- library / context ------------------------------------------------- */
int aoadmm_abi_version(void);
const char* aoadmm_last_error(void);
int aoadmm_device_count(int* n);
int aoadmm_create(aoadmm_ctx** ctx, int device);
/* One process, several GPUs (a MATLAB session): the context owns one engine and one host thread per
 * listed device, joined by RCCL; every call below is executed by all of them together and returns when the last is
 * done, outputs come from rank 0.  The model is sharded exactly as with one process per GPU.  Listing a device more
 * than once selects the host-staged bring-up transport of aoadmm_comm_init_local (that is how the one-GPU test box
 * runs this path).  aoadmm_comm_init_* are not valid on such a context. */
int aoadmm_create_multi(aoadmm_ctx** ctx, int n_devices, const int* devices);
int aoadmm_destroy(aoadmm_ctx* ctx);
int aoadmm_synchronize(aoadmm_ctx* ctx);
/* fn = NULL or every <= 0 switches the report off (options.DisplayIters is `every`) */
int aoadmm_set_progress(aoadmm_ctx* ctx, aoadmm_progress_fn fn, void* user, int every);

/* Multi-GPU (one process per GPU): rank 0 creates an id, the host layer
 * broadcasts it (torch.distributed / MPI / a file), every rank joins.  The CP
 * tensor is then row-sharded along its first mode; factor matrices
 * are replicated and only MTTKRP partials cross xGMI. */
int aoadmm_comm_unique_id(char id[128]);
int aoadmm_comm_init_rank(aoadmm_ctx* ctx, const char id[128], int rank, int world);
/* Measurement hook (bench.py --as-rank R --of N): the context takes rank `rank` of `world` in every sharding decision
 * (row block of mode 1, mode-3 slab of the mode-1 pass, own-rows buffers) but joins a ONE-rank RCCL communicator, so
 * every collective is issued (ncclAllReduce on the library's stream) without peers.  It times one rank's share of an
 * N-GPU job on a one-GPU box; the sums are that rank's partial sums only, so the factors are NOT those of the N-rank job. */
int aoadmm_comm_init_rank_share(aoadmm_ctx* ctx, const char id[128], int rank, int world);
/* Bring-up/test transport: `world` contexts driven by threads of ONE process (on one device or several) form
 * group `key`; collectives go through host staging in rank order.  It lets the sharded data path run with
 * world > 1 on a single GPU, which RCCL refuses.  Every rank must make the same sequence of library calls. */
int aoadmm_comm_init_local(aoadmm_ctx* ctx, int key, int rank, int world);
int aoadmm_comm_rank(aoadmm_ctx* ctx, int* rank, int* world);
/* What the collectives run on: ncclGetVersion() of the RCCL this process resolved, ncclCommCount() of the context's
 * communicator (0 without one; the group size for the bring-up transport) and the path of the loaded librccl
 * (bench.py reports all three).  Any pointer may be NULL. */
int aoadmm_comm_info(aoadmm_ctx* ctx, int* nccl_version, int* comm_ranks, char* lib_path, int lib_path_cap);

/* ---- model (the struct Z) ---------------------------------------------- */
/* Z.size / Z.modes / Z.model / Z.weights (example_script1_CP_PAR2_nonneg.m:74-89) */
int aoadmm_model_begin(aoadmm_ctx* ctx, int n_modes, int n_tensors, int n_couplings);
int aoadmm_model_set_mode(aoadmm_ctx* ctx, int mode, int64_t rows, int rank);
int aoadmm_model_set_mode_slabs(aoadmm_ctx* ctx, int mode, int K, const int64_t* rows_k, int rank);
int aoadmm_model_add_cp(aoadmm_ctx* ctx, int p, int n_tensor_modes, const int* modes, double weight);
int aoadmm_model_add_par2(aoadmm_ctx* ctx, int p, const int* modes3, double weight);
/* Z.constrained_modes / Z.constraints{m}; Lmat only for AOADMM_C_QUADRATIC (rows x rows) */
int aoadmm_model_set_constraint(aoadmm_ctx* ctx, int mode, int constraint, const double* params,
                                int n_params, const double* Lmat);
/* Z.coupling.lin_coupled_modes(mode)=coupling ; coupl_trafo_matrices{mode} (H, hr x hc) ;
 * coupl_trafo_matrices2{mode} (H2).  Pass NULL/0 when absent. */
int aoadmm_model_set_coupling(aoadmm_ctx* ctx, int mode, int coupling, const double* H, int64_t hr,
                              int64_t hc, const double* H2, int64_t h2r, int64_t h2c);
int aoadmm_model_set_coupling_type(aoadmm_ctx* ctx, int coupling, int type);
int aoadmm_model_set_ridge(aoadmm_ctx* ctx, const double* ridge_per_mode); /* Z.ridge */
int aoadmm_model_end(aoadmm_ctx* ctx);

/* ---- data (Z.object{p}) ------------------------------------------------- */
/* dense CP block: host column-major doubles, dims as given to model_set_mode.
 * With a communicator, every rank passes the FULL array and keeps its row block,
 * or passes only its block with local_rows/row_offset != full. */
int aoadmm_tensor_upload(aoadmm_ctx* ctx, int p, const double* data, int precision);
/* one-process-per-GPU contexts only (a multi-device context takes the full array and shards it itself) */
int aoadmm_tensor_upload_rows(aoadmm_ctx* ctx, int p, const double* block, int64_t row_offset,
                              int64_t local_rows, int precision);
/* PARAFAC2 slab k (I x J_k).  k = AOADMM_ALL_SLABS: the K slabs back to back (I x sum J_k) in one transfer;
 * the same convention holds for aoadmm_par2_slab_mask_upload and for the cell-valued state fields below. */
#define AOADMM_ALL_SLABS (-1)
int aoadmm_par2_slab_upload(aoadmm_ctx* ctx, int p, int k, const double* Xk);
/* Z.miss{p} (cmtf_AOADMM.m:68-121): one byte per entry, 1 = observed, 0 = missing, same shape and
 * column-major order as Z.object{p} (the FULL array also when row-sharded) / as slab k.  Upload after the
 * data.  A block with a mask is handled by EM imputation inside aoadmm_solve (cmtf_fun_AOADMM.m:408-441):
 * the resident copy of the data is overwritten at the missing positions every outer iteration. */
int aoadmm_tensor_mask_upload(aoadmm_ctx* ctx, int p, const uint8_t* mask);
int aoadmm_par2_slab_mask_upload(aoadmm_ctx* ctx, int p, int k, const uint8_t* mask_k);
/* device-side synthetic CP tensor: X = [[A1,..,AN]] + noise, ||X|| = 1;
 * never crosses PCIe.  normsq_out receives ||X||^2 after normalisation. */
int aoadmm_tensor_synth(aoadmm_ctx* ctx, int p, int rank, uint64_t seed, double noise, int precision);
/* Znorm_const{p} (cmtf_AOADMM.m:130-156) */
int aoadmm_tensor_normsq(aoadmm_ctx* ctx, int p, double* out);

/* ---- state (the struct G) ---------------------------------------------- */
/* slab = k for cell-valued fields (PARAFAC2 B mode, P, mu_DeltaB), else 0.  slab = AOADMM_ALL_SLABS moves
 * all K cells at once: host holds them back to back, each J_k x R column-major, rows = sum J_k. */
int aoadmm_state_set(aoadmm_ctx* ctx, int field, int index, int slab, const double* host,
                     int64_t rows, int64_t cols);
int aoadmm_state_get(aoadmm_ctx* ctx, int field, int index, int slab, double* host, int64_t rows,
                     int64_t cols);

/* ---- solver level: cmtf_fun_AOADMM.m:1 ---------------------------------- */
int aoadmm_solve(aoadmm_ctx* ctx, const aoadmm_options* opt, aoadmm_result* out);
/* one bare MTTKRP on the resident tensor p against the current factors (bench leg);
 * result stays on the device; elapsed device time of the kernels is returned */
int aoadmm_resident_mttkrp(aoadmm_ctx* ctx, int p, int tensor_mode, double* out_host_or_null,
                           float* elapsed_ms);
/* device time (ms, HIP events on the library's stream around the kernel only), launch count, algorithmic
 * bytes and flops of a tensor-pass kernel since the last reset.  which = 0: register-streaming contraction
 * (contract_f32/f64, trailing modes); which = 1: LDS-transposed leading-mode contraction (contract_lead_f32);
 * which = 2: the reductions over the partial contraction T that finish an MTTKRP (bytes = size of T per reduction;
 * timed only from the first call with which = 2 on, two more events per reduction) */
int aoadmm_kernel_stats(aoadmm_ctx* ctx, int which, int reset, double* contract_ms, int64_t* contract_launches,
                        double* contract_bytes, double* contract_flops);

/* ---- op level (host in / host out) -------------------------------------- */
/* mttkrp(X,U,n): cmtf_fun_AOADMM.m:97, cp_func.m:47.  X dense, dims[ndims]; U[m] is dims[m] x R */
int aoadmm_op_mttkrp(aoadmm_ctx* ctx, const double* X, int ndims, const int64_t* dims,
                     const double* const* U, int R, int n, int precision, double* out);
/* Y = X_(n)*X_(n)' (dims[n] x dims[n]) of a dense matrix / 3-way tensor: the Gram matrix whose leading eigenvectors
 * initialise mode n when init_options.nvecs = 1 (cmtf_nvecs.m:40-58, init_coupled_AOADMM_CMTF.m:50-73; for a
 * PARAFAC2 block pass [X_1 ... X_K] with n = 0, or X_k with n = 1).  The eigenvectors are taken by the caller
 * (MATLAB `eigs`, numpy `eigh`). */
int aoadmm_op_unfold_gram(aoadmm_ctx* ctx, const double* X, int ndims, const int64_t* dims, int n, int precision,
                          double* out);
/* The same Gram matrix from the RESIDENT data of tensor p (uploaded or generated before): no second transfer of the
 * tensor for init_options.nvecs = 1 (cmtf_nvecs.m:31-56 unfolds the data it already holds).  CP blocks: tensor_mode
 * 0..2, slab ignored; PARAFAC2 blocks: tensor_mode 0 (all slabs side by side) or 1 (slab `slab`).  With a communicator
 * the partial sums of the row blocks are all-reduced; the first mode of a row-sharded block has no local answer:
 * AOADMM_ERR_UNSUPPORTED (use aoadmm_op_unfold_gram with the host array). */
int aoadmm_resident_unfold_gram(aoadmm_ctx* ctx, int p, int tensor_mode, int slab, double* out);
/* G'*G : cmtf_fun_AOADMM.m:66,148 */
int aoadmm_op_gram(aoadmm_ctx* ctx, const double* F, int64_t rows, int R, double* out);
/* L = chol(B','lower') : cmtf_fun_AOADMM.m:142 ; AOADMM_ERR_NOT_PD on failure */
int aoadmm_op_chol(aoadmm_ctx* ctx, const double* B, int R, double* L);
/* prox handle built by constraints_to_prox.m, evaluated as prox(X,rho) */
int aoadmm_op_prox(aoadmm_ctx* ctx, int constraint, const double* params, int n_params,
                   const double* Lmat, const double* X, int64_t rows, int R, double rho, double* out);
/* ADMM_constrained_only (cmtf_fun_AOADMM.m:591-623) for one CP mode: A is the
 * MTTKRP (rows x R), Bsys the system matrix *before* +rho/2*I (line :141 is applied
 * inside), fac/Z/mu updated in place, inner iteration count returned. */
int aoadmm_op_admm_constrained(aoadmm_ctx* ctx, const double* A, const double* Bsys, double rho,
                               int constraint, const double* params, int n_params,
                               const double* Lmat, int64_t rows, int R, int max_inner,
                               double tol_pr, double tol_du, double* fac, double* Z, double* mu,
                               int* inner_iters);

#ifdef __cplusplus
}
#endif
#endif /* AOADMM_HIP_H */
