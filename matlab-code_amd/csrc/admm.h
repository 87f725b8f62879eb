// ADMM inner-loop kernels and the proximal operator catalogue
// (functions/cmtf_fun_AOADMM.m:591-623,1420-1429 ; functions/constraints_to_prox.m:13-91).
#pragma once
#include <vector>
#include "common.h"
#include "small.h"
#include "loopctl.h"

namespace aoadmm {

struct ProxSpec {
  int type = AOADMM_C_NONE;
  double p0 = 0.0, p1 = 0.0;       // constraint parameters (eta | l,u | nn)
  // AOADMM_C_QUADRATIC only (device pointers, see QuadPrep): L = U diag(w) U'
  const double* Lmat = nullptr;
  const double* LU = nullptr;
  const double* LUt = nullptr;
  const double* Lw = nullptr;
  struct QuadPrep* quad = nullptr; // non-symmetric L: the prox refreshes (2*eta/rho*L + I)^-1 through it when rho moved
};

// 'quadratic regularization' (constraints_to_prox.m:62-67): prox(x,rho) = (2*eta/rho*L + I) \ x with a fixed
// user matrix L.  rho changes every outer iteration, L does not: L is diagonalised once on the host
// (symmetric L required) and the prox becomes U * diag(1/(2*eta/rho*w_i + 1)) * U' * x, two small GEMMs.
// A non-symmetric L has no orthogonal eigenbasis: whenever 2*eta/rho may have moved (once per outer iteration: rho is
// fixed inside an inner loop; the engine sets `dirty`) the prox inverts 2*eta/rho*L + I ON THE DEVICE by Gauss-Jordan
// elimination with row pivoting (what MATLAB's `\` pivots on; admm.hip quad_gj_step_k, rho read from device memory, no
// host synchronisation) and is then one small GEMM.  The reference factorises the same matrix in EVERY prox call.
struct QuadPrep {
  DevBuf L, U, Ut, w;
  DevBuf Minv, Mwork, piv, pval;   // non-symmetric L: the two buffers the elimination alternates between, pivot rows, pivot value
  bool nonsym = false;
  bool dirty = true;               // the cached inverse may belong to another rho
  const double* key_rho = nullptr; // what the cached inverse was built for
  double key_mul = 0.0, key_eta = 0.0;
  const double* result = nullptr;
  int64_t n = 0;
  void build(const double* L_host, int64_t rows, hipStream_t s);
  void attach(ProxSpec& ps) {
    ps.Lmat = L.d(); ps.LU = U.d(); ps.LUt = Ut.d(); ps.Lw = w.d();
    ps.quad = nonsym ? this : nullptr;
  }
  // (g*L + I)^-1 for g = 2*eta/(rho*rho_mul), column-major n x n on the device; rebuilt when dirty or asked for another rho
  const double* refresh(double eta, const double* rho_dev, double rho_mul, hipStream_t s);
};

// iso.hip: workgroup-parallel isotonic / unimodal projections and the path-Laplacian smoothness prox
// mode 0: non-decreasing, 1: non-increasing, 2: unimodal (nonneg != 0: with projection onto x >= 0)
size_t iso_ws_bytes(int64_t rows, int R);
void prox_iso(const double* V, int64_t ldv, double* Z, int64_t ldz, int64_t rows, int R, int mode, double nonneg,
              double* ws, const AdmmCtl* ctl, hipStream_t s);
// beyond 4096 rows the reduction runs in `ws` (prox_gl_ws_doubles); false: no workspace (the caller solves sequentially)
bool prox_gl_pcr(const double* V, int64_t ldv, double* Z, int64_t ldz, int64_t rows, int R, double eta, const double* rho,
                 double rho_mul, const AdmmCtl* ctl, hipStream_t s, double* ws = nullptr);
size_t prox_gl_ws_doubles(int64_t rows, int R);

bool prox_is_fusable(int type);
// constraints with a reg_func entry (constraints_to_prox.m): their value enters f_tensors (cmtf_fun_AOADMM.m:1272-1288)
inline bool prox_has_reg_value(int t) {
  return t == AOADMM_C_L1_REG || t == AOADMM_C_L0_REG || t == AOADMM_C_L2_REG || t == AOADMM_C_RIDGE ||
         t == AOADMM_C_GL_SMOOTH || t == AOADMM_C_TV || t == AOADMM_C_QUADRATIC;
}     // element-wise or row-wise: folded into the primal kernel
size_t prox_ws_bytes(int type, int64_t rows, int R);

// Z_out = prox(V, rho) for any catalogue entry; rho read from device memory.
void prox_apply(const ProxSpec& ps, const double* V, int64_t ldv, double* Zout, int64_t ldz,
                int64_t rows, int R, const double* rho_dev, double rho_mul, double* ws,
                const AdmmCtl* ctl, hipStream_t s, const double* warm = nullptr, int64_t ldw = 0);
// `warm`: optional previous value of the prox output (same shape); only a speed hint (TV warm start)

// The whole ADMM_constrained_only loop (:596-622) for a CP mode, enqueued without host synchronisation:
//   fusable prox : one row-parallel kernel per inner iteration (fac, Z, mu, residual partial sums);
//   other prox   : primal kernel -> column/matrix prox kernel -> dual kernel per inner iteration.
// The loop condition is evaluated on the device at the head of each iteration (admm.hip admm_continue).
// `part` holds admm_partials(rows) * 4 doubles; `V`,`Znew` are rows*R scratch (non-fusable only).
struct AdmmMode {
  const double* A;      // MTTKRP (+bsum term)            rows x R
  const double* L;      // chol factor                     R x R
  const double* Binv = nullptr;   // inv(L*L') when available (well-conditioned ADMM systems)
  const double* rho;    // device scalar
  double *fac, *Z, *mu; // rows x R each
  int64_t rows;
  int R;
  ProxSpec prox;
};
int admm_partials(int64_t rows);
// `deferred_end`: when given, the closing evaluation of the loop is not launched; the caller hands the
// record to the next kernel it runs anyway (atb_small's `close`).
// `gf` (optional): when the loop takes the two-launch row-local path its second launch also leaves the partial Gram
// matrices of the new factor in gf->ws ([nb][R*R], nb returned) and the row-major copy in gf->At; the caller finishes
// with atb_fin.  gf->nb stays 0 when the path was not taken.
struct GramFold {
  double* ws = nullptr;
  double* At = nullptr;
  int nb = 0;
};
void admm_constrained_loop(const AdmmMode& m, double* part, double* V, double* Znew, double* prox_ws,
                           AdmmCtl* ctl, int max_inner, double tol_pr, double tol_du, hipStream_t s,
                           LoopEnd* deferred_end = nullptr, GramFold* gf = nullptr);

// One-workgroup form of the same loop for short modes (admm.hip admm_loop_wg_k): rows <= 256 and R <= 16; element-/row-wise
// prox or the column-norm family.  One launch for the loop, the Gram matrix and the
// row-major copy of the new factor.
struct WgLoopU {
  const double* A;          // right-hand side, rows x R column-major (ld = rows)
  const double* Binv;       // inv(L*L'), R x R (shared system; unused with per_row)
  const double* L;          // with per_row: the rows' Cholesky factors, [rows][R*R]
  const double* rho;        // device scalar, or one value per row with per_row
  const double* rho_prox;   // device scalar the prox sees (max(rho) for a PARAFAC2 C mode, :1423-1424); null with
                            // per_row: the kernel takes the maximum of the rows' rho itself
  double *fac, *Z, *mu;     // rows x R each
  int64_t rows;
  int R;
  int per_row;              // 1: row k has its own system L_k and rho_k (:602-606)
  int ptype;
  double p0, p1;
  int max_inner;
  double tol_pr, tol_du;
  AdmmCtl* ctl;             // active != 0 on entry; receives iters, res[1], res[3], active = 0
  int reset = 0;            // 1: run whatever ctl holds (the caller would otherwise launch ctl_reset first)
  double* gram;             // out: fac'*fac (R x R), or null
  double* facT;             // out: row-major copy of fac (rows x R), or null
};
bool admm_loop_wg_ok(int64_t rows, int R, int ptype, int max_inner);
void admm_loop_wg(const WgLoopU& a, hipStream_t s);

// generic pieces for the coupled / PARAFAC2 loops -------------------------------
// (Z,mu) <- update_constraint (:1420-1429): Zold kept in `Zold`; slots[0..3] receive
// ||fac-Z||^2, ||fac||^2, ||mu||^2, ||Z-Zold||^2
void constraint_update(const ProxSpec& ps, const double* fac, double* Z, double* mu, double* Zold,
                       double* V, int64_t rows, int R, const double* rho_dev, double rho_mul,
                       double* prox_ws, double* slots, double* red_ws, const AdmmCtl* ctl,
                       hipStream_t s);

// residual bookkeeping at the end of a generic inner iteration; per participating mode a block of
// 8 slots: [0]=||fac-Z||^2 [1]=||fac||^2 [2]=||mu||^2 [3]=||Z-Zold||^2
//          [4]=||Tf(fac)-Sd(Delta)||^2 [5]=||mu_Delta||^2 [6]=||Sd(Delta-Delta_old)||^2
//          [7]=denominator of the primal coupling residual (||fac||^2, or ||Tf(fac)||^2 for types 1, 2)
struct FinalizeArgs {
  const double* slots[8];
  int constrained[8];
  int coupled[8];
  int nmodes;
  int max_inner;
  double tol_pr_coupl, tol_pr_constr, tol_du_coupl, tol_du_constr;
};
void admm_finalize_generic(const FinalizeArgs& fa, AdmmCtl* ctl, hipStream_t s);

}  // namespace aoadmm
