// Small dense fp64 device primitives used by the ADMM inner loops: everything
// here works on factor-sized operands (I_n x R, R x R), column-major, and is
// launch/latency bound; all reductions use a fixed summation order so repeated
// runs are bit-identical.
#pragma once
#include "common.h"

namespace aoadmm {

// control block of one ADMM inner loop, resident in device memory; every
// kernel that takes `ctl` returns immediately when ctl->active == 0, which lets
// the host enqueue MaxInnerIters iterations without synchronising
// (the reference tests the residuals on the host each iteration,
// cmtf_fun_AOADMM.m:600,633).
struct AdmmCtl {
  int active;
  int iters;
  int notpd;      // set by sys_build when chol fails
  int pad;
  double res[4];  // pr_coupl, pr_constr, du_coupl, du_constr of the last iteration
};

// coefficient = mul * (dev ? *dev : 1)
struct Coef {
  const double* dev;
  double mul;
};
inline Coef coef(double m) { return Coef{nullptr, m}; }
inline Coef coef(const double* d, double m) { return Coef{d, m}; }

// out[i] = sum_k c_k * x_k[i]   (k < nterms <= 5; x_k may alias out)
void ew_lincomb(double* out, int64_t n, int nterms, const Coef* c, const double* const* x,
                const AdmmCtl* ctl, hipStream_t s);
// out(I x N) = alpha * A(I x K) * op(B) + beta * out ;  B is K x N (transB=0) or N x K (transB=1)
void gemm_small(double* out, int64_t ldo, const double* A, int64_t lda, const double* B, int64_t ldb,
                int64_t I, int K, int N, int transB, Coef alpha, double beta, const AdmmCtl* ctl,
                hipStream_t s);
// out = A' * B  (A: I x K, B: I x N, K,N <= 64) deterministic two-stage reduction; ws >= gram_ws_bytes
size_t atb_ws_bytes(int64_t I, int K, int N);
struct LoopEnd;
// `At_rowmajor` (optional, I x K doubles): row-major copy of A written on the way (the T reductions read the factor
// in that order); `close` (optional): closing record of the ADMM loop that produced A, evaluated by block 0.
void atb_small(double* out, const double* A, int64_t lda, const double* B, int64_t ldb, int64_t I,
               int K, int N, double* ws, const AdmmCtl* ctl, hipStream_t s, double* At_rowmajor = nullptr,
               const LoopEnd* close = nullptr);
// out = sum of nb partial K x N matrices left in ws by a producer other than atb_part_k (fixed order)
void atb_fin(double* out, const double* ws, int nb, int KN, const AdmmCtl* ctl, hipStream_t s);
// slot[0] = sum (x-y)^2 (y may be null) ; ws >= 64 doubles
void sumsq_diff(double* slot, const double* x, const double* y, int64_t n, double* ws,
                const AdmmCtl* ctl, hipStream_t s);
// Several independent reductions in one launch.  RT_SUMSQ_DIFF: sum (x-y)^2 (y may be null) over n;
// RT_DOT: sum x.*y over n; RT_REG: value of the regulariser `aux` (AOADMM_C_* id: l1, l0, ridge, TV,
// GL smoothness; constraints_to_prox.m reg_func) of the rows x R matrix x.  slot[0] = scale * sum.
enum { RT_SUMSQ_DIFF = 0, RT_DOT = 1, RT_REG = 2 };
struct ReduceTask {
  const double* x = nullptr;
  const double* y = nullptr;
  int64_t n = 0, rows = 0;
  double* slot = nullptr;
  double scale = 1.0;
  int kind = RT_SUMSQ_DIFF, R = 0, aux = 0;
};
constexpr int kReduceBatchMax = 40;
constexpr int kReduceBatchSplit = 64;
struct ReduceBatch {
  ReduceTask t[kReduceBatchMax];
  int n = 0;
  void add(const ReduceTask& k) { if (n < kReduceBatchMax) t[n] = k; ++n; }   // overflow is caught by reduce_batch
};
// ws >= kReduceBatchMax * kReduceBatchSplit doubles
void reduce_batch(const ReduceBatch& rb, double* ws, hipStream_t s);
// X <- RHS * inv(L*L')   (L lower R x R; per-row forward/backward substitution, cmtf_fun_AOADMM.m:609)
void row_solve(double* X, int64_t ldx, const double* RHS, int64_t ldr, const double* L, int64_t I, int R,
               const AdmmCtl* ctl, hipStream_t s);

// System build for one mode (cmtf_fun_AOADMM.m:98-127,141-142):
//   C = had_k grams[k]   (ngram factors; ngram = 0 -> C = C_in given in `Cpre`)
//   rho = trace(C)/R ; Bsys = w*C + ridge*I + bsum/2*I ; L = chol(Bsys + nrho*rho/2*I)
// Writes C (last_had), rho (device scalar), Bsys, L; resets ctl (active=1, iters=0); sets ctl->notpd.
struct SysBuild {
  const double* grams[8];
  int ngram;
  const double* Cpre;
  double w, ridge, bsum_half, rho_scale;   // rho_scale: options.increase_factor_rhoBk (1 otherwise)
  int nrho;
  int R;
  double *C, *rho, *Bsys, *L;
  double* Binv = nullptr;   // optional: inv(L*L') (used by the fused ADMM row kernel)
  const double* Madd = nullptr;   // optional R x R: the factored matrix also gets + rho/2 * Madd (coupling type 2: H*H')
  AdmmCtl* ctl;
};
void sys_build(const SysBuild& sb, hipStream_t s);
// active=1, iters=0 (notpd is sticky)
void ctl_reset(AdmmCtl* ctl, hipStream_t s);
// L = chol(B) only (B symmetric R x R); flag -> ctl->notpd
void chol_only(double* L, const double* B, int R, AdmmCtl* ctl, hipStream_t s);

// B = V diag(w) V' for a symmetric n x n matrix, n <= 64 (cyclic Jacobi, one wave)
void sym_eig_small(const double* B, int n, double* w, double* V, hipStream_t s, const AdmmCtl* ctl = nullptr);
// BB <- AA \ BB, AA symmetric positive definite q x q (destroyed), BB q x nrhs; failure -> ctl->notpd
void spd_solve_left(double* AA, int64_t q, double* BB, int nrhs, AdmmCtl* ctl, hipStream_t s);
// Dense SPD system of order n <= kDenseMaxN: M (n x n, column-major) is overwritten by its Cholesky factor, Minv
// receives inv(M); ctl->notpd is raised if a pivot is not positive.  dense_symv_rows applies Minv to the rows of a
// K x R matrix taken back to back (n = K*R).
constexpr int kDenseMaxN = 2048;
void dense_spd_inverse(double* M, double* Minv, int n, AdmmCtl* ctl, hipStream_t s);
void dense_symv_rows(const double* Minv, const double* rhs, double* out, int K, int R, const AdmmCtl* ctl,
                     hipStream_t s);
// W(i,j) /= rho/2*(lam_mul*lam[i] + shift_mul) + mu[j]
void sylv_scale(double* W, int64_t rows, int R, const double* lam, const double* mu, const double* rho, double lam_mul,
                double shift_mul, const AdmmCtl* ctl, hipStream_t s);

}  // namespace aoadmm
