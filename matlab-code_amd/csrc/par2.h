// PARAFAC2 slab kernels (functions/cmtf_fun_AOADMM.m:157-250 block updates, :509-589 ADMM_B_Parafac2,
// :1247-1268 / :1351-1362 objective pieces).  A PARAFAC2 block is K slabs X_k (I x J_k, ragged J_k);
// slabs are stored back to back, slab k at element offset I*off[k] (column-major I x J_k); slab-valued
// factors (B_k, P_k, mu_k, Z_k ...) are stored back to back too, slab k at off[k]*R (column-major J_k x R).
// All kernels run one workgroup per slab; sizes are small (cfg4: I=40, J_k<=120, R=3, K=256), so these
// are latency-bound and plain fp64 VALU code.
#pragma once
#include <functional>

#include "admm.h"
#include "common.h"
#include "small.h"

namespace aoadmm {

struct P2Dims {
  int K, I, R;
  const int64_t* off;   // device, K+1 prefix sums of J_k
  const int64_t* off_h; // the same on the host
  int64_t Jtot;
  int Jmax;
  int k0, k1;           // slabs this rank works on: [k0, k1) = [0, K) unless the block is slab-sharded
};
// all-reduce hook for slab-sharded blocks (Engine::allreduce on the library's stream)
using P2AllReduce = std::function<void(double*, int64_t)>;

// T1[k] = X_k * B_k  (I x R each)
void par2_xkb(const double* X, const double* B, const P2Dims& d, double* T1, hipStream_t s);
// GB[k] = B_k' * B_k  (R x R each)
void par2_gram(const double* B, const P2Dims& d, double* GB, hipStream_t s);
// Amt(i,r) = sum_k T1[k](i,r)*C(k,r) ; Csys(r,q) = sum_k C(k,r)C(k,q)GB[k](r,q)        (:160-165)
void par2_modeA_combine(const double* T1, const double* Cfac, const double* GB, const P2Dims& d, double* Amt,
                        double* Csys, hipStream_t s);
// Ak = w * X_k' * A * D_k  (J_k x R, concatenated)                                       (:193)
void par2_xta(const double* X, const double* A, const double* Cfac, double w, const P2Dims& d, double* Ak,
              hipStream_t s);
// per slab: C_k = D_k GA D_k, rho_k, B_k = w C_k + rho_k/2 (1+constr) I + ridge + bsum/2, L_k = chol   (:194-212)
void par2_b_system(const double* GA, const double* Cfac, double w, double ridge, double bsum_half, double rho_scale,
                   int nrho, const P2Dims& d, double* rho, double* L, AdmmCtl* ctl, hipStream_t s);

struct P2BArgs {
  const double *Ak, *L, *rho;          // rhs, chol factors [K][R*R], rho[K]
  double *B, *P, *Pold, *mu, *W;       // concatenated J_k x R
  double *DeltaB, *DeltaBold, *part;   // R x R, R x R, [K][R*R]
  const double *Z, *muZ;               // constraint variables (nullable)
  double* norms;                       // [K][8]
  int use_constr;
  // One-sided Jacobi of the polar factor (:532-534), warm start: the rotation the previous inner iteration ended with
  // ([K][R*R], column-major per slab) is applied first, so the sweeps start from almost orthogonal columns.  A speed
  // hint only (any orthogonal start gives the same polar factor).  null: cold start from the identity.
  double* Jrot = nullptr;
  int jrot_valid = 0;                  // 0: Jrot holds nothing yet (first inner iteration): cold start, then store
};
// one inner iteration of ADMM_B_Parafac2 up to (not including) the constraint update   (:525-547, :582-585)
// psum (R*R+1 doubles) != nullptr: slabs are sharded, DeltaB's sums go through `allreduce`
void par2_b_iteration(const P2BArgs& a, const P2Dims& d, const AdmmCtl* ctl, hipStream_t s, double* psum,
                      const P2AllReduce& allreduce);
// The whole loop without B_k constraints on unsharded slabs, R <= 8: two launches per inner iteration (the sum over the
// slabs and the while test ride at the head of the next kernel) and one closing launch; ctl must have been reset.
bool par2_b_loop_folded_ok(const P2Dims& d, bool constrained, bool sharded);
void par2_b_loop_folded(const P2BArgs& a, const P2Dims& d, AdmmCtl* ctl, int max_inner, double tol_pr_coupl,
                        double tol_pr_constr, double tol_du_coupl, double tol_du_constr, hipStream_t s);
// Z_k = prox(B_k + muZ_k, rho_k) ; muZ_k += B_k - Z_k ; norms[k][4..6] = ||B-Z||^2, ||muZ||^2, ||Z-Zold||^2   (:566-579)
void par2_b_constraint(const ProxSpec& ps, const double* B, double* Z, double* muZ, double* Zold, double* V,
                       const double* rho, const P2Dims& d, double* prox_ws, double* norms, const AdmmCtl* ctl,
                       hipStream_t s);
// residual averages over the slabs + loop condition (:520, :558-585)
void par2_b_finalize(const double* norms, const P2Dims& d, int use_constr, AdmmCtl* ctl, int max_inner,
                     double tol_pr_coupl, double tol_pr_constr, double tol_du_coupl, double tol_du_constr, hipStream_t s,
                     double* part4, const P2AllReduce& allreduce);

// mode C: a(k,r) = w * sum_i A(i,r) T1[k](i,r) ; C_k = GA .* GB[k] ; rho_k ; B_k (+rho_k/2 I if constrained) ; chol  (:221-240)
// nrho = how many rho_k/2*I terms the system gets (constraint, exact coupling :262-264); Madd (R x R, optional) enters
// as + rho_k/2*Madd (coupling type 2: H*H', :307); raw = 1: L receives B_k itself
void par2_c_system(const double* A, const double* T1, const double* GA, const double* GB, double w, double ridge,
                   double bsum_half, int nrho, int raw, const P2Dims& d, const double* Cfac, double* a, double* rho,
                   double* L, AdmmCtl* ctl, hipStream_t s, const double* Madd = nullptr);
// rhomax = max_k rho_k (:1424); separate because a slab-sharded block gathers rho first
void par2_rho_max(const double* rho, int K, double* rhomax, hipStream_t s, double* rhomean = nullptr,
                  double* rhosum = nullptr);
// (K*R) x (K*R) system of a C mode coupled through H*C = Delta (:283-293); HtH = H'*H (K x K), rhoC = mean(rho) on the device
// the same system when H'H = diag(d): K independent row systems, L (holding B_k) is overwritten by chol(B_k + rhoC/2*(d_k [+1])*I)
void par2_c_rowsys_diag(double* L, const double* d, const double* rhoC, int constrained, int K, int R, AdmmCtl* ctl,
                        hipStream_t s);
void par2_c_big_system(const double* Bk, const double* HtH, const double* rhoC, int constrained, int K, int R, double* M,
                       hipStream_t s);
// row k: rhs = a_k (+ rho_k/2 (Z(k,:) - mu(k,:))) ; C(k,:) = L_k'\(L_k\rhs)      (:236, :604-605)
void par2_c_rowsolve(const double* a, const double* rho, const double* L, const double* Z, const double* mu,
                     int use_admm, const P2Dims& d, double* Cfac, const AdmmCtl* ctl, hipStream_t s);

// res[k] = ||X_k - A D_k B_k'||_F^2                                                       (:1262-1264)
void par2_residual(const double* X, const double* A, const double* B, const double* Cfac, const P2Dims& d,
                   double* res, hipStream_t s);
// regv[k] = reg_func(B_k) of a regularisation-type constraint on the B_k mode            (:1279-1281)
void par2_reg_values(const double* B, int type, double eta, const P2Dims& d, double* regv, hipStream_t s);
// q[k][0..3] = ||B_k - P_k DeltaB||^2, ||B_k||^2, ||B_k - Z_k||^2 (Z nullable), 0          (:1355, :1337)
void par2_b_gaps(const double* B, const double* P, const double* DeltaB, const double* Z, const P2Dims& d,
                 double* q, hipStream_t s);

// out[0] = 1 if any of the three loop-control records carries the not-positive-definite flag
void par2_collect_notpd(const AdmmCtl* a, const AdmmCtl* b, const AdmmCtl* c, double* out, hipStream_t s);

}  // namespace aoadmm
