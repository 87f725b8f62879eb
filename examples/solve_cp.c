/* Plain-C caller of libaoadmm_hip.so: the calls a MEX gateway (or any other FFI) makes for
 *   [Zhat,Fac,G,out] = cmtf_AOADMM(Z,'alg_options',options,'init',G)
 * with one CP tensor, non-negativity on every mode (example_script3/10 family, scaled down).
 * Build:  gcc -std=c99 -O2 -I include examples/solve_cp.c -L matlab-code_amd -laoadmm_hip -lm -o solve_cp
 * Run:    LD_LIBRARY_PATH=matlab-code_amd ./solve_cp            (needs an MI355X; there is no CPU path) */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "aoadmm_hip.h"

#define CHECK(call)                                                                        \
  do {                                                                                     \
    int rc_ = (call);                                                                      \
    if (rc_ != AOADMM_OK) {                                                                \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, aoadmm_last_error());           \
      return 1;                                                                            \
    }                                                                                      \
  } while (0)

static double urand(unsigned long long* s) { /* xorshift64*, enough for a demo */
  *s ^= *s >> 12; *s ^= *s << 25; *s ^= *s >> 27;
  return (double)((*s * 2685821657736338717ull) >> 11) / 9007199254740992.0;
}

static void progress(void* user, int iter, const double f[4], double f_rel_missing) {
  (void)user; (void)f_rel_missing;
  printf("%6d %12f %12f %12f %17f %12f\n", iter, f[0] + f[1] + f[2] + f[3], f[0], f[1], f[2], f[3]);
}

int main(void) {
  enum { I = 30, J = 24, K = 18, R = 3 };
  const int64_t dims[3] = {I, J, K};
  unsigned long long seed = 88172645463325252ull;
  double *A[3], *X = malloc(sizeof(double) * I * J * K);
  double nrm = 0.0;
  int m, r;
  int64_t i, j, k;
  for (m = 0; m < 3; ++m) {                                /* ground truth, X = [[A1,A2,A3]] / ||X|| */
    A[m] = malloc(sizeof(double) * dims[m] * R);
    for (i = 0; i < dims[m] * R; ++i) A[m][i] = urand(&seed);
  }
  for (k = 0; k < K; ++k)
    for (j = 0; j < J; ++j)
      for (i = 0; i < I; ++i) {
        double v = 0.0;
        for (r = 0; r < R; ++r) v += A[0][i + I * r] * A[1][j + J * r] * A[2][k + K * r];
        X[i + I * (j + J * k)] = v;
        nrm += v * v;
      }
  for (i = 0; i < (int64_t)I * J * K; ++i) X[i] /= sqrt(nrm);

  aoadmm_ctx* ctx = NULL;
  CHECK(aoadmm_create(&ctx, 0));
  /* struct Z */
  CHECK(aoadmm_model_begin(ctx, 3, 1, 0));
  const int modes[3] = {0, 1, 2};
  for (m = 0; m < 3; ++m) CHECK(aoadmm_model_set_mode(ctx, m, dims[m], R));
  CHECK(aoadmm_model_add_cp(ctx, 0, 3, modes, 1.0));
  for (m = 0; m < 3; ++m) CHECK(aoadmm_model_set_constraint(ctx, m, AOADMM_C_NONNEG, NULL, 0, NULL));
  CHECK(aoadmm_model_end(ctx));
  CHECK(aoadmm_tensor_upload(ctx, 0, X, AOADMM_PREC_F64));
  /* struct G: random start, Z = prox(fac) = fac (non-negative), duals uniform */
  for (m = 0; m < 3; ++m) {
    double* F = malloc(sizeof(double) * dims[m] * R);
    double* U = malloc(sizeof(double) * dims[m] * R);
    for (i = 0; i < dims[m] * R; ++i) { F[i] = urand(&seed); U[i] = urand(&seed); }
    CHECK(aoadmm_state_set(ctx, AOADMM_F_FAC, m, 0, F, dims[m], R));
    CHECK(aoadmm_state_set(ctx, AOADMM_F_CONSTRAINT_FAC, m, 0, F, dims[m], R));
    CHECK(aoadmm_state_set(ctx, AOADMM_F_CONSTRAINT_DUAL, m, 0, U, dims[m], R));
    free(F); free(U);
  }
  /* options + out */
  aoadmm_options o;
  memset(&o, 0, sizeof o);
  o.MaxOuterIters = 400; o.MaxInnerIters = 5;
  o.AbsFuncTol = 1e-12; o.OuterRelTol = 1e-9;
  o.innerRelPrTol_coupl = o.innerRelPrTol_constr = o.innerRelDualTol_coupl = o.innerRelDualTol_constr = 1e-5;
  o.use_dimtree = 1;
  double* fv = calloc((size_t)o.MaxOuterIters + 1, sizeof(double));
  aoadmm_result res;
  memset(&res, 0, sizeof res);
  res.func_val_conv = fv;
  CHECK(aoadmm_set_progress(ctx, progress, NULL, 100));
  CHECK(aoadmm_solve(ctx, &o, &res));
  printf("outer iterations %d, exit %s, f_tensors %.3e (||X|| = 1), f_constraints %.3e\n", res.OuterIterations,
         res.exit_code ? "stopping rule" : "maxIterations", res.f_tensors, res.f_constraints);
  /* Fac.fac{1} back */
  double* F0 = malloc(sizeof(double) * I * R);
  CHECK(aoadmm_state_get(ctx, AOADMM_F_FAC, 0, 0, F0, I, R));
  double mn = F0[0];
  for (i = 1; i < I * R; ++i) mn = F0[i] < mn ? F0[i] : mn;
  printf("min(Fac.fac{1}) = %.3e\n", mn);
  printf(res.f_tensors < 1e-6 ? "RESULT ok\n" : "RESULT poor fit\n");
  CHECK(aoadmm_destroy(ctx));
  free(F0); free(fv); free(X);
  for (m = 0; m < 3; ++m) free(A[m]);
  return 0;
}
