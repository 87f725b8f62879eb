/* CPU restatement (plain C + OpenMP) of the AO-ADMM outer iteration for ONE dense 3-way CP block with per-mode
 * constraints none / non-negativity / TV -- the model family of BASELINE configs 2 and 5.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product path (matlab-code_amd/) never does.  It exists so that the CPU baseline next to the GPU
 * number is compiled, threaded code on the whole workload instead of the numpy oracle extrapolated from a slab.
 * Pinned against the numpy oracle (oracle/aoadmm.py) by tests/test_oracle_c.py; the numpy oracle is in turn pinned by
 * the properties and fixtures DESIGN.md section 2 lists ("parity unpinned": the reference holds no output vectors).
 *
 * Follows (file:line of /root/reference, read as text):
 *   functions/cmtf_fun_AOADMM.m:62-81   initial Gram matrices
 *   :87-155   outer loop over the (uncoupled) modes: mttkrp :97, Hadamard of the Grams :98-103, rho :115, B :116,
 *             +rho/2*I :141, chol :142, ADMM_constrained_only / least squares :134, Gram update :148
 *   :591-623  ADMM_constrained_only ; :1420-1429 update_constraint ; :1079-1096 eval_res_ADMM_constr
 *   :1235-1241 objective through last_mttkrp / last_had ; :1284 regulariser value
 *   functions/constraints_to_prox.m:13-14 (non-negativity), :78-81 (TV: prox_TV(x, eta/rho), value without abs)
 *   MTTKRP: Tensor Toolbox mttkrp (third-party, absent): V(i_n,r) = sum X(i1,i2,i3) prod_{m != n} U_m(i_m,r).
 *   TV prox: L. Condat, "A direct algorithm for 1-D total variation denoising", IEEE SPL 20(11), 2013 (TV_Condat_v2 is
 *   third-party, absent; restated from the paper like oracle/prox.py).
 * The tensor is column-major (first index fastest), float or double; all arithmetic is double.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define CPU_C_NONE 0
#define CPU_C_NONNEG 1
#define CPU_C_TV 19

void aoadmm_cpu_set_threads(int n) {
#ifdef _OPENMP
  if (n >= 1) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int aoadmm_cpu_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

static inline double x_at(const void* X, int f32, int64_t o) {
  return f32 ? (double)((const float*)X)[o] : ((const double*)X)[o];
}

/* out (I_n x R, column-major) = mttkrp(X, {A,B,C}, mode).  One pass over X per call. */
void aoadmm_cpu_mttkrp(const void* X, int f32, int64_t I, int64_t J, int64_t K, int R, const double* A, const double* B,
                       const double* C, int mode, double* out) {
  const int64_t n_out = (mode == 0 ? I : (mode == 1 ? J : K)) * R;
  memset(out, 0, (size_t)n_out * sizeof(double));
  if (mode == 2) {
    /* out(k,r) = sum_j B(j,r) * sum_i X(i,j,k) A(i,r): every k owns its row */
#pragma omp parallel
    {
      double* t = (double*)malloc((size_t)R * sizeof(double));
#pragma omp for schedule(static)
      for (int64_t k = 0; k < K; ++k) {
        for (int64_t j = 0; j < J; ++j) {
          const int64_t o = (j + J * k) * I;
          for (int r = 0; r < R; ++r) {
            const double* a = A + (int64_t)I * r;
            double s = 0.0;
            if (f32) { const float* x = (const float*)X + o; _Pragma("omp simd reduction(+ : s)") for (int64_t i = 0; i < I; ++i) s += (double)x[i] * a[i]; }
            else { const double* x = (const double*)X + o; _Pragma("omp simd reduction(+ : s)") for (int64_t i = 0; i < I; ++i) s += x[i] * a[i]; }
            t[r] = s;
          }
          for (int r = 0; r < R; ++r) out[k + K * r] += B[j + J * r] * t[r];
        }
      }
      free(t);
    }
    return;
  }
  /* modes 0 and 1: threads own ranges of k and private accumulators, added in thread order at the end */
  const int64_t rows = mode == 0 ? I : J;
  int nt = aoadmm_cpu_threads();
  if (nt > K) nt = (int)K;
  double* priv = (double*)calloc((size_t)nt * rows * R, sizeof(double));
#pragma omp parallel num_threads(nt)
  {
#ifdef _OPENMP
    const int tid = omp_get_thread_num();
#else
    const int tid = 0;
#endif
    double* acc = priv + (size_t)tid * rows * R;
    double* w = (double*)malloc((size_t)R * sizeof(double));
    const int64_t k0 = K * tid / nt, k1 = K * (tid + 1) / nt;
    for (int64_t k = k0; k < k1; ++k) {
      for (int64_t j = 0; j < J; ++j) {
        const int64_t o = (j + J * k) * I;
        if (mode == 0) {
          /* out(i,r) += X(i,j,k) * B(j,r) * C(k,r) */
          for (int r = 0; r < R; ++r) w[r] = B[j + J * r] * C[k + K * r];
          for (int r = 0; r < R; ++r) {
            double* a = acc + (int64_t)I * r;
            const double wr = w[r];
            if (f32) { const float* x = (const float*)X + o; _Pragma("omp simd") for (int64_t i = 0; i < I; ++i) a[i] += wr * (double)x[i]; }
            else { const double* x = (const double*)X + o; _Pragma("omp simd") for (int64_t i = 0; i < I; ++i) a[i] += wr * x[i]; }
          }
        } else {
          /* out(j,r) += C(k,r) * sum_i X(i,j,k) A(i,r) */
          for (int r = 0; r < R; ++r) {
            const double* a = A + (int64_t)I * r;
            double s = 0.0;
            if (f32) { const float* x = (const float*)X + o; _Pragma("omp simd reduction(+ : s)") for (int64_t i = 0; i < I; ++i) s += (double)x[i] * a[i]; }
            else { const double* x = (const double*)X + o; _Pragma("omp simd reduction(+ : s)") for (int64_t i = 0; i < I; ++i) s += x[i] * a[i]; }
            acc[j + J * r] += C[k + K * r] * s;
          }
        }
      }
    }
    free(w);
  }
#pragma omp parallel for schedule(static)
  for (int64_t e = 0; e < rows * R; ++e) {
    double s = 0.0;
    for (int t = 0; t < nt; ++t) s += priv[(size_t)t * rows * R + e];
    out[e] = s;
  }
  free(priv);
}

/* sum of squares of the tensor (Znorm_const, cmtf_AOADMM.m:124-131) */
double aoadmm_cpu_normsq(const void* X, int f32, int64_t n) {
  double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
  for (int64_t i = 0; i < n; ++i) { const double x = x_at(X, f32, i); s += x * x; }
  return s;
}

static void gram(const double* F, int64_t rows, int R, double* G) {       /* G = F'F */
  for (int p = 0; p < R; ++p)
    for (int q = p; q < R; ++q) {
      double s = 0.0;
      const double *a = F + rows * p, *b = F + rows * q;
#pragma omp parallel for reduction(+ : s) schedule(static)
      for (int64_t i = 0; i < rows; ++i) s += a[i] * b[i];
      G[p + R * q] = s; G[q + R * p] = s;
    }
}

static int chol_lower(double* M, int R) {                                 /* in place, lower; 0 = ok */
  for (int j = 0; j < R; ++j) {
    double d = M[j + R * j];
    for (int k = 0; k < j; ++k) d -= M[j + R * k] * M[j + R * k];
    if (!(d > 0.0)) return 1;
    d = sqrt(d);
    M[j + R * j] = d;
    for (int i = j + 1; i < R; ++i) {
      double v = M[i + R * j];
      for (int k = 0; k < j; ++k) v -= M[i + R * k] * M[j + R * k];
      M[i + R * j] = v / d;
    }
    for (int i = 0; i < j; ++i) M[i + R * j] = 0.0;
  }
  return 0;
}

/* rows of out = rows of rhs * inv(L*L')   ((A_inner/L')/L, :609) */
static void solve_rows(const double* rhs, int64_t rows, int R, const double* L, double* out) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < rows; ++i) {
    double x[64];
    for (int r = 0; r < R; ++r) x[r] = rhs[i + rows * r];
    for (int r = 0; r < R; ++r) {
      double v = x[r];
      for (int q = 0; q < r; ++q) v -= L[r + R * q] * x[q];
      x[r] = v / L[r + R * r];
    }
    for (int r = R - 1; r >= 0; --r) {
      double v = x[r];
      for (int q = r + 1; q < R; ++q) v -= L[q + R * r] * x[q];
      x[r] = v / L[r + R * r];
    }
    for (int r = 0; r < R; ++r) out[i + rows * r] = x[r];
  }
}

/* Condat's direct 1-D TV algorithm (see the header) */
static void tv1d(const double* y, double* x, int64_t n, double lam) {
  if (n == 0) return;
  if (!(lam > 0.0)) { memcpy(x, y, (size_t)n * sizeof(double)); return; }
  int64_t k = 0, k0 = 0, km = 0, kp = 0;
  double vmin = y[0] - lam, vmax = y[0] + lam, umin = lam, umax = -lam;
  for (;;) {
    if (k == n - 1) {
      if (umin < 0.0) {
        while (k0 <= km) x[k0++] = vmin;
        k = km = kp = k0; vmin = y[k]; umin = lam; umax = vmin + umin - vmax;
      } else if (umax > 0.0) {
        while (k0 <= kp) x[k0++] = vmax;
        k = km = kp = k0; vmax = y[k]; umax = -lam; umin = vmax + umax - vmin;
      } else {
        vmin += umin / (double)(k - k0 + 1);
        while (k0 <= k) x[k0++] = vmin;
        return;
      }
    } else {
      umin += y[k + 1] - vmin;
      if (umin < -lam) {
        while (k0 <= km) x[k0++] = vmin;
        k = km = kp = k0; vmin = y[k]; vmax = vmin + 2.0 * lam; umin = lam; umax = -lam;
      } else {
        umax += y[k + 1] - vmax;
        if (umax > lam) {
          while (k0 <= kp) x[k0++] = vmax;
          k = km = kp = k0; vmax = y[k]; vmin = vmax - 2.0 * lam; umin = lam; umax = -lam;
        } else {
          ++k;
          if (umin >= lam) { km = k; vmin += (umin - lam) / (double)(km - k0 + 1); umin = lam; }
          if (umax <= -lam) { kp = k; vmax += (umax + lam) / (double)(kp - k0 + 1); umax = -lam; }
        }
      }
    }
  }
}

static void prox(int type, double p0, double rho, const double* V, int64_t rows, int R, double* Z) {
  if (type == CPU_C_TV) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int r = 0; r < R; ++r) tv1d(V + rows * r, Z + rows * r, rows, p0 / rho);      /* prox_TV(x, eta/rho) */
    return;
  }
#pragma omp parallel for schedule(static)
  for (int64_t e = 0; e < rows * R; ++e) Z[e] = type == CPU_C_NONNEG ? (V[e] > 0.0 ? V[e] : 0.0) : V[e];
}

static double sumsq_diff(const double* a, const double* b, int64_t n) {    /* ||a - b||^2, b may be NULL */
  double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
  for (int64_t i = 0; i < n; ++i) { const double d = b ? a[i] - b[i] : a[i]; s += d * d; }
  return s;
}

/* One solve: max_outer outer iterations (tolerances of the OUTER loop are the caller's business: this runs a fixed
 * count, i.e. AbsFuncTol = OuterRelTol = 0).  fac/Zc/mu: three column-major I_n x R matrices each, in/out.
 * f_tensors[0..max_outer]: objective before the first and after every iteration; inner_iters[m + 3*(it-1)].
 * Returns 0, or 1 when a system matrix is not positive definite. */
int aoadmm_cpu_solve_cp3(const void* X, int f32, int64_t I, int64_t J, int64_t K, int R, double weight,
                         const int* ctype, const double* cparam, double** fac, double** Zc, double** mu, int max_outer,
                         int max_inner, double tol_pr, double tol_du, double normsq, double* f_tensors,
                         int* inner_iters) {
  const int64_t dims[3] = {I, J, K};
  const int RR = R * R;
  double* G[3];
  double *C = (double*)malloc((size_t)RR * sizeof(double)), *Bm = (double*)malloc((size_t)RR * sizeof(double));
  double* last_had = (double*)malloc((size_t)RR * sizeof(double));
  int64_t maxrows = I > J ? (I > K ? I : K) : (J > K ? J : K);
  double* M = (double*)malloc((size_t)maxrows * R * sizeof(double));        /* A{m} */
  double* last_mttkrp = (double*)malloc((size_t)maxrows * R * sizeof(double));
  double* tmp = (double*)malloc((size_t)maxrows * R * sizeof(double));
  double* Zold = (double*)malloc((size_t)maxrows * R * sizeof(double));
  int last_m = 0, rc = 0;
  for (int m = 0; m < 3; ++m) { G[m] = (double*)malloc((size_t)RR * sizeof(double)); gram(fac[m], dims[m], R, G[m]); }   /* :62-81 */
  /* initial objective: cp_func (cp_func.m:19-56) = w*(||X||^2 - 2<X, model> + ||model||^2) through one MTTKRP */
  {
    aoadmm_cpu_mttkrp(X, f32, I, J, K, R, fac[0], fac[1], fac[2], 2, M);
    double f2 = 0.0, f3 = 0.0;
    for (int64_t e = 0; e < K * R; ++e) f2 += M[e] * fac[2][e];
    for (int e = 0; e < RR; ++e) f3 += G[0][e] * G[1][e] * G[2][e];
    f_tensors[0] = weight * (normsq - 2.0 * f2 + f3);
    for (int m = 0; m < 3; ++m)
      if (ctype[m] == CPU_C_TV)
        for (int r = 0; r < R; ++r) f_tensors[0] += cparam[m] * (fac[m][dims[m] - 1 + dims[m] * r] - fac[m][dims[m] * r]);
  }
  for (int it = 1; it <= max_outer && rc == 0; ++it) {
    for (int m = 0; m < 3 && rc == 0; ++m) {
      const int64_t rows = dims[m];
      const int64_t n = rows * R;
      aoadmm_cpu_mttkrp(X, f32, I, J, K, R, fac[0], fac[1], fac[2], m, M);                  /* :97 */
      for (int e = 0; e < RR; ++e) {                                                         /* :98-103 */
        double c = 1.0;
        for (int j = 0; j < 3; ++j) if (j != m) c *= G[j][e];
        C[e] = c;
      }
      double rho = 0.0;
      for (int r = 0; r < R; ++r) rho += C[r + R * r];
      rho /= R;                                                                              /* :115 */
      memcpy(last_mttkrp, M, (size_t)n * sizeof(double));                                   /* :121 (A*1/w) */
      memcpy(last_had, C, (size_t)RR * sizeof(double));
      last_m = m;
      for (int64_t e = 0; e < n; ++e) M[e] *= weight;
      for (int e = 0; e < RR; ++e) Bm[e] = weight * C[e];                                    /* :116 */
      int inner = 1;
      if (ctype[m] == CPU_C_NONE) {
        if (chol_lower(Bm, R)) { rc = 1; break; }                                            /* :134, B is SPD */
        solve_rows(M, rows, R, Bm, fac[m]);
      } else {
        for (int r = 0; r < R; ++r) Bm[r + R * r] += rho / 2;                                /* :141 */
        if (chol_lower(Bm, R)) { rc = 1; break; }                                            /* :142 */
        double pr = INFINITY, du = INFINITY;
        inner = 1;
        while (inner <= max_inner && (pr > tol_pr || du > tol_du)) {                         /* :600 */
#pragma omp parallel for schedule(static)
          for (int64_t e = 0; e < n; ++e) tmp[e] = M[e] + rho / 2 * (Zc[m][e] - mu[m][e]);   /* :608 */
          solve_rows(tmp, rows, R, Bm, fac[m]);                                              /* :609 */
          memcpy(Zold, Zc[m], (size_t)n * sizeof(double));                                   /* :1421 */
#pragma omp parallel for schedule(static)
          for (int64_t e = 0; e < n; ++e) tmp[e] = fac[m][e] + mu[m][e];
          prox(ctype[m], cparam[m], rho, tmp, rows, R, Zc[m]);                               /* :1425 */
#pragma omp parallel for schedule(static)
          for (int64_t e = 0; e < n; ++e) mu[m][e] += fac[m][e] - Zc[m][e];                  /* :1428 */
          ++inner;
          const double nf = sqrt(sumsq_diff(fac[m], NULL, n));
          pr = sqrt(sumsq_diff(fac[m], Zc[m], n)) / nf;                                      /* :1085 */
          const double sc = sqrt(sumsq_diff(mu[m], NULL, n));
          const double dz = sqrt(sumsq_diff(Zc[m], Zold, n));
          du = sc > 0 ? dz / sc : dz;                                                        /* :1087-1092 */
        }
        inner -= 1;                                                                          /* :622 */
      }
      if (inner_iters) inner_iters[m + 3 * (it - 1)] = inner;
      gram(fac[m], rows, R, G[m]);                                                           /* :148 */
    }
    if (rc) break;
    /* objective (:1235-1241) + regulariser values (:1284; TV value telescopes, no absolute value: constraints_to_prox.m:81) */
    double f2 = 0.0, f3 = 0.0;
    for (int64_t e = 0; e < dims[last_m] * R; ++e) f2 += last_mttkrp[e] * fac[last_m][e];
    for (int e = 0; e < RR; ++e) f3 += last_had[e] * G[last_m][e];
    double f = weight * (normsq - 2.0 * f2 + f3);
    for (int m = 0; m < 3; ++m)
      if (ctype[m] == CPU_C_TV)
        for (int r = 0; r < R; ++r) f += cparam[m] * (fac[m][dims[m] - 1 + dims[m] * r] - fac[m][dims[m] * r]);
    f_tensors[it] = f;
  }
  for (int m = 0; m < 3; ++m) free(G[m]);
  free(C); free(Bm); free(last_had); free(M); free(last_mttkrp); free(tmp); free(Zold);
  return rc;
}

/* Synthetic workload of bench.py generated on the host: X = [[A1,A2,A3]] + noise, A_n uniform[0,1),
 * noise level `noise` relative to ||X||, then X <- X/||X||.  Counter-based generator, so the result does not depend on
 * the thread count.  (Not the device generator's stream: the baseline needs the same shape and statistics, not bits.) */
static inline uint64_t mix64(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
void aoadmm_cpu_synth(float* X, int64_t I, int64_t J, int64_t K, int R, double noise, uint64_t seed, double* A,
                      double* B, double* C) {
  for (int64_t e = 0; e < I * R; ++e) A[e] = (double)(mix64(seed ^ (uint64_t)(e + 1)) >> 11) * (1.0 / 9007199254740992.0);
  for (int64_t e = 0; e < J * R; ++e) B[e] = (double)(mix64(seed ^ (uint64_t)(e + 1) ^ 0x1111ull << 40) >> 11) * (1.0 / 9007199254740992.0);
  for (int64_t e = 0; e < K * R; ++e) C[e] = (double)(mix64(seed ^ (uint64_t)(e + 1) ^ 0x2222ull << 40) >> 11) * (1.0 / 9007199254740992.0);
  double sx = 0.0, sn = 0.0;
#pragma omp parallel for reduction(+ : sx, sn) schedule(static)
  for (int64_t k = 0; k < K; ++k) {
    double w[64];
    for (int64_t j = 0; j < J; ++j) {
      for (int r = 0; r < R; ++r) w[r] = B[j + J * r] * C[k + K * r];
      float* x = X + (j + J * k) * I;
      for (int64_t i = 0; i < I; ++i) {
        double v = 0.0;
        for (int r = 0; r < R; ++r) v += A[i + I * r] * w[r];
        /* approximately normal noise: sum of four uniforms, variance 1 */
        const uint64_t h = mix64(seed ^ (uint64_t)((j + J * k) * I + i) ^ 0x3333ull << 40);
        const double u = ((double)(h & 0xffff) + (double)((h >> 16) & 0xffff) + (double)((h >> 32) & 0xffff) +
                          (double)((h >> 48) & 0xffff)) * (1.0 / 65536.0) - 2.0;
        const double g = u * 1.7320508075688772;
        sx += v * v; sn += g * g;
        x[i] = (float)v;
        /* keep the noise in a second pass: its scale needs ||X|| and ||N|| first */
      }
    }
  }
  const double sc = noise * sqrt(sx) / sqrt(sn);
  double tot = 0.0;
#pragma omp parallel for reduction(+ : tot) schedule(static)
  for (int64_t o = 0; o < I * J * K; ++o) {
    const uint64_t h = mix64(seed ^ (uint64_t)o ^ 0x3333ull << 40);
    const double u = ((double)(h & 0xffff) + (double)((h >> 16) & 0xffff) + (double)((h >> 32) & 0xffff) +
                      (double)((h >> 48) & 0xffff)) * (1.0 / 65536.0) - 2.0;
    const double v = (double)X[o] + sc * u * 1.7320508075688772;
    X[o] = (float)v;
    tot += v * v;
  }
  const float inv = (float)(1.0 / sqrt(tot));
#pragma omp parallel for schedule(static)
  for (int64_t o = 0; o < I * J * K; ++o) X[o] *= inv;
}
