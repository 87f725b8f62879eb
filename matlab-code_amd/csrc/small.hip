// Small dense fp64 primitives (see small.h).  Reference lines are cited at each kernel.
#include "small.h"
#include "device_utils.h"

namespace aoadmm {

#define CTL_GUARD(ctl) \
  if ((ctl) != nullptr && (ctl)->active == 0) return;

__device__ __forceinline__ double coef_val(const Coef& c) { return c.dev ? c.mul * (*c.dev) : c.mul; }

// ---------------------------------------------------------------------------
struct LinArgs {
  const double* x[5];
  Coef c[5];
  int nterms;
};
__global__ void ew_lincomb_k(double* out, int64_t n, LinArgs a, const AdmmCtl* ctl) {
  CTL_GUARD(ctl);
  double cv[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) cv[k] = k < a.nterms ? coef_val(a.c[k]) : 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < 5; ++k)
      if (k < a.nterms) v += cv[k] * a.x[k][i];
    out[i] = v;
  }
}
void ew_lincomb(double* out, int64_t n, int nterms, const Coef* c, const double* const* x,
                const AdmmCtl* ctl, hipStream_t s) {
  AO_REQUIRE(nterms >= 1 && nterms <= 5, "ew_lincomb: bad term count");
  if (n <= 0) return;
  LinArgs a;
  a.nterms = nterms;
  for (int k = 0; k < 5; ++k) {
    a.x[k] = k < nterms ? x[k] : nullptr;
    a.c[k] = k < nterms ? c[k] : Coef{nullptr, 0.0};
  }
  int64_t blocks = cdiv(n, 256);
  if (blocks > 2048) blocks = 2048;
  ew_lincomb_k<<<(unsigned)blocks, 256, 0, s>>>(out, n, a, ctl);
  AO_KERNEL_CHECK();
}

// ---------------------------------------------------------------------------
__global__ void gemm_small_k(double* out, int64_t ldo, const double* A, int64_t lda, const double* B,
                             int64_t ldb, int64_t I, int K, int N, int transB, Coef alpha, double beta,
                             const AdmmCtl* ctl) {
  CTL_GUARD(ctl);
  const double al = coef_val(alpha);
  const int64_t total = I * N;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = idx % I;
    const int n = (int)(idx / I);
    double acc = 0.0;
    for (int k = 0; k < K; ++k) {
      const double b = transB ? B[n + ldb * k] : B[k + ldb * n];
      acc += A[i + lda * k] * b;
    }
    double o = al * acc;
    if (beta != 0.0) o += beta * out[i + ldo * n];
    out[i + ldo * n] = o;
  }
}
void gemm_small(double* out, int64_t ldo, const double* A, int64_t lda, const double* B, int64_t ldb,
                int64_t I, int K, int N, int transB, Coef alpha, double beta, const AdmmCtl* ctl,
                hipStream_t s) {
  if (I * N <= 0) return;
  int64_t blocks = cdiv(I * N, 256);
  if (blocks > 4096) blocks = 4096;
  gemm_small_k<<<(unsigned)blocks, 256, 0, s>>>(out, ldo, A, lda, B, ldb, I, K, N, transB, alpha, beta, ctl);
  AO_KERNEL_CHECK();
}

// ---------------------------------------------------------------------------
// A'B with fixed summation order: block b sums rows [b*rpb, (b+1)*rpb) -> ws[b][K*N]; then one block adds
static constexpr int kAtbRows = 48;   // rows per block, staged through LDS
static int atb_blocks(int64_t I) {
  int64_t nb = cdiv(I, kAtbRows);
  if (nb > 512) nb = 512;
  if (nb < 1) nb = 1;
  return (int)nb;
}
size_t atb_ws_bytes(int64_t I, int K, int N) { return (size_t)atb_blocks(I) * K * N * sizeof(double); }

// block b owns rows [b*rpb, (b+1)*rpb): tiles of 64 rows of A and B are staged in LDS (coalesced
// column reads), every thread accumulates its (k,n) outputs over the tile in a fixed order
__global__ void atb_part_k(const double* A, int64_t lda, const double* B, int64_t ldb, int64_t I, int K, int N,
                           double* ws, const AdmmCtl* ctl) {
  CTL_GUARD(ctl);
  extern __shared__ double tile[];           // [kAtbRows][K] then [kAtbRows][N], row-major, padded by 1
  const int Kp = K + 1, Np = N + 1;
  double* ta = tile;
  double* tb = tile + kAtbRows * Kp;
  const int nb = gridDim.x;
  const int64_t rpb = (I + nb - 1) / nb;
  const int64_t i0 = blockIdx.x * rpb;
  int64_t i1 = i0 + rpb;
  if (i1 > I) i1 = I;
  const int KN = K * N;
  double acc[16];                            // K*N <= 4096 = 16 outputs per thread at 256 threads
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.0;
  for (int64_t r0 = i0; r0 < i1; r0 += kAtbRows) {
    const int nr = (int)((i1 - r0 < kAtbRows) ? (i1 - r0) : kAtbRows);
    __syncthreads();
    for (int e = threadIdx.x; e < nr * K; e += blockDim.x) {
      const int i = e % nr, k = e / nr;
      ta[i * Kp + k] = A[r0 + i + lda * k];
    }
    for (int e = threadIdx.x; e < nr * N; e += blockDim.x) {
      const int i = e % nr, n = e / nr;
      tb[i * Np + n] = B[r0 + i + ldb * n];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int e = threadIdx.x + q * 256;
      if (e < KN) {
        const int k = e % K, n = e / K;
        double a = acc[q];
        for (int i = 0; i < nr; ++i) a += ta[i * Kp + k] * tb[i * Np + n];
        acc[q] = a;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int e = threadIdx.x + q * 256;
    if (e < KN) ws[(int64_t)blockIdx.x * KN + e] = acc[q];
  }
}
__global__ void atb_fin_k(double* out, const double* ws, int nb, int KN, const AdmmCtl* ctl) {
  CTL_GUARD(ctl);
  for (int e = threadIdx.x; e < KN; e += blockDim.x) {
    double t = 0.0;
#pragma unroll 8
    for (int b = 0; b < nb; ++b) t += ws[(int64_t)b * KN + e];     // fixed order; loads issue ahead of the adds
    out[e] = t;
  }
}
void atb_small(double* out, const double* A, int64_t lda, const double* B, int64_t ldb, int64_t I,
               int K, int N, double* ws, const AdmmCtl* ctl, hipStream_t s) {
  const int nb = atb_blocks(I);
  const size_t sh = (size_t)kAtbRows * (K + N + 2) * sizeof(double);
  atb_part_k<<<nb, 256, sh, s>>>(A, lda, B, ldb, I, K, N, ws, ctl);
  AO_KERNEL_CHECK();
  atb_fin_k<<<1, 256, 0, s>>>(out, ws, nb, K * N, ctl);
  AO_KERNEL_CHECK();
}

// ---------------------------------------------------------------------------
template <int MODE>  // 0: sum (x-y)^2 (y nullable), 1: sum x*y
__global__ void reduce_part_k(double* ws, const double* x, const double* y, int64_t n, const AdmmCtl* ctl) {
  CTL_GUARD(ctl);
  __shared__ double sh[256];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (MODE == 0) {
      const double d = y ? x[i] - y[i] : x[i];
      acc += d * d;
    } else {
      acc += x[i] * y[i];
    }
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) ws[blockIdx.x] = sh[0];
}
__global__ void reduce_fin_k(double* slot, const double* ws, int nb, const AdmmCtl* ctl) {
  CTL_GUARD(ctl);
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int b = 0; b < nb; ++b) t += ws[b];
    slot[0] = t;
  }
}
static int red_blocks(int64_t n) {
  int64_t nb = cdiv(n, 2048);
  if (nb > 64) nb = 64;
  if (nb < 1) nb = 1;
  return (int)nb;
}
void sumsq_diff(double* slot, const double* x, const double* y, int64_t n, double* ws, const AdmmCtl* ctl,
                hipStream_t s) {
  const int nb = red_blocks(n);
  if (nb == 1) {
    reduce_part_k<0><<<1, 256, 0, s>>>(slot, x, y, n, ctl);
    AO_KERNEL_CHECK();
    return;
  }
  reduce_part_k<0><<<nb, 256, 0, s>>>(ws, x, y, n, ctl);
  AO_KERNEL_CHECK();
  reduce_fin_k<<<1, 64, 0, s>>>(slot, ws, nb, ctl);
  AO_KERNEL_CHECK();
}
void dot(double* slot, const double* x, const double* y, int64_t n, double* ws, const AdmmCtl* ctl,
         hipStream_t s) {
  const int nb = red_blocks(n);
  if (nb == 1) {
    reduce_part_k<1><<<1, 256, 0, s>>>(slot, x, y, n, ctl);
    AO_KERNEL_CHECK();
    return;
  }
  reduce_part_k<1><<<nb, 256, 0, s>>>(ws, x, y, n, ctl);
  AO_KERNEL_CHECK();
  reduce_fin_k<<<1, 64, 0, s>>>(slot, ws, nb, ctl);
  AO_KERNEL_CHECK();
}

// ---------------------------------------------------------------------------
// per-row forward/backward substitution: x*L' = a (forward), then x*L = y (backward)
template <int RMAX>
__device__ __forceinline__ void row_solve_regs(double (&x)[RMAX], const double* Lsh, int R) {
#pragma unroll
  for (int j = 0; j < RMAX; ++j) {
    if (j < R) {
      double v = x[j];
#pragma unroll
      for (int q = 0; q < j; ++q) v -= Lsh[j + R * q] * x[q];   // L(j,q), column-major
      x[j] = v / Lsh[j + R * j];
    }
  }
#pragma unroll
  for (int j = RMAX - 1; j >= 0; --j) {
    if (j < R) {
      double v = x[j];
#pragma unroll
      for (int q = j + 1; q < RMAX; ++q)
        if (q < R) v -= Lsh[q + R * j] * x[q];                  // L(q,j)
      x[j] = v / Lsh[j + R * j];
    }
  }
}

template <int RMAX>
__global__ void row_solve_k(double* X, int64_t ldx, const double* RHS, int64_t ldr, const double* L, int64_t I,
                            int R, const AdmmCtl* ctl) {
  CTL_GUARD(ctl);
  extern __shared__ double Lsh[];
  for (int e = threadIdx.x; e < R * R; e += blockDim.x) Lsh[e] = L[e];
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= I) return;
  double x[RMAX];
#pragma unroll
  for (int r = 0; r < RMAX; ++r) x[r] = r < R ? RHS[i + ldr * r] : 0.0;
  row_solve_regs<RMAX>(x, Lsh, R);
#pragma unroll
  for (int r = 0; r < RMAX; ++r)
    if (r < R) X[i + ldx * r] = x[r];
}
void row_solve(double* X, int64_t ldx, const double* RHS, int64_t ldr, const double* L, int64_t I, int R,
               const AdmmCtl* ctl, hipStream_t s) {
  if (I <= 0) return;
  const int threads = 128;
  const unsigned blocks = (unsigned)cdiv(I, threads);
  const size_t sh = (size_t)R * R * sizeof(double);
  if (R <= 8) row_solve_k<8><<<blocks, threads, sh, s>>>(X, ldx, RHS, ldr, L, I, R, ctl);
  else if (R <= 16) row_solve_k<16><<<blocks, threads, sh, s>>>(X, ldx, RHS, ldr, L, I, R, ctl);
  else if (R <= 32) row_solve_k<32><<<blocks, threads, sh, s>>>(X, ldx, RHS, ldr, L, I, R, ctl);
  else row_solve_k<64><<<blocks, threads, sh, s>>>(X, ldx, RHS, ldr, L, I, R, ctl);
  AO_KERNEL_CHECK();
}

// ---------------------------------------------------------------------------
__global__ void sys_build_k(SysBuild sb) {
  extern __shared__ double sh[];   // R*R
  __shared__ double tr;
  const int R = sb.R, RR = R * R;
  for (int e = threadIdx.x; e < RR; e += blockDim.x) {
    double c;
    if (sb.ngram == 0) {
      c = sb.Cpre[e];
    } else {
      c = 1.0;                                   // C = ones .* G_transp_G{j}...  (:98-103)
      for (int k = 0; k < sb.ngram; ++k) c *= sb.grams[k][e];
    }
    sb.C[e] = c;
    sh[e] = c;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int r = 0; r < R; ++r) t += sh[r + R * r];
    tr = sb.rho_scale * (t / R);                 // rho = trace(C)/size(C,1)  (:115)
    sb.rho[0] = tr;
  }
  __syncthreads();
  const double rho = tr;
  for (int e = threadIdx.x; e < RR; e += blockDim.x) {
    const int i = e % R, k = e / R;
    double b = sb.w * sh[e];                     // B = w*C (:116)
    if (i == k) b += sb.ridge + sb.bsum_half;    // :117-119, :126
    sb.Bsys[e] = b;
    if (i == k) b += sb.nrho * (rho / 2);        // :141 / :269-271
    sh[e] = b;
  }
  __syncthreads();
  const bool ok = chol_lds(sh, R);               // chol(B','lower') (:142); B symmetric
  if (ok)
    for (int e = threadIdx.x; e < RR; e += blockDim.x) sb.L[e] = sh[e];
  if (ok && sb.Binv) {
    // inv(L*L') column by column: forward then backward substitution on e_j.  The system matrix of
    // an ADMM mode carries +rho/2*I with rho = trace(C)/R, so cond(B) <= 2R+1: the explicit inverse
    // loses nothing measurable and turns the per-row solve into R independent dot products.
    // Linv (lower) in a second LDS matrix: thread j owns column j (forward substitution on e_j),
    // then Binv(i,j) = sum_{k >= max(i,j)} Linv(k,i)*Linv(k,j) with one thread per entry.
    double* Li = sh + RR;
    for (int j = threadIdx.x; j < R; j += blockDim.x) {
      for (int i = 0; i < j; ++i) Li[i + R * j] = 0.0;
      for (int i = j; i < R; ++i) {
        double v = (i == j) ? 1.0 : 0.0;
        for (int q = j; q < i; ++q) v -= sh[i + R * q] * Li[q + R * j];
        Li[i + R * j] = v / sh[i + R * i];
      }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < RR; e += blockDim.x) {
      const int i = e % R, j = e / R;
      double acc = 0.0;
      for (int k = (i > j ? i : j); k < R; ++k) acc += Li[k + R * i] * Li[k + R * j];
      sb.Binv[e] = acc;
    }
  }
  if (threadIdx.x == 0 && sb.ctl) {
    sb.ctl->active = 1;
    sb.ctl->iters = 0;
    if (!ok) sb.ctl->notpd = 1;
    sb.ctl->res[0] = sb.ctl->res[1] = sb.ctl->res[2] = sb.ctl->res[3] = 0.0;
  }
}
void sys_build(const SysBuild& sb, hipStream_t s) {
  AO_REQUIRE(sb.R >= 1 && sb.R <= kMaxRank, "sys_build: bad R");
  sys_build_k<<<1, 256, (size_t)2 * sb.R * sb.R * sizeof(double), s>>>(sb);
  AO_KERNEL_CHECK();
}

__global__ void chol_only_k(double* L, const double* B, int R, AdmmCtl* ctl) {
  extern __shared__ double sh[];
  for (int e = threadIdx.x; e < R * R; e += blockDim.x) sh[e] = B[e];
  __syncthreads();
  const bool ok = chol_lds(sh, R);
  if (ok)
    for (int e = threadIdx.x; e < R * R; e += blockDim.x) L[e] = sh[e];
  if (!ok && threadIdx.x == 0 && ctl) ctl->notpd = 1;
}
__global__ void ctl_reset_k(AdmmCtl* ctl) {
  if (threadIdx.x == 0) {
    ctl->active = 1;
    ctl->iters = 0;
    ctl->res[0] = ctl->res[1] = ctl->res[2] = ctl->res[3] = 0.0;
  }
}
void ctl_reset(AdmmCtl* ctl, hipStream_t s) {
  ctl_reset_k<<<1, 64, 0, s>>>(ctl);
  AO_KERNEL_CHECK();
}
void chol_only(double* L, const double* B, int R, AdmmCtl* ctl, hipStream_t s) {
  chol_only_k<<<1, 256, (size_t)R * R * sizeof(double), s>>>(L, B, R, ctl);
  AO_KERNEL_CHECK();
}

}  // namespace aoadmm
