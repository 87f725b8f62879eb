// Small dense fp64 primitives (see small.h).  Reference lines are cited at each kernel.
#include "small.h"
#include "small_dev.h"
#include "device_utils.h"
#include "loopctl.h"

#include <algorithm>

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
static constexpr int kAtbOneBlockRows = 256;   // up to here one block walks all tiles and writes the result itself
static int atb_blocks(int64_t I) {
  if (I <= kAtbOneBlockRows) return 1;
  int64_t nb = cdiv(I, kAtbRows);
  if (nb > 512) nb = 512;
  if (nb < 1) nb = 1;
  return (int)nb;
}
size_t atb_ws_bytes(int64_t I, int K, int N) { return (size_t)atb_blocks(I) * K * N * sizeof(double); }

// block b owns rows [b*rpb, (b+1)*rpb): tiles of 64 rows of A and B are staged in LDS (coalesced
// column reads), every thread accumulates its (k,n) outputs over the tile in a fixed order
__global__ void atb_part_k(const double* A, int64_t lda, const double* B, int64_t ldb, int64_t I, int K, int N,
                           double* ws /* a single block: the result itself */, const AdmmCtl* ctl, double* At, LoopEnd le) {
  CTL_GUARD(ctl);
  if (le.ctl != nullptr && blockIdx.x == 0 && threadIdx.x < 64) loop_end_eval(le);   // first wave of block 0
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
    // staging: four loads per thread issued before the first LDS store (one load per step made every step wait for its
    // own round trip); a Gram matrix (A == B) stages its rows once
    auto stage = [&](const double* M, int64_t ldm, int C, int Cp, double* dst) {
      const int n = nr * C, st = blockDim.x;
      int e = threadIdx.x;
      for (; e + 3 * st < n; e += 4 * st) {
        const int e1 = e + st, e2 = e + 2 * st, e3 = e + 3 * st;
        const double v0 = M[r0 + e % nr + ldm * (e / nr)], v1 = M[r0 + e1 % nr + ldm * (e1 / nr)];
        const double v2 = M[r0 + e2 % nr + ldm * (e2 / nr)], v3 = M[r0 + e3 % nr + ldm * (e3 / nr)];
        dst[(e % nr) * Cp + e / nr] = v0; dst[(e1 % nr) * Cp + e1 / nr] = v1;
        dst[(e2 % nr) * Cp + e2 / nr] = v2; dst[(e3 % nr) * Cp + e3 / nr] = v3;
      }
      for (; e < n; e += st) dst[(e % nr) * Cp + e / nr] = M[r0 + e % nr + ldm * (e / nr)];
    };
    const bool same = A == B && lda == ldb && K == N;
    stage(A, lda, K, Kp, ta);
    if (!same) stage(B, ldb, N, Np, tb);
    __syncthreads();
    if (same) tb = ta;
    if (At)                                        // row-major copy of the staged rows (contiguous store)
      for (int e = threadIdx.x; e < nr * K; e += blockDim.x) {
        const int i = e / K, k = e - i * K;
        At[(r0 + i) * K + k] = ta[i * Kp + k];
      }
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
// 32 entries x 8 slices per block: slice sl adds partials sl, sl+8, ... (independent loads, one memory
// round trip), then the 8 slice sums are added in a fixed order
__global__ __launch_bounds__(256) void atb_fin_k(double* out, const double* ws, int nb, int KN, const AdmmCtl* ctl) {
  CTL_GUARD(ctl);
  __shared__ double sh[8][32];
  const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + el;
  double t = 0.0;
  if (e < KN) {
    int b = sl;
    for (; b + 24 < nb; b += 32) {                 // four partials per step, loads issued together, added in order
      const double v0 = ws[(int64_t)b * KN + e], v1 = ws[(int64_t)(b + 8) * KN + e];
      const double v2 = ws[(int64_t)(b + 16) * KN + e], v3 = ws[(int64_t)(b + 24) * KN + e];
      t += v0; t += v1; t += v2; t += v3;
    }
    for (; b < nb; b += 8) t += ws[(int64_t)b * KN + e];
  }
  sh[sl][el] = t;
  __syncthreads();
  if (sl == 0 && e < KN) {
    double tot = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) tot += sh[q][el];
    out[e] = tot;
  }
}
void atb_small(double* out, const double* A, int64_t lda, const double* B, int64_t ldb, int64_t I,
               int K, int N, double* ws, const AdmmCtl* ctl, hipStream_t s, double* At_rowmajor, const LoopEnd* close) {
  const int nb = atb_blocks(I);
  const size_t sh = (size_t)kAtbRows * (K + N + 2) * sizeof(double);
  atb_part_k<<<nb, 256, sh, s>>>(A, lda, B, ldb, I, K, N, nb == 1 ? out : ws, ctl, At_rowmajor, close ? *close : LoopEnd());
  AO_KERNEL_CHECK();
  if (nb == 1) return;                               // short matrices: one launch
  atb_fin_k<<<(unsigned)cdiv(K * N, 32), 256, 0, s>>>(out, ws, nb, K * N, ctl);
  AO_KERNEL_CHECK();
}

void atb_fin(double* out, const double* ws, int nb, int KN, const AdmmCtl* ctl, hipStream_t s) {
  atb_fin_k<<<(unsigned)cdiv(KN, 32), 256, 0, s>>>(out, ws, nb, KN, ctl);
  AO_KERNEL_CHECK();
}

// ---------------------------------------------------------------------------
// batch of independent reductions in one launch (the objective evaluation needs ~10 of them per outer
// iteration; one kernel each was ~10 us of latency apiece).  grid = (nsplit, ntasks); fixed summation order.
__global__ __launch_bounds__(256) void reduce_batch_k(ReduceBatch rb, double* ws) {
  __shared__ double sh4[4];
  const ReduceTask& tk = rb.t[blockIdx.y];
  const int nsplit = gridDim.x, sp = blockIdx.x;
  double acc = 0.0;
  if (tk.kind == RT_SUMSQ_DIFF || tk.kind == RT_DOT) {
    const int64_t per = (tk.n + nsplit - 1) / nsplit;
    const int64_t e0 = sp * per;
    int64_t e1 = e0 + per;
    if (e1 > tk.n) e1 = tk.n;
    // four independent accumulators per thread: four (eight) loads in flight instead of one
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    int64_t e = e0 + threadIdx.x;
    if (tk.kind == RT_DOT) {
      for (; e + 768 < e1; e += 1024) {
        a0 += tk.x[e] * tk.y[e]; a1 += tk.x[e + 256] * tk.y[e + 256];
        a2 += tk.x[e + 512] * tk.y[e + 512]; a3 += tk.x[e + 768] * tk.y[e + 768];
      }
      for (; e < e1; e += 256) a0 += tk.x[e] * tk.y[e];
    } else if (tk.y) {
      for (; e + 768 < e1; e += 1024) {
        const double d0 = tk.x[e] - tk.y[e], d1 = tk.x[e + 256] - tk.y[e + 256];
        const double d2 = tk.x[e + 512] - tk.y[e + 512], d3 = tk.x[e + 768] - tk.y[e + 768];
        a0 += d0 * d0; a1 += d1 * d1; a2 += d2 * d2; a3 += d3 * d3;
      }
      for (; e < e1; e += 256) { const double d = tk.x[e] - tk.y[e]; a0 += d * d; }
    } else {
      for (; e + 768 < e1; e += 1024) {
        const double d0 = tk.x[e], d1 = tk.x[e + 256], d2 = tk.x[e + 512], d3 = tk.x[e + 768];
        a0 += d0 * d0; a1 += d1 * d1; a2 += d2 * d2; a3 += d3 * d3;
      }
      for (; e < e1; e += 256) { const double d = tk.x[e]; a0 += d * d; }
    }
    acc = (a0 + a1) + (a2 + a3);
  } else {
    // regulariser values of constraints_to_prox.m (:50,:54,:58 and the difference forms :75,:81), column by column
    const int64_t per = (tk.rows + nsplit - 1) / nsplit;
    const int64_t i0 = sp * per;
    int64_t i1 = i0 + per;
    if (i1 > tk.rows) i1 = tk.rows;
    // (column, row) flattened and four entries per step with their loads issued together: column by column with one
    // entry per step this was R * rows / 256 dependent round trips (the 30 us that made this kernel the longest of
    // the objective evaluation)
    const int64_t nloc = i1 > i0 ? i1 - i0 : 0, tot_e = nloc * tk.R;
    const bool diff = tk.aux == AOADMM_C_TV || tk.aux == AOADMM_C_GL_SMOOTH;
    auto term = [&](double v, double vn, bool has_next) {
      switch (tk.aux) {
        case AOADMM_C_L1_REG: return fabs(v);
        case AOADMM_C_L0_REG: return (v != 0.0) ? 1.0 : 0.0;
        case AOADMM_C_RIDGE: return v * v;
        case AOADMM_C_TV: return has_next ? vn - v : 0.0;                              // no abs(): quirk of :81
        case AOADMM_C_GL_SMOOTH: return has_next ? (vn - v) * (vn - v) : 0.0;
        default: return 0.0;
      }
    };
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    int64_t e = threadIdx.x;
    for (; e + 768 < tot_e; e += 1024) {
      double v[4], vn[4];
      bool hn[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t ee = e + 256 * q;
        const int64_t r = ee / nloc, i = i0 + (ee - r * nloc);
        const double* x = tk.x + tk.rows * r;
        hn[q] = i + 1 < tk.rows;
        v[q] = x[i];
        vn[q] = (diff && hn[q]) ? x[i + 1] : 0.0;
      }
      a0 += term(v[0], vn[0], hn[0]); a1 += term(v[1], vn[1], hn[1]);
      a2 += term(v[2], vn[2], hn[2]); a3 += term(v[3], vn[3], hn[3]);
    }
    for (; e < tot_e; e += 256) {
      const int64_t r = e / nloc, i = i0 + (e - r * nloc);
      const double* x = tk.x + tk.rows * r;
      const bool hn = i + 1 < tk.rows;
      a0 += term(x[i], (diff && hn) ? x[i + 1] : 0.0, hn);
    }
    acc = (a0 + a1) + (a2 + a3);
  }
  const double tot = block256_sum(acc, sh4);
  if (threadIdx.x == 0) {
    if (nsplit == 1) tk.slot[0] = tk.scale * tot;
    else ws[(int64_t)blockIdx.y * nsplit + sp] = tot;
  }
}
// one wave per task (the task record is then addressed by a uniform index: a per-thread index into the by-value
// argument made every thread copy the whole batch to scratch first); lanes hold the partial sums, DPP tree
__global__ __launch_bounds__(64) void reduce_batch_fin_k(ReduceBatch rb, const double* ws, int nsplit) {
  const ReduceTask& tk = rb.t[blockIdx.x];
  double t = 0.0;
  for (int q = threadIdx.x; q < nsplit; q += 64) t += ws[(int64_t)blockIdx.x * nsplit + q];
  t = wave_sum(t);
  if (threadIdx.x == 0) tk.slot[0] = tk.scale * t;
}
void reduce_batch(const ReduceBatch& rb, double* ws, hipStream_t s) {
  if (rb.n <= 0) return;
  AO_REQUIRE(rb.n <= kReduceBatchMax, "reduce_batch: too many tasks");
  int64_t big = 0;
  for (int k = 0; k < rb.n; ++k) big = std::max<int64_t>(big, rb.t[k].kind >= RT_REG ? rb.t[k].rows * rb.t[k].R : rb.t[k].n);
  int nsplit = (int)std::min<int64_t>(kReduceBatchSplit, cdiv(big, 2048));   // two steps of four loads per thread: the kernel is latency, not bandwidth
  if (nsplit < 1) nsplit = 1;
  reduce_batch_k<<<dim3((unsigned)nsplit, (unsigned)rb.n), 256, 0, s>>>(rb, ws);
  AO_KERNEL_CHECK();
  if (nsplit > 1) {
    reduce_batch_fin_k<<<(unsigned)rb.n, 64, 0, s>>>(rb, ws, nsplit);
    AO_KERNEL_CHECK();
  }
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
// One wave, thread t <-> row t of the R x R system, every matrix held in registers (fully unrolled,
// RMAX >= R); values move between lanes with v_readlane broadcasts, so the R Cholesky steps cost a few
// dozen cycles each instead of LDS round trips and barriers (this kernel is on the critical path of
// every mode update: the 256-thread LDS version took ~60 us at R = 20, this one a few us).
template <int RMAX, bool EXACT>
__global__ __launch_bounds__(64) void sys_build_k(SysBuild sb) { sys_build_dev<RMAX, EXACT>(sb); }
void sys_build(const SysBuild& sb, hipStream_t s) {
  AO_REQUIRE(sb.R >= 1 && sb.R <= kMaxRank, "sys_build: bad R");
  static_assert(kMaxRank <= 64, "sys_build_k maps one lane to one row");
#define AO_SB(RM) { if (sb.R == RM) sys_build_k<RM, true><<<1, 64, 0, s>>>(sb); else sys_build_k<RM, false><<<1, 64, 0, s>>>(sb); }
  if (sb.R <= 4) AO_SB(4)
  else if (sb.R <= 8) AO_SB(8)
  else if (sb.R <= 12) AO_SB(12)
  else if (sb.R <= 16) AO_SB(16)
  else if (sb.R <= 20) AO_SB(20)
  else if (sb.R <= 32) AO_SB(32)
  else AO_SB(64)
#undef AO_SB
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

// ---------------------------------------------------------------------------
// Dense SPD systems of a few hundred unknowns (the (K*R) x (K*R) system of a PARAFAC2 C mode coupled through a
// transformation matrix, cmtf_fun_AOADMM.m:282-297): Cholesky in global memory by one workgroup (right-looking,
// column by column), then the explicit inverse, one workgroup per column of the identity, so that every inner
// iteration is a single matrix-vector product instead of 2n dependent substitution steps.
constexpr int kDenseThreads = 1024;
__global__ __launch_bounds__(kDenseThreads) void dense_chol_k(double* M, int n, AdmmCtl* ctl) {
  __shared__ int bad;
  if (threadIdx.x == 0) bad = 0;
  __syncthreads();
  for (int j = 0; j < n; ++j) {
    double* cj = M + (int64_t)n * j;
    if (threadIdx.x == 0) {
      const double d = cj[j];
      if (!(d > 0.0)) bad = 1;
      cj[j] = sqrt(d);
    }
    __syncthreads();
    if (bad) break;
    const double piv = cj[j];
    __syncthreads();                                 // every thread has read the pivot before the column is scaled
    for (int i = j + 1 + threadIdx.x; i < n; i += kDenseThreads) cj[i] /= piv;
    __syncthreads();
    // trailing update of the lower triangle: M(i,c) -= L(i,j) * L(c,j), c = j+1..n-1, i = c..n-1
    const int m = n - j - 1;
    for (int64_t e = threadIdx.x; e < (int64_t)m * m; e += kDenseThreads) {
      const int c = j + 1 + (int)(e / m), i = j + 1 + (int)(e % m);
      if (i >= c) M[i + (int64_t)n * c] -= cj[i] * cj[c];
    }
    __syncthreads();
  }
  if (bad && threadIdx.x == 0 && ctl) ctl->notpd = 1;
}
// column `blockIdx.x` of inv(L*L'): forward and backward substitution of a unit vector, x in LDS
__global__ __launch_bounds__(256) void dense_inverse_k(const double* L, int n, double* Minv) {
  extern __shared__ double x[];
  const int col = blockIdx.x;
  for (int i = threadIdx.x; i < n; i += blockDim.x) x[i] = (i == col) ? 1.0 : 0.0;
  __syncthreads();
  for (int j = col; j < n; ++j) {                    // L y = e_col  (y_j = 0 for j < col)
    const double* cj = L + (int64_t)n * j;
    if (threadIdx.x == 0) x[j] /= cj[j];
    __syncthreads();
    const double xj = x[j];
    for (int i = j + 1 + threadIdx.x; i < n; i += blockDim.x) x[i] -= cj[i] * xj;
    __syncthreads();
  }
  for (int j = n - 1; j >= 0; --j) {                 // L' x = y
    const double* cj = L + (int64_t)n * j;
    double acc = 0.0;
    for (int i = j + 1 + threadIdx.x; i < n; i += blockDim.x) acc += cj[i] * x[i];
    // fixed-order block sum (blockDim is a power of two)
    __shared__ double red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int st = blockDim.x >> 1; st > 0; st >>= 1) {
      if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
      __syncthreads();
    }
    if (threadIdx.x == 0) x[j] = (x[j] - red[0]) / cj[j];
    __syncthreads();
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) Minv[i + (int64_t)n * col] = x[i];
}
void dense_spd_inverse(double* M, double* Minv, int n, AdmmCtl* ctl, hipStream_t s) {
  AO_REQUIRE(n > 0 && n <= kDenseMaxN, "dense system of order %d (limit %d)", n, kDenseMaxN);
  dense_chol_k<<<1, kDenseThreads, 0, s>>>(M, n, ctl);
  AO_KERNEL_CHECK();
  dense_inverse_k<<<n, 256, (size_t)n * sizeof(double), s>>>(M, n, Minv);
  AO_KERNEL_CHECK();
}
// out(k,r) = sum_{k',q} Minv[(k*R+r), (k'*R+q)] * rhs(k',q): the unknowns are the rows of the K x R matrix back to
// back (vec(C'), cmtf_fun_AOADMM.m:717-722); rhs and out are column-major K x R.  One wave per unknown.
__global__ __launch_bounds__(64) void dense_symv_rows_k(const double* Minv, const double* rhs, double* out, int K, int R,
                                                        const AdmmCtl* ctl) {
  CTL_GUARD(ctl);
  const int n = K * R, u = blockIdx.x, k = u / R, r = u % R;
  const double* row = Minv + (int64_t)n * u;         // symmetric: column u = row u, contiguous
  double acc = 0.0;
  for (int v = threadIdx.x; v < n; v += 64) acc += row[v] * rhs[(v / R) + K * (v % R)];
  for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m, 64);
  if (threadIdx.x == 0) out[k + K * r] = acc;
}
void dense_symv_rows(const double* Minv, const double* rhs, double* out, int K, int R, const AdmmCtl* ctl,
                     hipStream_t s) {
  dense_symv_rows_k<<<K * R, 64, 0, s>>>(Minv, rhs, out, K, R, ctl);
  AO_KERNEL_CHECK();
}

// ---------------------------------------------------------------------------
// Symmetric eigendecomposition of a small matrix (n <= 64) by Jacobi rotations, one wave, matrix and eigenvectors in
// LDS: B = V diag(w) V'.  Used by the Sylvester-type primal updates of the transformed couplings
// (cmtf_fun_AOADMM.m:707, :1016) and by the orthonormality prox (project_ortho.m), where B is a Gram matrix.
// Parallel ordering: a sweep is n-1 (n even) tournament rounds of n/2 DISJOINT pairs; lane p computes the rotation of
// its pair, then all column rotations of the round are applied at once, then all row rotations (disjoint rotations
// commute).  A serial cyclic sweep took ~1 us per pair with two barriers each (190 pairs at n = 20); a round here is two
// wave barriers for n/2 pairs.  Fixed schedule, fixed arithmetic order: deterministic.
__global__ __launch_bounds__(64) void sym_eig_small_k(const double* B, int n, double* w, double* V, const AdmmCtl* ctl) {
  CTL_GUARD(ctl);
  extern __shared__ double sh[];          // A[n*n], Q[n*n], cs[2*32], pq[2*32] (as ints)
  __shared__ int done;
  double* A = sh;
  double* Q = sh + n * n;
  double* cs = Q + n * n;                 // c, s of the pair handled by lane p
  int* pr = reinterpret_cast<int*>(cs + 64);   // p, q of that pair (-1: idle)
  const int t = threadIdx.x;
  for (int e = t; e < n * n; e += 64) { A[e] = B[e]; Q[e] = (e % n == e / n) ? 1.0 : 0.0; }
  __syncthreads();
  const int m = (n + 1) & ~1;             // players of the tournament (one dummy when n is odd)
  const int half = m / 2;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0, tot = 0.0;
    for (int e = t; e < n * n; e += 64) { const double v = A[e]; tot += v * v; if (e % n != e / n) off += v * v; }
    for (int o = 32; o > 0; o >>= 1) { off += __shfl_xor(off, o); tot += __shfl_xor(tot, o); }
    if (t == 0) done = !(off > 1e-30 * tot);
    __syncthreads();
    if (done) break;
    for (int r = 0; r < m - 1; ++r) {
      if (t < half) {
        // round r of the circle method: player m-1 is fixed, the others rotate
        int a, b;
        if (t == 0) { a = m - 1; b = r; }
        else { a = (r + t) % (m - 1); b = (r - t + (m - 1)) % (m - 1); }
        int p = a < b ? a : b, q = a < b ? b : a;
        double c = 1.0, sn = 0.0;
        if (q < n) {
          const double apq = A[p + n * q];
          if (apq != 0.0) {
            const double theta = (A[q + n * q] - A[p + n * p]) / (2.0 * apq);
            const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            c = 1.0 / sqrt(tt * tt + 1.0);
            sn = tt * c;
          }
        } else {
          p = -1;                          // the pair with the dummy player sits out
        }
        cs[2 * t] = c; cs[2 * t + 1] = sn;
        pr[2 * t] = p; pr[2 * t + 1] = q;
      }
      __syncthreads();
      // columns p, q of A and of Q for every pair: element (k, pair)
      for (int e = t; e < half * n; e += 64) {
        const int pi = e / n, k = e - pi * n;
        const int p = pr[2 * pi], q = pr[2 * pi + 1];
        const double c = cs[2 * pi], sn = cs[2 * pi + 1];
        if (p >= 0 && sn != 0.0) {
          const double akp = A[k + n * p], akq = A[k + n * q];
          A[k + n * p] = c * akp - sn * akq;
          A[k + n * q] = sn * akp + c * akq;
          const double qkp = Q[k + n * p], qkq = Q[k + n * q];
          Q[k + n * p] = c * qkp - sn * qkq;
          Q[k + n * q] = sn * qkp + c * qkq;
        }
      }
      __syncthreads();
      // rows p, q of A for every pair: element (pair, k)
      for (int e = t; e < half * n; e += 64) {
        const int pi = e / n, k = e - pi * n;
        const int p = pr[2 * pi], q = pr[2 * pi + 1];
        const double c = cs[2 * pi], sn = cs[2 * pi + 1];
        if (p >= 0 && sn != 0.0) {
          const double apk = A[p + n * k], aqk = A[q + n * k];
          A[p + n * k] = c * apk - sn * aqk;
          A[q + n * k] = sn * apk + c * aqk;
        }
      }
      __syncthreads();
    }
  }
  for (int e = t; e < n * n; e += 64) V[e] = Q[e];
  if (t < n) w[t] = A[t + n * t];
}
void sym_eig_small(const double* B, int n, double* w, double* V, hipStream_t s, const AdmmCtl* ctl) {
  AO_REQUIRE(n >= 1 && n <= kMaxRank, "sym_eig_small: n out of range");
  const size_t sh = ((size_t)2 * n * n + 64 + 64) * sizeof(double);
  if (sh > 65536) ensure_dynamic_lds(reinterpret_cast<const void*>(sym_eig_small_k), (int)sh);
  sym_eig_small_k<<<1, 64, sh, s>>>(B, n, w, V, ctl);
  AO_KERNEL_CHECK();
}

// X = AA \ BB for a symmetric positive definite AA (q x q, overwritten by its Cholesky factor) and nrhs
// right-hand sides (BB, q x nrhs, overwritten by X): the Delta update of coupling type 3 (:839).  One
// workgroup, matrix in global memory; q is the row count of Delta (tens to hundreds).
__global__ __launch_bounds__(256) void spd_solve_left_k(double* AA, int64_t q, double* BB, int nrhs, AdmmCtl* ctl) {
  if (ctl && ctl->active == 0) return;
  __shared__ double djj;
  __shared__ int bad;
  const int t = threadIdx.x;
  if (t == 0) bad = 0;
  __syncthreads();
  for (int64_t j = 0; j < q; ++j) {
    if (t == 0) {
      const double d = AA[j + q * j];
      if (!(d > 0.0)) bad = 1;
      djj = sqrt(d);
      AA[j + q * j] = djj;
    }
    __syncthreads();
    if (bad) { if (t == 0 && ctl) ctl->notpd = 1; return; }
    for (int64_t i = j + 1 + t; i < q; i += 256) AA[i + q * j] /= djj;
    __syncthreads();
    for (int64_t k = j + 1; k < q; ++k) {            // column k of the trailing block, rows k..q-1
      const double lkj = AA[k + q * j];
      for (int64_t i = k + t; i < q; i += 256) AA[i + q * k] -= AA[i + q * j] * lkj;
    }
    __syncthreads();
  }
  for (int r = t; r < nrhs; r += 256) {               // one right-hand side per thread
    double* x = BB + q * r;
    for (int64_t i = 0; i < q; ++i) {
      double v = x[i];
      for (int64_t k = 0; k < i; ++k) v -= AA[i + q * k] * x[k];
      x[i] = v / AA[i + q * i];
    }
    for (int64_t i = q - 1; i >= 0; --i) {
      double v = x[i];
      for (int64_t k = i + 1; k < q; ++k) v -= AA[k + q * i] * x[k];
      x[i] = v / AA[i + q * i];
    }
  }
}
void spd_solve_left(double* AA, int64_t q, double* BB, int nrhs, AdmmCtl* ctl, hipStream_t s) {
  spd_solve_left_k<<<1, 256, 0, s>>>(AA, q, BB, nrhs, ctl);
  AO_KERNEL_CHECK();
}

// W(i,j) /= (scale_row * lam[i] + shift + mu[j])   (Sylvester solve in the two eigenbases)
__global__ void sylv_scale_k(double* W, int64_t rows, int R, const double* lam, const double* mu, const double* rho,
                             double lam_mul, double shift_mul, const AdmmCtl* ctl) {
  CTL_GUARD(ctl);
  const double r2 = rho[0] / 2;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < rows * R; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e % rows;
    const int j = (int)(e / rows);
    W[e] = W[e] / (r2 * lam_mul * lam[i] + r2 * shift_mul + mu[j]);
  }
}
void sylv_scale(double* W, int64_t rows, int R, const double* lam, const double* mu, const double* rho, double lam_mul,
                double shift_mul, const AdmmCtl* ctl, hipStream_t s) {
  int64_t nb = cdiv(rows * R, 256);
  if (nb > 1024) nb = 1024;
  sylv_scale_k<<<(unsigned)nb, 256, 0, s>>>(W, rows, R, lam, mu, rho, lam_mul, shift_mul, ctl);
  AO_KERNEL_CHECK();
}

}  // namespace aoadmm
