// EM missing-data pass (see em.h).  Reference: functions/cmtf_fun_AOADMM.m:408-441 (imputation and
// f_rel_missing), :1224-1226 / :1249-1252 (objective over the observed entries), cmtf_AOADMM.m:133-148
// (Znorm_const of the masked data).
//
// CP block: one workgroup owns a strip of 256*VEC first-mode rows at one third-mode index k and walks the
// second mode.  A thread keeps its VEC rows of A, pre-multiplied by C(k,:), in registers; rows of B are
// staged through LDS 64 at a time and broadcast.  The model value is VEC*R FMAs per vector of entries, the
// tensor is read once (16-byte loads) and only vectors that contain a missing entry are written back:
// the pass is bound by HBM like the contractions.
#include "em.h"

namespace aoadmm {

template <typename T, int VEC> struct EmVec;
template <> struct EmVec<float, 4> { typedef float type __attribute__((ext_vector_type(4))); typedef uint32_t mtype; };
template <> struct EmVec<double, 2> { typedef double type __attribute__((ext_vector_type(2))); typedef uint16_t mtype; };
template <> struct EmVec<float, 1> { typedef float type; typedef uint8_t mtype; };
template <> struct EmVec<double, 1> { typedef double type; typedef uint8_t mtype; };

static constexpr int kEmThreads = 256;
static constexpr int kEmJTile = 64;

__device__ __forceinline__ double em_block_sum(double v, double* sh4) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = v;
  __syncthreads();
  const double r = (sh4[0] + sh4[1]) + (sh4[2] + sh4[3]);
  __syncthreads();
  return r;
}

template <typename T, int VEC, int RMAX>
__global__ __launch_bounds__(kEmThreads) void em_cp_k(EmCpArgs a, double* ws) {
  typedef typename EmVec<T, VEC>::type XV;
  typedef typename EmVec<T, VEC>::mtype MV;
  __shared__ T Bsh[kEmJTile][RMAX];
  __shared__ double sh4[4];
  const int R = a.R;
  const int t = threadIdx.x;
  const int64_t k = blockIdx.y;
  const int64_t i0 = ((int64_t)blockIdx.x * kEmThreads + t) * VEC;       // first row of this thread
  // rows of A scaled by C(k,:): m(i,j,k) = sum_r (A(i,r) C(k,r)) B(j,r)
  T areg[VEC][RMAX];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    const int64_t i = i0 + v < a.I ? i0 + v : a.I - 1;                  // clamped: padding rows are skipped below
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
      const int rr = r < R ? r : 0;
      const double c = a.C ? a.C[k + a.ldC * rr] : 1.0;
      areg[v][r] = r < R ? (T)(a.A[i + a.ldA * rr] * c) : (T)0;
    }
  }
  T* X = reinterpret_cast<T*>(a.X) + a.Ipad * a.J * k;
  const uint8_t* M = a.mask + a.Ipad * a.J * k;
  const bool in_range = i0 < a.Ipad;                                     // Ipad is a multiple of VEC
  double num = 0, den = 0, ores = 0, ox2 = 0;
  for (int64_t j0 = 0; j0 < a.J; j0 += kEmJTile) {
    __syncthreads();
    for (int e = t; e < kEmJTile * RMAX; e += kEmThreads) {
      const int jj = e / RMAX, r = e - jj * RMAX;
      Bsh[jj][r] = (r < R && j0 + jj < a.J) ? (T)a.B[j0 + jj + a.ldB * r] : (T)0;
    }
    __syncthreads();
    const int nj = (int)((a.J - j0 < kEmJTile) ? (a.J - j0) : kEmJTile);
    if (!in_range) continue;
    // the loads run PD columns ahead of the arithmetic (register ring): with one 16-byte load and its
    // mask word per iteration and ~100 FMAs behind them, nothing else hides the memory latency at 4 waves per SIMD
    constexpr int PD = 4;                              // columns in flight
    XV xq[PD]; MV mq[PD];
#pragma unroll
    for (int p = 0; p < PD; ++p) {
      const int64_t op = i0 + a.Ipad * (j0 + (p < nj ? p : nj - 1));
      xq[p] = *reinterpret_cast<const XV*>(X + op);
      mq[p] = *reinterpret_cast<const MV*>(M + op);
    }
    for (int jj = 0; jj < nj; ++jj) {
      const int64_t o = i0 + a.Ipad * (j0 + jj);
      XV xv = xq[0];
      const MV mv = mq[0];
#pragma unroll
      for (int p = 0; p + 1 < PD; ++p) { xq[p] = xq[p + 1]; mq[p] = mq[p + 1]; }
      {
        const int jn = jj + PD < nj ? jj + PD : nj - 1;                  // clamped: the last loads are discarded
        const int64_t on = i0 + a.Ipad * (j0 + jn);
        xq[PD - 1] = *reinterpret_cast<const XV*>(X + on);
        mq[PD - 1] = *reinterpret_cast<const MV*>(M + on);
      }
      T m[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) m[v] = (T)0;
#pragma unroll
      for (int r = 0; r < RMAX; ++r) {
        const T b = Bsh[jj][r];
#pragma unroll
        for (int v = 0; v < VEC; ++v) m[v] += areg[v][r] * b;
      }
      bool any_missing = false;
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        if (i0 + v < a.I) {
          T x;
          if constexpr (VEC == 1) x = xv; else x = xv[v];
          const bool observed = ((mv >> (8 * v)) & 0xff) != 0;
          const double xd = (double)x, md = (double)m[v];
          if (observed) {
            ores += (xd - md) * (xd - md);
            ox2 += xd * xd;
          } else {
            num += (md - xd) * (md - xd);
            den += xd * xd;
            if constexpr (VEC == 1) xv = m[v]; else xv[v] = m[v];
            any_missing = true;
          }
        }
      }
      if (a.update && any_missing) *reinterpret_cast<XV*>(X + o) = xv;
    }
  }
  num = em_block_sum(num, sh4); den = em_block_sum(den, sh4);
  ores = em_block_sum(ores, sh4); ox2 = em_block_sum(ox2, sh4);
  if (t == 0) {
    double* w = ws + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 4;
    w[0] = num; w[1] = den; w[2] = ores; w[3] = ox2;
  }
}

// out4[q] = sum_b ws[b][q], fixed order (256 threads, strided partial sums then a tree)
__global__ __launch_bounds__(256) void em_sum4_k(const double* ws, int64_t nb, double* out4) {
  __shared__ double sh4[4];
  double s[4] = {0, 0, 0, 0};
  for (int64_t b = threadIdx.x; b < nb; b += 256) {
#pragma unroll
    for (int q = 0; q < 4; ++q) s[q] += ws[b * 4 + q];
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const double tot = em_block_sum(s[q], sh4);
    if (threadIdx.x == 0) out4[q] = tot;
  }
}

size_t em_cp_ws_bytes(int64_t Ipad, int64_t K) {
  return (size_t)(cdiv(Ipad, kEmThreads) * K) * 4 * sizeof(double);   // VEC >= 1: never more strips than this
}

template <typename T, int VEC>
static void em_cp_launch(const EmCpArgs& a, double* ws, dim3 grid, hipStream_t s) {
  if (a.R <= 8) em_cp_k<T, VEC, 8><<<grid, kEmThreads, 0, s>>>(a, ws);
  else if (a.R <= 16) em_cp_k<T, VEC, 16><<<grid, kEmThreads, 0, s>>>(a, ws);
  else if (a.R <= 24) em_cp_k<T, VEC, 24><<<grid, kEmThreads, 0, s>>>(a, ws);
  else em_cp_k<T, VEC, 32><<<grid, kEmThreads, 0, s>>>(a, ws);
}

void em_cp_pass(const EmCpArgs& a, int prec, double* ws, double* out4, hipStream_t s) {
  AO_REQUIRE(a.R >= 1 && a.R <= kMaxRank && a.I > 0 && a.J > 0 && a.K > 0, "em_cp_pass: bad sizes");
  AO_REQUIRE(a.K <= 65535, "em_cp_pass: third mode too long for one launch");
  const bool wide = a.R <= 32;                       // VEC rows of A in registers; beyond 32 columns one row per thread
  const int vec = !wide ? 1 : (prec == AOADMM_PREC_F32 ? 4 : 2);
  const dim3 grid((unsigned)cdiv(a.Ipad, (int64_t)kEmThreads * vec), (unsigned)a.K);
  if (prec == AOADMM_PREC_F32) {
    if (wide) em_cp_launch<float, 4>(a, ws, grid, s);
    else em_cp_k<float, 1, 64><<<grid, kEmThreads, 0, s>>>(a, ws);
  } else {
    if (wide) em_cp_launch<double, 2>(a, ws, grid, s);
    else em_cp_k<double, 1, 64><<<grid, kEmThreads, 0, s>>>(a, ws);
  }
  AO_KERNEL_CHECK();
  em_sum4_k<<<1, 256, 0, s>>>(ws, (int64_t)grid.x * grid.y, out4);
  AO_KERNEL_CHECK();
}

// PARAFAC2: one workgroup per slab, M_k = A diag(C(k,:)) B_k'  (:423-433, :1249-1252)
__global__ __launch_bounds__(256) void em_par2_k(EmPar2Args a, double* ws) {
  __shared__ double sh4[4];
  const int k = blockIdx.x;
  const int64_t o = a.off[k];
  const int Jk = (int)(a.off[k + 1] - o);
  double* X = a.X + (int64_t)a.I * o;
  const uint8_t* M = a.mask + (int64_t)a.I * o;
  const double* Bk = a.B + o * a.R;
  double num = 0, den = 0, ores = 0, ox2 = 0;
  for (int e = threadIdx.x; e < a.I * Jk; e += 256) {
    const int j = e / a.I, i = e - j * a.I;
    double m = 0.0;
    for (int r = 0; r < a.R; ++r) m += a.A[i + (int64_t)a.I * r] * a.C[k + (int64_t)a.K * r] * Bk[j + (int64_t)Jk * r];
    const double x = X[e];
    if (M[e]) {
      ores += (x - m) * (x - m);
      ox2 += x * x;
    } else {
      num += (m - x) * (m - x);
      den += x * x;
      if (a.update) X[e] = m;
    }
  }
  num = em_block_sum(num, sh4); den = em_block_sum(den, sh4);
  ores = em_block_sum(ores, sh4); ox2 = em_block_sum(ox2, sh4);
  if (threadIdx.x == 0) {
    double* w = ws + (int64_t)k * 4;
    w[0] = num; w[1] = den; w[2] = ores; w[3] = ox2;
  }
}

void em_par2_pass(const EmPar2Args& a, double* ws, double* out4, hipStream_t s) {
  em_par2_k<<<a.K, 256, 0, s>>>(a, ws);
  AO_KERNEL_CHECK();
  em_sum4_k<<<1, 256, 0, s>>>(ws, a.K, out4);
  AO_KERNEL_CHECK();
}

}  // namespace aoadmm
