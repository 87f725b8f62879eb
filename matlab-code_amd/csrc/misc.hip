// Data movement / generation helpers (see misc.h).
#include "misc.h"

#include <map>
#include <mutex>
#include <utility>

namespace aoadmm {

void ensure_dynamic_lds(const void* kernel, int bytes) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, int> granted;
  int dev = 0;
  AO_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(mu);
  int& have = granted[std::make_pair(dev, kernel)];
  if (have >= bytes) return;
  AO_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  have = bytes;
}


// ---------------------------------------------------------------------------
template <typename T>
__global__ void pad_convert_k(T* dst, int64_t pad_rows, const double* src, int64_t rows, int64_t cols,
                              int64_t dst_col0) {
  const int64_t total = pad_rows * cols;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = idx % pad_rows, c = idx / pad_rows;
    const double v = i < rows ? src[i + rows * c] : 0.0;
    dst[i + pad_rows * (dst_col0 + c)] = (T)v;
  }
}
void pad_convert(void* dst, int prec, int64_t pad_rows, const double* src, int64_t rows, int64_t cols,
                 int64_t dst_col0, hipStream_t s) {
  if (rows <= 0 || cols <= 0) return;
  int64_t blocks = cdiv(pad_rows * cols, 256);
  if (blocks > 65536) blocks = 65536;
  if (prec == AOADMM_PREC_F32)
    pad_convert_k<float><<<(unsigned)blocks, 256, 0, s>>>((float*)dst, pad_rows, src, rows, cols, dst_col0);
  else
    pad_convert_k<double><<<(unsigned)blocks, 256, 0, s>>>((double*)dst, pad_rows, src, rows, cols, dst_col0);
  AO_KERNEL_CHECK();
}

template <typename T>
__global__ void transpose_convert_k(T* dst, int64_t pad_c, const double* src, int64_t rows, int64_t cols) {
  // dst(c, i) = src(i, c); dst leading dim pad_c (rows of dst = cols of src, zero padded)
  const int64_t total = pad_c * rows;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = idx % pad_c, i = idx / pad_c;
    dst[idx] = (T)(c < cols ? src[i + rows * c] : 0.0);
  }
}
void transpose_convert(void* dst, int prec, int64_t pad_c, const double* src, int64_t rows, int64_t cols,
                       hipStream_t s) {
  int64_t blocks = cdiv(pad_c * rows, 256);
  if (blocks > 65536) blocks = 65536;
  if (blocks < 1) return;
  if (prec == AOADMM_PREC_F32)
    transpose_convert_k<float><<<(unsigned)blocks, 256, 0, s>>>((float*)dst, pad_c, src, rows, cols);
  else
    transpose_convert_k<double><<<(unsigned)blocks, 256, 0, s>>>((double*)dst, pad_c, src, rows, cols);
  AO_KERNEL_CHECK();
}

// ---------------------------------------------------------------------------
static constexpr int kNormBlocks = 1024;

__device__ __forceinline__ double block_reduce256(double v, double* sh) {
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
    __syncthreads();
  }
  const double r = sh[0];
  __syncthreads();
  return r;
}

__device__ __forceinline__ double block_reduce_n(double v, double* sh) {   // blockDim.x a power of two
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int st = blockDim.x >> 1; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
    __syncthreads();
  }
  const double r = sh[0];
  __syncthreads();
  return r;
}

template <typename T>
__global__ void tensor_sumsq_k(double* ws, const T* X, int64_t n) {
  __shared__ double sh[256];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double v = (double)X[i];
    acc += v * v;
  }
  const double t = block_reduce256(acc, sh);
  if (threadIdx.x == 0) ws[blockIdx.x] = t;
}
__global__ void sum_ws_k(double* slot, const double* ws, int nb, int stride, int nout) {
  if ((int)threadIdx.x < nout) {
    double t = 0.0;
    for (int b = 0; b < nb; ++b) t += ws[(int64_t)b * stride + threadIdx.x];
    slot[threadIdx.x] = t;
  }
}
void tensor_sumsq(double* slot, const void* X, int prec, int64_t n, double* ws, hipStream_t s) {
  if (prec == AOADMM_PREC_F32) tensor_sumsq_k<float><<<kNormBlocks, 256, 0, s>>>(ws, (const float*)X, n);
  else tensor_sumsq_k<double><<<kNormBlocks, 256, 0, s>>>(ws, (const double*)X, n);
  AO_KERNEL_CHECK();
  sum_ws_k<<<1, 64, 0, s>>>(slot, ws, kNormBlocks, 1, 1);
  AO_KERNEL_CHECK();
}

// ---------------------------------------------------------------------------
// counter-based generator: splitmix64 finaliser of (seed, stream, index)
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ double u01(uint64_t seed, uint64_t stream, uint64_t idx) {
  const uint64_t h = mix64(mix64(seed ^ (stream * 0xD1B54A32D192ED03ull)) + idx);
  return (double)(h >> 11) * (1.0 / 9007199254740992.0);    // [0,1)
}
__device__ __forceinline__ double gauss01(uint64_t seed, uint64_t idx) {
  const double u1 = 1.0 - u01(seed, 101, idx);               // (0,1]
  const double u2 = u01(seed, 102, idx);
  return sqrt(-2.0 * log(u1)) * cospi(2.0 * u2);
}

__global__ void fill_uniform_k(double* x, int64_t n, uint64_t seed, uint64_t stream) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    x[i] = u01(seed, stream, (uint64_t)i);
}
void synth_factors(double* A, double* B, double* C, const SynthArgs& a, hipStream_t s) {
  fill_uniform_k<<<256, 256, 0, s>>>(A, a.I_full * a.R, a.seed, 1);
  fill_uniform_k<<<256, 256, 0, s>>>(B, a.J * a.R, a.seed, 2);
  fill_uniform_k<<<256, 256, 0, s>>>(C, a.K * a.R, a.seed, 3);
  AO_KERNEL_CHECK();
}

static constexpr int kSynthBlocks = 4096;
size_t synth_ws_bytes() { return (size_t)kSynthBlocks * 3 * sizeof(double); }

// one (j,k) fibre per block iteration; threads sweep the local rows
template <int PASS, typename T>
__global__ void synth_k(T* X, double* ws, const double* A, const double* B, const double* C, SynthArgs a,
                        double sigma, double inv_norm) {
  __shared__ double bc[kMaxRank];
  __shared__ double sh[256];
  double s_cc = 0, s_nn = 0, s_cn = 0;
  const int64_t Kl = a.K_loc < 0 ? a.K : a.K_loc;
  const int64_t ncol = a.J * Kl;
  for (int64_t lcol = blockIdx.x; lcol < ncol; lcol += gridDim.x) {
    const int64_t j = lcol % a.J, k = lcol / a.J + a.k0;
    const int64_t col = j + a.J * k;                 // global column: the noise stream is indexed by global position
    __syncthreads();
    if ((int)threadIdx.x < a.R) bc[threadIdx.x] = B[j + a.J * threadIdx.x] * C[k + a.K * threadIdx.x];
    __syncthreads();
    for (int64_t i = threadIdx.x; i < a.I_pad; i += blockDim.x) {
      double clean = 0.0, nz = 0.0;
      if (i < a.I_loc) {
        const int64_t ig = i + a.row0;
        for (int r = 0; r < a.R; ++r) clean += A[ig + a.I_full * r] * bc[r];
        nz = gauss01(a.seed, (uint64_t)(ig + a.I_full * col));
      }
      if (PASS == 1) {
        s_cc += clean * clean; s_nn += nz * nz; s_cn += clean * nz;
      } else {
        X[i + a.I_pad * lcol] = (T)((clean + sigma * nz) * inv_norm);
      }
    }
  }
  if (PASS == 1) {
    const double t0 = block_reduce256(s_cc, sh);
    const double t1 = block_reduce256(s_nn, sh);
    const double t2 = block_reduce256(s_cn, sh);
    if (threadIdx.x == 0) {
      ws[(int64_t)blockIdx.x * 3 + 0] = t0;
      ws[(int64_t)blockIdx.x * 3 + 1] = t1;
      ws[(int64_t)blockIdx.x * 3 + 2] = t2;
    }
  }
}
// out3 = { sum clean^2, sum noise^2, sum clean*noise } over the local block
void synth_norms(double* out3, const double* A, const double* B, const double* C, const SynthArgs& a,
                 double* ws, hipStream_t s) {
  synth_k<1, double><<<kSynthBlocks, 256, 0, s>>>(nullptr, ws, A, B, C, a, 0.0, 1.0);
  AO_KERNEL_CHECK();
  sum_ws_k<<<1, 64, 0, s>>>(out3, ws, kSynthBlocks, 3, 3);
  AO_KERNEL_CHECK();
}
void synth_write(void* X, int prec, const double* A, const double* B, const double* C, const SynthArgs& a,
                  double sigma, double inv_norm, hipStream_t s) {
  if (prec == AOADMM_PREC_F32)
    synth_k<2, float><<<kSynthBlocks * 4, 256, 0, s>>>((float*)X, nullptr, A, B, C, a, sigma, inv_norm);
  else
    synth_k<2, double><<<kSynthBlocks * 4, 256, 0, s>>>((double*)X, nullptr, A, B, C, a, sigma, inv_norm);
  AO_KERNEL_CHECK();
}

// ---------------------------------------------------------------------------
// regulariser values: constraints_to_prox.m:49,53,57,61,77,81
__global__ __launch_bounds__(1024) void reg_value_k(double* slot, int type, double eta, const double* X, int64_t rows, int R) {
  __shared__ double sh[1024];
  double acc = 0.0;
  const int64_t n = rows * R;
  if (type == AOADMM_C_L2_REG) {
    double tot = 0.0;
    for (int r = 0; r < R; ++r) {
      double a = 0.0;
      for (int64_t i = threadIdx.x; i < rows; i += blockDim.x) { const double v = X[i + rows * r]; a += v * v; }
      tot += sqrt(block_reduce_n(a, sh));
    }
    if (threadIdx.x == 0) slot[0] = eta * tot;
    return;
  }
  for (int64_t e = threadIdx.x; e < n; e += blockDim.x) {
    const double v = X[e];
    const int64_t i = e % rows;
    switch (type) {
      case AOADMM_C_L1_REG: acc += fabs(v); break;
      case AOADMM_C_L0_REG: acc += (v != 0.0) ? 1.0 : 0.0; break;
      case AOADMM_C_RIDGE: acc += v * v; break;
      case AOADMM_C_TV: if (i + 1 < rows) acc += X[e + 1] - v; break;            // no abs(): quirk of :81
      case AOADMM_C_GL_SMOOTH: if (i + 1 < rows) { const double d = X[e + 1] - v; acc += d * d; } break;
      default: break;
    }
  }
  const double t = block_reduce_n(acc, sh);
  if (threadIdx.x == 0) slot[0] = eta * t;
}
void reg_value(double* slot, int type, double p0, const double* X, int64_t rows, int R, double* ws,
               hipStream_t s) {
  (void)ws;
  reg_value_k<<<1, 1024, 0, s>>>(slot, type, p0, X, rows, R);
  AO_KERNEL_CHECK();
}

// ---------------------------------------------------------------------------
// Resident copies of a 3-way tensor, one per tensor pass, each in the layout its pass streams best (solver.h CpBlock).
// A pass contracts column index c of an "unfolding" whose row index m is the contiguous one.  Rows are cut into blocks
// of kRowBlockElems; a block stores its columns one after the other,
//     element (m, c)  ->  (m / MB) * MB * C  +  c * MB  +  m % MB ,
// so the workgroups that own a block read ONE contiguous range of MB*C elements from front to back instead of one
// 2 KB piece per column, 4*ld bytes apart (2000^3 fp32: 4.80-4.92 ms against 4.96-5.09 ms per pass on the same
// buffers, tools/micro/pass_placement.hip).  Rows beyond the data (up to the next multiple of MB) are zeros.
//   which = 0  (contracts mode 3):  m = i + Ip*j, c = k     -- a re-blocked copy, no transposition
//   which = 1  (contracts mode 1):  m = j + Jp*k, c = i     -- 32 x 32 tiles over (i, j) through LDS per k
//   which = 2  (contracts mode 2):  m = k + Kp*i, c = j     -- 32 x 32 tiles over (i, k) through LDS per j
// Jp / Kp: J / K padded to a multiple of 4 (fp32) or 2 (fp64); the padding rows are written as zeros.
template <typename T>
__global__ __launch_bounds__(256) void block_copy_k(const T* __restrict__ X, T* __restrict__ D, int64_t M, int64_t Mpad,
                                                   int64_t K, int64_t MB) {
  const int64_t k = blockIdx.y;
  constexpr int V = 16 / sizeof(T);
  typedef T VT __attribute__((ext_vector_type(V)));
  for (int64_t m = ((int64_t)blockIdx.x * 256 + threadIdx.x) * V; m < Mpad; m += (int64_t)gridDim.x * 256 * V) {
    VT v;
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] = (T)0;
    if (m < M) v = *reinterpret_cast<const VT*>(X + M * k + m);       // M is a multiple of V (padded first dimension)
    *reinterpret_cast<VT*>(D + (m / MB) * MB * K + k * MB + m % MB) = v;
  }
}
template <typename T, int WHICH>
__global__ __launch_bounds__(256) void block_permute_k(const T* __restrict__ X, T* __restrict__ D, int64_t I, int64_t Ip,
                                                      int64_t J, int64_t K, int64_t Tp, int64_t MB) {
  __shared__ T tile[32][33];
  // WHICH 1: t = j (extent J, padded Tp), u = k;  WHICH 2: t = k (extent K, padded Tp), u = j
  const int64_t u = blockIdx.z;
  const int64_t i0 = (int64_t)blockIdx.x * 32, t0 = (int64_t)blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;            // 32 x 8
  const int64_t nT = WHICH == 1 ? J : K;
  const int64_t st = WHICH == 1 ? Ip : Ip * J, su = WHICH == 1 ? Ip * J : Ip;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t i = i0 + tx, t = t0 + ty + 8 * q;
    tile[ty + 8 * q][tx] = (i < I && t < nT) ? X[i + st * t + su * u] : (T)0;
  }
  __syncthreads();
  const int64_t C = WHICH == 1 ? I : J;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t t = t0 + tx, i = i0 + ty + 8 * q;
    if (t < Tp && i < I) {                              // t >= nT: zeros from the load guard
      const int64_t m = WHICH == 1 ? t + Tp * u : t + Tp * i;
      const int64_t c = WHICH == 1 ? i : u;
      D[(m / MB) * MB * C + c * MB + m % MB] = tile[tx][ty + 8 * q];
    }
  }
}
bool block_layout_copy(const void* X, void* D, int which, int prec, int64_t I, int64_t Ip, int64_t J, int64_t K,
                       int64_t Tp, hipStream_t s) {
  const int64_t MB = kRowBlockElems;
  const size_t es = prec == AOADMM_PREC_F32 ? 4 : 8;
  if (which == 0) {
    if (K > 65535) return false;
    const int64_t M = Ip * J, Mpad = round_up(M, MB);
    const int V = (int)(16 / es);
    int64_t gx = cdiv(Mpad, (int64_t)256 * V);
    if (gx > 65535) gx = 65535;
    const dim3 grid((unsigned)gx, (unsigned)K);
    if (prec == AOADMM_PREC_F32) block_copy_k<float><<<grid, 256, 0, s>>>((const float*)X, (float*)D, M, Mpad, K, MB);
    else block_copy_k<double><<<grid, 256, 0, s>>>((const double*)X, (double*)D, M, Mpad, K, MB);
    AO_KERNEL_CHECK();
    return true;
  }
  const int64_t nU = which == 1 ? K : J, C = which == 1 ? I : J;
  const int64_t M = which == 1 ? Tp * K : Tp * I, Mpad = round_up(M, MB);
  if (nU > 65535 || cdiv(Tp, 32) > 65535) return false;
  if (Mpad > M)                                        // rows of the last block beyond the data
    AO_HIP(hipMemsetAsync((char*)D + (size_t)(Mpad / MB - 1) * MB * C * es, 0, (size_t)MB * C * es, s));
  const dim3 grid((unsigned)cdiv(I, 32), (unsigned)cdiv(Tp, 32), (unsigned)nU);
#define AO_BP(TT, W) block_permute_k<TT, W><<<grid, 256, 0, s>>>((const TT*)X, (TT*)D, I, Ip, J, K, Tp, MB)
  if (prec == AOADMM_PREC_F32) { if (which == 1) AO_BP(float, 1); else AO_BP(float, 2); }
  else { if (which == 1) AO_BP(double, 1); else AO_BP(double, 2); }
#undef AO_BP
  AO_KERNEL_CHECK();
  return true;
}

// ---------------------------------------------------------------------------
// Gram of a mode-n unfolding, Y = X_(n) X_(n)'  (functions/cmtf_nvecs.m:56: `Y = A*A'` for the SVD-based
// initialisation; the leading eigenvectors are taken on the host).  Unfolding row a sits at X + a*sa; the
// reduction runs over t1 < n1 (stride s1) and t2 < n2 (stride s2).  64 x 64 output tiles, 16 reduction
// entries staged through LDS per step, fp64 accumulation; the reduction is split over gridDim.z and the
// partial tiles are added in a fixed order by unfold_gram_sum_k.
template <typename T>
__global__ __launch_bounds__(256) void unfold_gram_k(UnfoldGramArgs a, double* ws) {
  __shared__ double As[16][65], Bs[16][65];
  const T* X = reinterpret_cast<const T*>(a.X);
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int64_t a0 = (int64_t)blockIdx.x * 64, b0 = (int64_t)blockIdx.y * 64;
  const int64_t nch1 = (a.n1 + 15) / 16, nchunks = nch1 * a.n2;
  const int64_t per = (nchunks + gridDim.z - 1) / gridDim.z;
  const int64_t c_lo = (int64_t)blockIdx.z * per;
  int64_t c_hi = c_lo + per;
  if (c_hi > nchunks) c_hi = nchunks;
  double acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
  const bool row_contig = a.sa == 1;                  // which index is contiguous decides the load mapping
  for (int64_t c = c_lo; c < c_hi; ++c) {
    const int64_t t2 = c / nch1, t1_0 = (c - t2 * nch1) * 16;
    const int64_t base = t2 * a.s2;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int aa, tt;
      if (row_contig) { aa = tid & 63; tt = (tid >> 6) + 4 * q; }
      else { tt = tid & 15; aa = (tid >> 4) + 16 * q; }
      const int64_t t1 = t1_0 + tt;
      const bool tv = t1 < a.n1;
      const int64_t off = base + (tv ? t1 : 0) * a.s1;
      const int64_t ra = a0 + aa, rb = b0 + aa;
      As[tt][aa] = (tv && ra < a.n) ? (double)X[ra * a.sa + off] : 0.0;
      Bs[tt][aa] = (tv && rb < a.n) ? (double)X[rb * a.sa + off] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < 16; ++tt) {
      double av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { av[i] = As[tt][ty * 4 + i]; bv[i] = Bs[tt][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += av[i] * bv[j];
    }
    __syncthreads();
  }
  double* W = ws + (int64_t)blockIdx.z * a.n * a.n;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t ra = a0 + ty * 4 + i, rb = b0 + tx * 4 + j;
      if (ra < a.n && rb < a.n) W[ra + a.n * rb] = acc[i][j];
    }
}
__global__ void unfold_gram_sum_k(const double* ws, int64_t nn, int nz, double* out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nn) return;
  double t = 0.0;
  for (int z = 0; z < nz; ++z) t += ws[(int64_t)z * nn + e];
  out[e] = t;
}
static int unfold_gram_splits(const UnfoldGramArgs& a) {
  const int64_t tiles = cdiv(a.n, 64) * cdiv(a.n, 64);
  const int64_t nchunks = cdiv(a.n1, 16) * a.n2;
  int64_t z = cdiv(2048, tiles);
  if (z > nchunks) z = nchunks;
  if (z > 1024) z = 1024;
  while (z > 1 && z * a.n * a.n * 8 > (int64_t)(1ll << 30)) z /= 2;   // partial tiles <= 1 GiB
  return (int)(z < 1 ? 1 : z);
}
size_t unfold_gram_ws_bytes(const UnfoldGramArgs& a) { return (size_t)unfold_gram_splits(a) * a.n * a.n * sizeof(double); }
void unfold_gram(const UnfoldGramArgs& a, int prec, double* ws, double* out, hipStream_t s) {
  AO_REQUIRE(a.n > 0 && a.n1 > 0 && a.n2 > 0, "unfold_gram: bad sizes");
  AO_REQUIRE(cdiv(a.n, 64) <= 65535, "unfold_gram: mode too long");
  const int nz = unfold_gram_splits(a);
  const dim3 grid((unsigned)cdiv(a.n, 64), (unsigned)cdiv(a.n, 64), (unsigned)nz);
  if (prec == AOADMM_PREC_F32) unfold_gram_k<float><<<grid, 256, 0, s>>>(a, ws);
  else unfold_gram_k<double><<<grid, 256, 0, s>>>(a, ws);
  AO_KERNEL_CHECK();
  unfold_gram_sum_k<<<(unsigned)cdiv(a.n * a.n, 256), 256, 0, s>>>(ws, a.n * a.n, nz, out);
  AO_KERNEL_CHECK();
}

// out(a + A*b, r) = Fa(a, r) * Fb(b, r): the Khatri-Rao factor of two merged trailing modes (column-major, the same
// order in which the tensor stores them), used to present an N-way block to the 3-way EM kernel
__global__ void kr_merge_k(double* out, const double* Fa, int64_t lda, int64_t A, const double* Fb, int64_t ldb, int64_t B,
                           int R) {
  const int64_t n = A * B * R;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t ab = e % (A * B);
    const int r = (int)(e / (A * B));
    out[e] = Fa[ab % A + lda * r] * Fb[ab / A + ldb * r];
  }
}
void kr_merge(double* out, const double* Fa, int64_t lda, int64_t A, const double* Fb, int64_t ldb, int64_t B, int R,
              hipStream_t s) {
  int64_t nb = cdiv(A * B * R, 256);
  if (nb > 4096) nb = 4096;
  kr_merge_k<<<(unsigned)nb, 256, 0, s>>>(out, Fa, lda, A, Fb, ldb, B, R);
  AO_KERNEL_CHECK();
}

}  // namespace aoadmm
