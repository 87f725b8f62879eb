// EM missing-data pass (see em.h).  Reference: functions/cmtf_fun_AOADMM.m:408-441 (imputation and
// f_rel_missing), :1224-1226 / :1249-1252 (objective over the observed entries), cmtf_AOADMM.m:133-148
// (Znorm_const of the masked data).
//
// CP block: one workgroup owns a strip of 256*VEC first-mode rows at one third-mode index k and walks the
// second mode.  A thread keeps its VEC rows of A, pre-multiplied by C(k,:), in registers; rows of B are
// staged through LDS 64 at a time and broadcast.  The model value is VEC*R FMAs per vector of entries, the
// tensor is read once (16-byte loads), the mask is ONE BIT per entry (packed on the device when Z.miss arrives: at
// 1000^3 the byte-per-entry mask was 1 GB of the pass's 9 GB), and only 128-byte lines that contain a missing entry are
// written back: the pass is bound by HBM like the contractions.
#include "em.h"

#include <type_traits>

namespace aoadmm {

template <typename T, int VEC> struct EmVec;
template <> struct EmVec<float, 4> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct EmVec<double, 2> { typedef double type __attribute__((ext_vector_type(2))); };

// Z.miss as one bit per entry (bit e of the padded layout in byte e >> 3, position e & 7), 1 = observed.  A thread's VEC
// entries start at a multiple of VEC, so they sit in one byte.
__global__ void em_mask_pack_k(const uint8_t* __restrict__ bytes, uint8_t* __restrict__ bits, int64_t n) {
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;        // output byte
  const int64_t e0 = o * 8;
  if (e0 >= n) return;
  unsigned v = 0;
  if (e0 + 8 <= n) {
    const uint64_t w = *reinterpret_cast<const uint64_t*>(bytes + e0);     // cudaMalloc-aligned base, e0 multiple of 8
#pragma unroll
    for (int q = 0; q < 8; ++q) v |= (((w >> (8 * q)) & 0xff) != 0 ? 1u : 0u) << q;
  } else {
    for (int q = 0; q < 8; ++q) v |= ((e0 + q < n ? bytes[e0 + q] : 1) != 0 ? 1u : 0u) << q;   // past the end: observed
  }
  bits[o] = (uint8_t)v;
}
void em_mask_pack(const uint8_t* bytes, uint8_t* bits, int64_t n, hipStream_t s) {
  const int64_t nb = cdiv(n, (int64_t)8);
  em_mask_pack_k<<<(unsigned)cdiv(nb, (int64_t)256), 256, 0, s>>>(bytes, bits, n);
  AO_KERNEL_CHECK();
}

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

// acc + a * b[h] on both rows of a register pair: v_pk_fma_f32 reading one half of the pair `b` for both lanes
// (op_sel), so a broadcast LDS value needs no copy into a second register -- the compiler's form of `a * splat(b)` spent
// a v_mov and a register per rank column on it.
typedef float em_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ em_f2 pk_fma_lo(em_f2 a, em_f2 b, em_f2 acc) {
  em_f2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(acc));
  return d;
}
__device__ __forceinline__ em_f2 pk_fma_hi(em_f2 a, em_f2 b, em_f2 acc) {
  em_f2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(acc));
  return d;
}

// m + sum_r areg[r] * brow[r]   (brow: one row of the LDS tile, the same for every lane)
template <typename T, int VEC, int RMAX, typename XV>
__device__ __forceinline__ XV em_row_dot(const XV* areg, const T* brow, XV m) {
  if constexpr (std::is_same<T, float>::value && VEC == 4) {
    em_f2 m01 = m.xy, m23 = m.zw;
#pragma unroll
    for (int r = 0; r < RMAX; r += 2) {
      const em_f2 b2 = *reinterpret_cast<const em_f2*>(brow + r);
      m01 = pk_fma_lo(areg[r].xy, b2, m01); m23 = pk_fma_lo(areg[r].zw, b2, m23);
      m01 = pk_fma_hi(areg[r + 1].xy, b2, m01); m23 = pk_fma_hi(areg[r + 1].zw, b2, m23);
    }
    XV o;
    o.xy = m01; o.zw = m23;
    return o;
  } else {
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
      const T b = brow[r];
      XV bb;
#pragma unroll
      for (int v = 0; v < VEC; ++v) bb[v] = b;
      m = __builtin_elementwise_fma(areg[r], bb, m);
    }
    return m;
  }
}

// tacc[r] += x * brow[r]
template <typename T, int VEC, int RMAX, typename XV>
__device__ __forceinline__ void em_row_axpy(XV* tacc, const T* brow, XV x) {
  if constexpr (std::is_same<T, float>::value && VEC == 4) {
    const em_f2 x01 = x.xy, x23 = x.zw;
#pragma unroll
    for (int r = 0; r < RMAX; r += 2) {
      const em_f2 b2 = *reinterpret_cast<const em_f2*>(brow + r);
      tacc[r].xy = pk_fma_lo(x01, b2, tacc[r].xy); tacc[r].zw = pk_fma_lo(x23, b2, tacc[r].zw);
      tacc[r + 1].xy = pk_fma_hi(x01, b2, tacc[r + 1].xy); tacc[r + 1].zw = pk_fma_hi(x23, b2, tacc[r + 1].zw);
    }
  } else {
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
      const T b = brow[r];
      XV bb;
#pragma unroll
      for (int v = 0; v < VEC; ++v) bb[v] = b;
      tacc[r] = __builtin_elementwise_fma(x, bb, tacc[r]);
    }
  }
}

// Strip kernel for R <= 32.  Everything in the column loop is branch-free vector arithmetic in the tensor's
// precision (v_pk_*_f32 for fp32): the first version spent ~330 vector instructions per 16-byte vector of entries
// (a register ring shifted with moves, one branch per entry, fp64 statistics) and was bound by instruction issue
// at 4.3 TB/s, not by memory.
//   * VEC rows of A (times C(k,:)) per thread, held as R vectors areg[r]; B rows broadcast from LDS (ds_read_b128)
//   * loads run PD columns ahead in a statically indexed register ring (the column loop is unrolled by PD)
//   * the four statistics are accumulated per 64-column tile in the tensor's precision (64 terms per accumulator
//     lane) and join the fp64 sums once per tile
//   * padding rows (i >= I; the tensor holds zeros there) get a zero row of A and are forced 'observed': they add
//     exact zeros and are written back unchanged
//   * write-back by 128-byte line: a line leaves (streaming stores of all its vectors) when any of its entries is
//     missing -- whole cache lines, nothing read-modified-written in L2 (storing only the VECTORS with a missing entry
//     measured slower than storing everything); at 20 % missing every line qualifies, at 1 % about a quarter
//   * the strip can walk the second mode at a fixed third-mode index (walk = 1) or the third mode at a fixed
//     second-mode index (walk = 2, steps of Ipad*J elements); with FUSE it also accumulates, from the values it
//     writes back, the partial contraction of the walked mode  T(i, f, r) = sum_w x_new(i, w, f) W(w, r)  -- the
//     tensor pass the next outer iteration would start with (contract.h ContractPlan: T is [chunk][i + Ipad*f][R]
//     in the tensor's precision, one chunk per piece of the walk)
template <typename T, int VEC, int RMAX, bool FUSE>
__global__ __launch_bounds__(kEmThreads, FUSE ? 2 : 1) void em_cp_vec_k(EmCpArgs a, int jchunks, int64_t jlen, double* ws) {
  typedef typename EmVec<T, VEC>::type XV;
  typedef uint8_t MV;                                                    // the byte that holds this thread's VEC mask bits
  constexpr unsigned kFull = (1u << VEC) - 1u;
  __shared__ __attribute__((aligned(16))) T Bsh[kEmJTile][RMAX];
  __shared__ double sh4[4];
  const int R = a.R;
  const int t = threadIdx.x;
  const bool wj = a.walk != 2;                                           // walking the second mode
  const int64_t f = blockIdx.y;                                          // the fixed index (k, or j when walking k)
  const int64_t NW = wj ? a.J : a.K;
  const double* Ff = wj ? a.C : a.B;                                     // factor of the fixed mode: scales the rows of A
  const int64_t ldf = wj ? a.ldC : a.ldB;
  const double* Fw = wj ? a.B : a.C;                                     // factor of the walked mode: rows through LDS
  const int64_t ldw = wj ? a.ldB : a.ldC;
  const int64_t step = wj ? a.Ipad : a.Ipad * a.J;                       // elements between two walk positions
  const int chunk = blockIdx.x % jchunks;                                // the walk is cut into jchunks pieces
  const int64_t i0 = ((int64_t)(blockIdx.x / jchunks) * kEmThreads + t) * VEC;   // first row of this thread
  const int64_t jbeg = chunk * jlen, jend = jbeg + jlen < NW ? jbeg + jlen : NW;
  XV areg[RMAX];                                                         // areg[r][v] = A(i0+v, r) * Ff(f, r)
  unsigned padmask = 0;
#pragma unroll
  for (int v = 0; v < VEC; ++v)
    if (i0 + v >= a.I) padmask |= 1u << v;
#pragma unroll
  for (int r = 0; r < RMAX; ++r) {
    const int rr = r < R ? r : 0;
    const double c = Ff ? Ff[f + ldf * rr] : 1.0;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const bool valid = i0 + v < a.I;
      const int64_t i = valid ? i0 + v : a.I - 1;
      areg[r][v] = (r < R && valid) ? (T)(a.A[i + a.ldA * rr] * c) : (T)0;
    }
    // the fp64 loads of this prologue must not all be in flight at once: 2 * VEC * RMAX registers on top of tacc
    if constexpr (FUSE) { if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0); }
  }
  const int64_t base = (wj ? a.Ipad * a.J : a.Ipad) * f;
  T* X = reinterpret_cast<T*>(a.X) + base;
  const uint8_t* M = a.mask;                                             // bits: entry e at byte e >> 3, position e & 7
  const int64_t mbase = base + i0;                                       // this thread's first entry at walk position 0
  const unsigned sh0 = (unsigned)(mbase & 7), shs = (unsigned)(step & 7);   // position inside the byte at walk position w: (sh0 + shs*w) & 7
  const int lane = t & 63;
  const bool in_range = i0 < a.Ipad;                                     // Ipad is a multiple of VEC
  const XV zero = {};
  XV tacc[FUSE ? RMAX : 1];
#pragma unroll
  for (int r = 0; r < (FUSE ? RMAX : 1); ++r) tacc[r] = zero;
  double num = 0, den = 0, ores = 0, ox2 = 0;
  for (int64_t j0 = jbeg; j0 < jend; j0 += kEmJTile) {
    __syncthreads();
    for (int e = t; e < kEmJTile * RMAX; e += kEmThreads) {
      const int jj = e / RMAX, r = e - jj * RMAX;
      Bsh[jj][r] = (r < R && j0 + jj < jend) ? (T)Fw[j0 + jj + ldw * r] : (T)0;
    }
    __syncthreads();
    const int nj = (int)((jend - j0 < kEmJTile) ? (jend - j0) : kEmJTile);
    if (!in_range) continue;
    constexpr int PD = 4;                              // columns in flight
    XV xq[PD]; MV mq[PD];
#pragma unroll
    for (int p = 0; p < PD; ++p) {
      const int64_t wp = j0 + (p < nj ? p : nj - 1);
      const int64_t op = i0 + step * wp;
      xq[p] = __builtin_nontemporal_load(reinterpret_cast<const XV*>(X + op));
      mq[p] = M[(mbase + step * wp) >> 3];
    }
    XV s_ores = zero, s_ox2 = zero, s_num = zero, s_den = zero;
    for (int jj = 0; jj < nj; jj += PD) {
#pragma unroll
      for (int p = 0; p < PD; ++p) {
        const int jc = jj + p;                                           // wave-uniform
        const XV xv = xq[p];
        const unsigned mv = (((unsigned)mq[p] >> ((sh0 + shs * (unsigned)(j0 + jc)) & 7u)) & kFull) | padmask;
        {
          const int jn = jc + PD < nj ? jc + PD : nj - 1;                // clamped: the last loads are discarded
          const int64_t on = i0 + step * (j0 + jn);
          xq[p] = __builtin_nontemporal_load(reinterpret_cast<const XV*>(X + on));
          mq[p] = M[(mbase + step * (j0 + jn)) >> 3];
        }
        if (jc < nj) {
          const XV m = em_row_dot<T, VEC, RMAX, XV>(areg, &Bsh[jc][0], zero);
          const XV d = xv - m;
          XV od, ox, xn;                                                 // residual / value where observed, else 0
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            const bool observed = ((mv >> v) & 1u) != 0;
            od[v] = observed ? d[v] : (T)0;
            ox[v] = observed ? xv[v] : (T)0;
            xn[v] = observed ? xv[v] : m[v];
          }
          const XV md = d - od, mx = xv - ox;                            // the same where missing (exact)
          s_ores = __builtin_elementwise_fma(od, od, s_ores);
          s_ox2 = __builtin_elementwise_fma(ox, ox, s_ox2);
          s_num = __builtin_elementwise_fma(md, md, s_num);
          s_den = __builtin_elementwise_fma(mx, mx, s_den);
          if (a.update) {
            // Only 128-byte lines with a missing entry go back (at 1-5 % missing most lines hold none; whole lines, so
            // nothing is read-modified-written in L2 -- storing single 16-byte vectors was slower than storing everything).
            // The wave's 1024 bytes start 16-byte aligned: lane l shares its line with the lanes of the same (l + s) >> 3,
            // s = the first lane's 16-byte slot inside its line; lines cut by the wave's ends go back if this wave's part
            // of them has a missing entry.
            XV* dst = reinterpret_cast<XV*>(X + i0 + step * (j0 + jc));
            const unsigned long long bal = __ballot(mv != kFull);
            const unsigned s16 = (unsigned)(((reinterpret_cast<uintptr_t>(dst) >> 4) - (unsigned)lane) & 7u);
            const unsigned g = ((unsigned)lane + s16) >> 3;
            const unsigned grp = g < 8 ? (unsigned)((bal << s16) >> (8 * g)) & 0xffu : (unsigned)(bal >> (64 - s16)) & 0xffu;
            if (grp) __builtin_nontemporal_store(xn, dst);
          }
          if constexpr (FUSE) em_row_axpy<T, VEC, RMAX, XV>(tacc, &Bsh[jc][0], xn);
        }
        // 2 * RMAX vectors are live across the loop: keep the scheduler from interleaving the columns of one
        // unrolled group (it held four columns' temporaries at once and spilled at R = 20)
        if constexpr (FUSE) __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      ores += (double)s_ores[v]; ox2 += (double)s_ox2[v]; num += (double)s_num[v]; den += (double)s_den[v];
    }
  }
  if constexpr (FUSE) {
    if (in_range) {
      // rows i0 .. i0+VEC-1 of T at fixed index f: VEC*R contiguous values per thread, consecutive threads adjacent
      T* Tt = reinterpret_cast<T*>(a.T) + (int64_t)chunk * a.t_chunk_stride + (i0 + a.Ipad * f) * R;
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        if (R == RMAX && (RMAX * sizeof(T)) % 16 == 0) {
          constexpr int W = 16 / sizeof(T);
          typedef T TV __attribute__((ext_vector_type(W)));
#pragma unroll
          for (int r = 0; r < RMAX; r += W) {
            TV o;
#pragma unroll
            for (int q = 0; q < W; ++q) o[q] = tacc[r + q][v];
            *reinterpret_cast<TV*>(Tt + (int64_t)v * R + r) = o;
          }
        } else {
#pragma unroll
          for (int r = 0; r < RMAX; ++r)
            if (r < R) Tt[(int64_t)v * R + r] = tacc[r][v];
        }
      }
    }
  }
  num = em_block_sum(num, sh4); den = em_block_sum(den, sh4);
  ores = em_block_sum(ores, sh4); ox2 = em_block_sum(ox2, sh4);
  if (t == 0) {
    double* w = ws + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 4;
    w[0] = num; w[1] = den; w[2] = ores; w[3] = ox2;
  }
}

// One row per thread, for R > 32 (the rows of A no longer fit in registers VEC at a time).
template <typename T, int RMAX>
__global__ __launch_bounds__(kEmThreads) void em_cp_k(EmCpArgs a, int jchunks, int64_t jlen, double* ws) {
  __shared__ T Bsh[kEmJTile][RMAX];
  __shared__ double sh4[4];
  const int R = a.R;
  const int t = threadIdx.x;
  const int64_t k = blockIdx.y;
  const int chunk = blockIdx.x % jchunks;
  const int64_t i0 = (int64_t)(blockIdx.x / jchunks) * kEmThreads + t;
  const int64_t jbeg = chunk * jlen, jend = jbeg + jlen < a.J ? jbeg + jlen : a.J;
  T areg[RMAX];
  {
    const int64_t i = i0 < a.I ? i0 : a.I - 1;                          // clamped: padding rows are skipped below
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
      const int rr = r < R ? r : 0;
      const double c = a.C ? a.C[k + a.ldC * rr] : 1.0;
      areg[r] = r < R ? (T)(a.A[i + a.ldA * rr] * c) : (T)0;
    }
  }
  T* X = reinterpret_cast<T*>(a.X) + a.Ipad * a.J * k;
  const uint8_t* M = a.mask;                                             // bits (see em_mask_pack_k)
  const int64_t mb = a.Ipad * a.J * k;
  double num = 0, den = 0, ores = 0, ox2 = 0;
  for (int64_t j0 = jbeg; j0 < jend; j0 += kEmJTile) {
    __syncthreads();
    for (int e = t; e < kEmJTile * RMAX; e += kEmThreads) {
      const int jj = e / RMAX, r = e - jj * RMAX;
      Bsh[jj][r] = (r < R && j0 + jj < jend) ? (T)a.B[j0 + jj + a.ldB * r] : (T)0;
    }
    __syncthreads();
    const int nj = (int)((jend - j0 < kEmJTile) ? (jend - j0) : kEmJTile);
    if (i0 >= a.I) continue;
    for (int jj = 0; jj < nj; ++jj) {
      const int64_t o = i0 + a.Ipad * (j0 + jj);
      const T x = X[o];
      T m = (T)0;
#pragma unroll
      for (int r = 0; r < RMAX; ++r) m += areg[r] * Bsh[jj][r];
      const T d = x - m;
      if ((M[(mb + o) >> 3] >> ((mb + o) & 7)) & 1) {
        ores += (double)(d * d); ox2 += (double)(x * x);
      } else {
        num += (double)(d * d); den += (double)(x * x);
        if (a.update) X[o] = m;
      }
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

// The walk is cut into pieces of whole 64-column tiles so that the launch has several thousand workgroups:
// with one workgroup per (strip, k) a matrix block would be one workgroup per strip.  A fused pass writes one chunk of T per
// piece, so it only cuts when the launch would not fill the chip, and never runs a piece past the contraction kernels'
// bound of 2048 terms accumulated in fp32.
static constexpr int64_t kEmTargetBlocks = 6144;
static constexpr int64_t kEmFuseBlocks = 768;
static constexpr int64_t kEmFuseMaxRun = 2048;
static void em_chunking(int64_t strips, int64_t NW, int64_t NF, bool fuse, int* jchunks, int64_t* jlen) {
  const int64_t tiles = cdiv(NW, (int64_t)kEmJTile);
  int64_t want = cdiv(fuse ? kEmFuseBlocks : kEmTargetBlocks, strips * NF);
  if (fuse && want < cdiv(NW, kEmFuseMaxRun)) want = cdiv(NW, kEmFuseMaxRun);
  if (want > tiles) want = tiles;
  if (want < 1) want = 1;
  *jlen = cdiv(tiles, want) * kEmJTile;
  *jchunks = (int)cdiv(NW, *jlen);
}

static bool em_wide(const EmCpArgs& a) { return a.R <= 32; }
static int em_vec(const EmCpArgs& a, int prec) { return !em_wide(a) ? 1 : (prec == AOADMM_PREC_F32 ? 4 : 2); }

bool em_cp_can_fuse(const EmCpArgs& a, int prec) {
  (void)prec;
  return a.R <= kEmFuseMaxRank && a.C != nullptr;    // 2 * R vectors of registers per thread (rows of A and of T)
}

int em_cp_fused_chunks(const EmCpArgs& a, int prec) {
  const bool wj = a.walk != 2;
  int jchunks; int64_t jlen;
  em_chunking(cdiv(a.Ipad, (int64_t)kEmThreads * em_vec(a, prec)), wj ? a.J : a.K, wj ? a.K : a.J, true, &jchunks, &jlen);
  return jchunks;
}

size_t em_cp_ws_bytes(int64_t Ipad, int64_t J, int64_t K) {
  // strips <= cdiv(Ipad, 256) (VEC >= 1), the fixed index runs over J or K, pieces <= cdiv(kEmTargetBlocks, strips * fixed):
  // never more workgroups than this
  return (size_t)(cdiv(Ipad, (int64_t)kEmThreads) * (J > K ? J : K) + kEmTargetBlocks) * 4 * sizeof(double);
}

template <typename T, int VEC, bool FUSE>
static void em_cp_launch(const EmCpArgs& a, int jchunks, int64_t jlen, double* ws, dim3 grid, hipStream_t s) {
  switch ((a.R + 3) / 4) {                           // rank rounded up to a multiple of 4 (one ds_read_b128 of fp32)
    case 1: em_cp_vec_k<T, VEC, 4, FUSE><<<grid, kEmThreads, 0, s>>>(a, jchunks, jlen, ws); break;
    case 2: em_cp_vec_k<T, VEC, 8, FUSE><<<grid, kEmThreads, 0, s>>>(a, jchunks, jlen, ws); break;
    case 3: em_cp_vec_k<T, VEC, 12, FUSE><<<grid, kEmThreads, 0, s>>>(a, jchunks, jlen, ws); break;
    case 4: em_cp_vec_k<T, VEC, 16, FUSE><<<grid, kEmThreads, 0, s>>>(a, jchunks, jlen, ws); break;
    case 5: em_cp_vec_k<T, VEC, 20, FUSE><<<grid, kEmThreads, 0, s>>>(a, jchunks, jlen, ws); break;
    case 6: em_cp_vec_k<T, VEC, 24, FUSE><<<grid, kEmThreads, 0, s>>>(a, jchunks, jlen, ws); break;
    default:
      if constexpr (!FUSE) {
        if ((a.R + 3) / 4 == 7) em_cp_vec_k<T, VEC, 28, false><<<grid, kEmThreads, 0, s>>>(a, jchunks, jlen, ws);
        else em_cp_vec_k<T, VEC, 32, false><<<grid, kEmThreads, 0, s>>>(a, jchunks, jlen, ws);
      }
      break;
  }
}

void em_cp_pass(const EmCpArgs& a, int prec, double* ws, double* out4, hipStream_t s) {
  AO_REQUIRE(a.R >= 1 && a.R <= kMaxRank && a.I > 0 && a.J > 0 && a.K > 0, "em_cp_pass: bad sizes");
  const bool fuse = a.T != nullptr;
  const bool wj = a.walk != 2;
  AO_REQUIRE((wj ? a.K : a.J) <= 65535, "em_cp_pass: mode too long for one launch");   // grid.y = the fixed index
  AO_REQUIRE(!fuse || em_cp_can_fuse(a, prec), "em_cp_pass: this block cannot take the fused contraction");
  AO_REQUIRE(wj || a.C != nullptr, "em_cp_pass: a matrix block has no third mode to walk");
  const int vec = em_vec(a, prec);
  const int64_t strips = cdiv(a.Ipad, (int64_t)kEmThreads * vec);
  int jchunks; int64_t jlen;
  em_chunking(strips, wj ? a.J : a.K, wj ? a.K : a.J, fuse, &jchunks, &jlen);
  const dim3 grid((unsigned)(strips * jchunks), (unsigned)(wj ? a.K : a.J));
  AO_REQUIRE((size_t)grid.x * grid.y * 4 * sizeof(double) <= em_cp_ws_bytes(a.Ipad, a.J, a.K), "em_cp_pass: workspace");
  if (prec == AOADMM_PREC_F32) {
    if (fuse) em_cp_launch<float, 4, true>(a, jchunks, jlen, ws, grid, s);
    else if (em_wide(a)) em_cp_launch<float, 4, false>(a, jchunks, jlen, ws, grid, s);
    else em_cp_k<float, 64><<<grid, kEmThreads, 0, s>>>(a, jchunks, jlen, ws);
  } else {
    if (fuse) em_cp_launch<double, 2, true>(a, jchunks, jlen, ws, grid, s);
    else if (em_wide(a)) em_cp_launch<double, 2, false>(a, jchunks, jlen, ws, grid, s);
    else em_cp_k<double, 64><<<grid, kEmThreads, 0, s>>>(a, jchunks, jlen, ws);
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
