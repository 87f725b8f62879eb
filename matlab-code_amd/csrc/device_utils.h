// Device-side helpers shared by several translation units.
#pragma once
#include <hip/hip_runtime.h>

namespace aoadmm {

// in-LDS Cholesky of a symmetric R x R matrix (column-major), lower factor; returns false if not PD
__device__ inline bool chol_lds(double* M, int R) {
  __shared__ int bad;
  if (threadIdx.x == 0) bad = 0;
  __syncthreads();
  for (int j = 0; j < R; ++j) {
    if (threadIdx.x == 0) {
      const double d = M[j + R * j];
      if (!(d > 0.0)) bad = 1;
      M[j + R * j] = sqrt(d);
    }
    __syncthreads();
    if (bad) return false;
    const double djj = M[j + R * j];
    for (int i = j + 1 + threadIdx.x; i < R; i += blockDim.x) M[i + R * j] /= djj;
    __syncthreads();
    const int rem = R - j - 1;
    for (int e = threadIdx.x; e < rem * rem; e += blockDim.x) {
      const int i = j + 1 + e % rem, k = j + 1 + e / rem;
      if (i >= k) M[i + R * k] -= M[i + R * j] * M[k + R * j];
    }
    __syncthreads();
  }
  for (int e = threadIdx.x; e < R * R; e += blockDim.x) {
    const int i = e % R, k = e / R;
    if (i < k) M[e] = 0.0;
  }
  __syncthreads();
  return true;
}


// fixed-order block sum (blockDim.x a power of two <= 256); sh must hold blockDim.x doubles
__device__ inline double block_sum_pow2(double v, double* sh) {
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

// sum over a block of exactly 256 threads (four wavefronts); sh4 holds four doubles
__device__ __forceinline__ double block256_sum(double v, double* sh4) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = v;
  __syncthreads();
  const double r = sh4[0] + sh4[1] + sh4[2] + sh4[3];
  __syncthreads();
  return r;
}

// Sum over the 64 lanes of a FULL wavefront, returned in every lane, by DPP moves (v_mov_b32_dpp, a few cycles each)
// instead of six dependent ds_bpermute round trips: inclusive sums move towards higher lanes inside rows of 16
// (row_shr 1, 2, 4, 8 with zeros shifted in), the row totals cross rows (row_bcast:15 into rows 1 and 3, row_bcast:31
// into rows 2 and 3), lane 63 holds the total.  Fixed order.  All 64 lanes must be active.
__device__ __forceinline__ double wave_sum(double v) {
  v += __builtin_amdgcn_update_dpp(0.0, v, 0x111, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0.0, v, 0x112, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0.0, v, 0x114, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0.0, v, 0x118, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0.0, v, 0x142, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0.0, v, 0x143, 0xc, 0xf, false);
  const unsigned long long bits = __builtin_bit_cast(unsigned long long, v);
  const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)bits, 63);
  const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(bits >> 32), 63);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// Inclusive prefix sums over the 64 lanes of a FULL wavefront by the same DPP sequence (Hillis-Steele inside rows of 16,
// then the row totals across rows).  All 64 lanes must be active.
__device__ __forceinline__ double wave_scan_incl(double v) {
  v += __builtin_amdgcn_update_dpp(0.0, v, 0x111, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0.0, v, 0x112, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0.0, v, 0x114, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0.0, v, 0x118, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0.0, v, 0x142, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0.0, v, 0x143, 0xc, 0xf, false);
  return v;
}
__device__ __forceinline__ int wave_scan_incl(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
  return v;
}

}  // namespace aoadmm
