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

}  // namespace aoadmm
