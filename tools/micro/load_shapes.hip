// Development probe (round 2): which access SHAPE lets a row-tile streaming kernel approach the plain-read rate
// (7.1 TB/s measured) -- independent of the MFMA work?  Every wave owns 128 fp32 rows of the unfolding and walks all
// C columns, like contract16_f32, but only adds the values up.
//   SHAPE 0: the contraction's operand map: one load = 4 columns x 256 B (lane (r16,q): rows 4*r16.., column c+q)
//   SHAPE 1: wave-contiguous: one load = 2 columns x 512 B (lanes 0-31 column c, lanes 32-63 column c+1)
//   layout natural (column stride = M rows) or blocked per wave (column stride = 128 rows: 1 KB contiguous per load)
//   DEPTH = 8-column groups in flight per wave; occupancy limited with dynamic LDS (waves per SIMD)
#include <cstdio>
#include <vector>
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int DEPTH>
__global__ __launch_bounds__(256) void walk_k(const float* __restrict__ X, int64_t ntiles, int64_t wave_stride, int64_t ld,
                                              int64_t ngroups, float* out) {
  extern __shared__ float dummy[];
  const int lane = threadIdx.x & 63;
  const int64_t wt = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wt >= ntiles) return;
  const float* base = X + wt * wave_stride;
  const float* p[4];
  if (SHAPE == 0) {
    const int r16 = lane & 15, q = lane >> 4;
#pragma unroll
    for (int s = 0; s < 4; ++s) p[s] = base + 64 * (s & 1) + 4 * r16 + (4 * (s >> 1) + q) * ld;
  } else {
    const int j = lane & 31, h = lane >> 5;
#pragma unroll
    for (int s = 0; s < 4; ++s) p[s] = base + 4 * j + (2 * s + h) * ld;
  }
  const int64_t gstep = 8 * ld;
  f32x4 x[DEPTH][4];
  float acc = 0.f;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int s = 0; s < 4; ++s) x[d][s] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p[s] + d * gstep));
  int64_t g = 0;
  for (; g + 2 * DEPTH <= ngroups; g += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc += (x[d][s].x + x[d][s].y) + (x[d][s].z + x[d][s].w);
        x[d][s] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p[s] + (g + DEPTH + d) * gstep));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int s = 0; s < 4; ++s) acc += (x[d][s].x + x[d][s].y) + (x[d][s].z + x[d][s].w);
  if (acc == 12345.678f) out[0] = acc + dummy[0];
}

static hipStream_t s;
template <int SHAPE, int DEPTH>
static float run(const float* X, int64_t M, int64_t C, bool blocked, int waves_per_simd, float* out) {
  const int64_t ntiles = M / 128;
  const int64_t wave_stride = blocked ? 128 * C : 128;
  const int64_t ld = blocked ? 128 : M;
  const int blocks_per_cu = waves_per_simd;              // 4 waves per block = 1 wave per SIMD per block
  size_t lds = (size_t)(160 * 1024) / blocks_per_cu - 1024;
  if (lds > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(walk_k<SHAPE, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0, s);
    walk_k<SHAPE, DEPTH><<<(unsigned)((ntiles + 3) / 4), 256, lds, s>>>(X, ntiles, wave_stride, ld, C / 8, out);
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return best;
}

int main() {
  (void)hipStreamCreate(&s);
  const int64_t n = 2000, M = n * n, C = n;
  const size_t bytes = (size_t)M * C * 4;
  float *A, *B, *out;
  if (hipMalloc(&A, bytes) != hipSuccess || hipMalloc(&B, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  (void)hipMalloc(&out, 64);
  (void)hipMemsetAsync(A, 0, bytes, s); (void)hipMemsetAsync(B, 0, bytes, s);
  (void)hipStreamSynchronize(s);
  const float* bufs[2] = {A, B};
  for (int b = 0; b < 2; ++b) {
    printf("buffer %d (%p): ms per 32 GB pass\n", b, bufs[b]);
    for (int blocked = 0; blocked < 2; ++blocked)
      for (int wps : {2, 3, 4, 8}) {
        printf("  %-8s %d waves/SIMD :", blocked ? "blocked" : "natural", wps);
#define RUN(SH, D) printf("  s%d/d%d %.3f", SH, D, run<SH, D>(bufs[b], M, C, blocked, wps, out));
        RUN(0, 2) RUN(0, 3) RUN(0, 4) RUN(0, 6) RUN(1, 2) RUN(1, 3) RUN(1, 4) RUN(1, 6)
        printf("\n");
        fflush(stdout);
      }
  }
  return 0;
}
