// Microbenchmark 2 (development tool): f32 MFMA rate with 12+12 operand registers and odd data classes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
#include <cmath>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k32(const float* a, const float* b, float* out, int iters) {
  float av[12], bv[12];
  for (int i = 0; i < 12; ++i) { av[i] = a[threadIdx.x * 12 + i]; bv[i] = b[threadIdx.x * 12 + i]; }
  f32x16 acc[4];
  for (int v = 0; v < 4; ++v) for (int i = 0; i < 16; ++i) acc[v][i] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int st = 0; st < 3; ++st)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[v] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[st * 4 + v], bv[st * 4 + s], acc[v], 0, 0, 0);
  }
  float t = 0;
  for (int v = 0; v < 4; ++v) for (int i = 0; i < 16; ++i) t += acc[v][i];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}
int main() {
  const int iters = 4000;
  float *a, *b, *o;
  (void)hipMalloc(&a, 256 * 12 * 4); (void)hipMalloc(&b, 256 * 12 * 4); (void)hipMalloc(&o, 256 * 8192 * 4);
  const char* names[] = {"uniform", "tiny x * [0,1) w/ 37% zeros", "NaN", "Inf", "denormal a", "denormal both", "1e-5 signed a, unif b", "big 1e30"};
  for (int mode = 0; mode < 8; ++mode) {
    std::vector<float> ha(256 * 12), hb(256 * 12);
    for (int i = 0; i < 256 * 12; ++i) {
      double u = rand() / (double)RAND_MAX, w = rand() / (double)RAND_MAX;
      float x = u, y = w;
      if (mode == 1) { x = 1e-5 * (2 * u - 1); y = (i % 32) < 20 ? w : 0.0; }
      if (mode == 2) { x = NAN; y = w; }
      if (mode == 3) { x = INFINITY; y = w; }
      if (mode == 4) { x = 1e-40 * u; y = w; }
      if (mode == 5) { x = 1e-40 * u; y = 1e-41 * w; }
      if (mode == 6) { x = 1e-5 * (2 * u - 1); y = w; }
      if (mode == 7) { x = 1e30 * u; y = 1e30 * w; }
      ha[i] = x; hb[i] = y;
    }
    (void)hipMemcpy(a, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int grid : {768, 7813}) {
      int it = grid == 768 ? iters : iters / 10;
      for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0); k32<<<grid, 256>>>(a, b, o, it); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      }
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      double fl = (double)grid * 4 * it * 48 * 4096.0;
      printf("f32 32x32x2 grid=%5d %-30s %8.3f ms  %7.1f TF/s\n", grid, names[mode], ms, fl / ms / 1e9);
    }
  }
  return 0;
}
