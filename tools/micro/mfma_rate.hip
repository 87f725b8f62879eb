// Microbenchmark (development tool): sustained v_mfma_f32_32x32x2_f32 / v_mfma_f64_16x16x4_f64 rate
// as a function of the operand data (zeros / ones / uniform random / tiny), one to three waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k32(const float* a, const float* b, float* out, int iters) {
  float av[4], bv[4];
  for (int i = 0; i < 4; ++i) { av[i] = a[threadIdx.x * 4 + i]; bv[i] = b[threadIdx.x * 4 + i]; }
  f32x16 acc[4];
  for (int v = 0; v < 4; ++v) for (int i = 0; i < 16; ++i) acc[v][i] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[v] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[v], bv[s], acc[v], 0, 0, 0);
  }
  float t = 0;
  for (int v = 0; v < 4; ++v) for (int i = 0; i < 16; ++i) t += acc[v][i];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}
__global__ __launch_bounds__(256) void k64(const double* a, const double* b, double* out, int iters) {
  double av[4], bv[4];
  for (int i = 0; i < 4; ++i) { av[i] = a[threadIdx.x * 4 + i]; bv[i] = b[threadIdx.x * 4 + i]; }
  f64x4 acc[4];
  for (int v = 0; v < 4; ++v) for (int i = 0; i < 4; ++i) acc[v][i] = 0.0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int v = 0; v < 4; ++v) acc[v] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[v], bv[s], acc[v], 0, 0, 0);
  }
  double t = 0;
  for (int v = 0; v < 4; ++v) for (int i = 0; i < 4; ++i) t += acc[v][i];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}
int main() {
  const int iters = 20000;
  float *a, *b, *o; double *a64, *b64, *o64;
  hipMalloc(&a, 4096); hipMalloc(&b, 4096); hipMalloc(&o, 256 * 1024 * 16 * 4);
  hipMalloc(&a64, 8192); hipMalloc(&b64, 8192); hipMalloc(&o64, 256 * 1024 * 16 * 8);
  const char* names[] = {"zeros", "ones", "uniform[0,1)", "1e-5*uniform", "signed random"};
  for (int wg_per_cu = 1; wg_per_cu <= 3; wg_per_cu += 2)
  for (int mode = 0; mode < 5; ++mode) {
    std::vector<float> ha(1024), hb(1024); std::vector<double> da(1024), db(1024);
    for (int i = 0; i < 1024; ++i) {
      double u = rand() / (double)RAND_MAX, w = rand() / (double)RAND_MAX;
      double x = mode == 0 ? 0 : mode == 1 ? 1 : mode == 2 ? u : mode == 3 ? 1e-5 * u : 2 * u - 1;
      double y = mode == 0 ? 0 : mode == 1 ? 1 : mode == 2 ? w : mode == 3 ? w : 2 * w - 1;
      ha[i] = x; hb[i] = y; da[i] = x; db[i] = y;
    }
    hipMemcpy(a, ha.data(), 4096, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 4096, hipMemcpyHostToDevice);
    hipMemcpy(a64, da.data(), 8192, hipMemcpyHostToDevice); hipMemcpy(b64, db.data(), 8192, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grid = 256 * wg_per_cu;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0); k32<<<grid, 256>>>(a, b, o, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)grid * 4 * iters * 16 * 4096.0;
    printf("f32 32x32x2  wg/cu=%d %-14s %8.3f ms  %7.1f TF/s\n", wg_per_cu, names[mode], ms, fl / ms / 1e9);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0); k64<<<grid, 256>>>(a64, b64, o64, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    }
    hipEventElapsedTime(&ms, e0, e1);
    fl = (double)grid * 4 * iters * 16 * 2048.0;
    printf("f64 16x16x4  wg/cu=%d %-14s %8.3f ms  %7.1f TF/s\n", wg_per_cu, names[mode], ms, fl / ms / 1e9);
  }
  return 0;
}
