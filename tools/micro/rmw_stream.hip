// Development probe (round 2): what does the memory system give a pass shaped like the EM imputation --
// read a 16-byte vector of the tensor and a 4-byte mask word, write the vector back in place -- with no arithmetic?
// 1000^3 fp32 (4 GB tensor, 1 GB mask).  Variants: loads plain / streaming, stores none / plain / streaming / to a
// second buffer, the per-workgroup walk of the EM kernel (1024 rows x a run of columns) against a flat grid-stride copy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

// LD: 0 plain 1 nt ; ST: 0 none 1 plain 2 nt ; mask: read or not
template <int LD, int ST, bool MASK>
__global__ __launch_bounds__(256) void walk_k(f32x4* X, f32x4* Y, const uint32_t* M, int64_t Ipad4, int64_t J, int jchunks, int64_t jlen, float* sink) {
  const int chunk = blockIdx.x % jchunks;
  const int64_t k = blockIdx.y;
  const int64_t i4 = threadIdx.x;
  if (i4 >= Ipad4) return;
  const int64_t jbeg = chunk * jlen, jend = jbeg + jlen < J ? jbeg + jlen : J;
  f32x4* x = X + Ipad4 * J * k + i4;
  f32x4* y = Y + Ipad4 * J * k + i4;
  const uint32_t* m = M + Ipad4 * J * k + i4;
  constexpr int PD = 4;
  f32x4 q[PD]; uint32_t mq[PD];
#pragma unroll
  for (int p = 0; p < PD; ++p) {
    const int64_t j = jbeg + p < jend ? jbeg + p : jend - 1;
    q[p] = LD ? __builtin_nontemporal_load(x + Ipad4 * j) : x[Ipad4 * j];
    if (MASK) mq[p] = LD ? __builtin_nontemporal_load(m + Ipad4 * j) : m[Ipad4 * j];
  }
  float acc = 0.f;
  for (int64_t j = jbeg; j < jend; j += PD) {
#pragma unroll
    for (int p = 0; p < PD; ++p) {
      f32x4 v = q[p];
      const uint32_t mv = MASK ? mq[p] : 0u;
      const int64_t jn = j + p + PD < jend ? j + p + PD : jend - 1;
      q[p] = LD ? __builtin_nontemporal_load(x + Ipad4 * jn) : x[Ipad4 * jn];
      if (MASK) mq[p] = LD ? __builtin_nontemporal_load(m + Ipad4 * jn) : m[Ipad4 * jn];
      if (j + p < jend) {
        if (mv & 0xff) v.x += 1.f;
        if (mv & 0xff00) v.y += 1.f;
        if (mv & 0xff0000) v.z += 1.f;
        if (mv & 0xff000000) v.w += 1.f;
        if (ST == 0) acc += v.x + v.y + v.z + v.w;
        if (ST == 1) y[Ipad4 * (j + p)] = v;
        if (ST == 2) __builtin_nontemporal_store(v, y + Ipad4 * (j + p));
      }
    }
  }
  if (ST == 0 && acc == 123.456f) sink[0] = acc;
}

template <int LD, int ST>
__global__ __launch_bounds__(256) void flat_k(f32x4* X, f32x4* Y, const uint32_t* M, int64_t n4, float* sink) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 v = LD ? __builtin_nontemporal_load(X + i) : X[i];
    const uint32_t mv = LD ? __builtin_nontemporal_load(M + i) : M[i];
    if (mv & 0xff) v.x += 1.f;
    if (mv & 0xff000000) v.w += 1.f;
    if (ST == 1) Y[i] = v;
    if (ST == 2) __builtin_nontemporal_store(v, Y + i);
  }
}

template <int LD, int ST>
__global__ __launch_bounds__(256) void cyc_k(f32x4* X, const uint32_t* M, int64_t Ipad4, int64_t ncols) {
  const int64_t i4 = threadIdx.x;
  if (i4 >= Ipad4) return;
  const int64_t G = gridDim.x;
  constexpr int PD = 4;
  f32x4 q[PD]; uint32_t mq[PD];
  int64_t c = blockIdx.x;
#pragma unroll
  for (int p = 0; p < PD; ++p) {
    const int64_t cc = c + p * G < ncols ? c + p * G : c;
    q[p] = LD ? __builtin_nontemporal_load(X + Ipad4 * cc + i4) : X[Ipad4 * cc + i4];
    mq[p] = LD ? __builtin_nontemporal_load(M + Ipad4 * cc + i4) : M[Ipad4 * cc + i4];
  }
  for (; c < ncols; c += PD * G) {
#pragma unroll
    for (int p = 0; p < PD; ++p) {
      f32x4 v = q[p];
      const uint32_t mv = mq[p];
      const int64_t cur = c + p * G;
      const int64_t cn = cur + PD * G < ncols ? cur + PD * G : blockIdx.x;
      q[p] = LD ? __builtin_nontemporal_load(X + Ipad4 * cn + i4) : X[Ipad4 * cn + i4];
      mq[p] = LD ? __builtin_nontemporal_load(M + Ipad4 * cn + i4) : M[Ipad4 * cn + i4];
      if (cur < ncols) {
        if (mv & 0xff) v.x += 1.f;
        if (mv & 0xff000000) v.w += 1.f;
        if (ST == 1) X[Ipad4 * cur + i4] = v;
        if (ST == 2) __builtin_nontemporal_store(v, X + Ipad4 * cur + i4);
      }
    }
  }
}

static hipStream_t s;
template <typename F> static float timeit(F f) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(); f();
  (void)hipEventRecord(e0, s);
  for (int i = 0; i < 10; ++i) f();
  (void)hipEventRecord(e1, s);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / 10;
}

int main(int argc, char** argv) {
  CK(hipSetDevice(0));
  CK(hipStreamCreate(&s));
  const int64_t I = argc > 1 ? atoll(argv[1]) : 1000, J = 1000, K = 1000, I4 = I / 4, n4 = I4 * J * K;
  f32x4 *X, *Y; uint32_t* M; float* sink;
  CK(hipMalloc(&X, n4 * 16)); CK(hipMalloc(&Y, n4 * 16)); CK(hipMalloc(&M, n4 * 4)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(X, 0, n4 * 16)); CK(hipMemset(Y, 0, n4 * 16)); CK(hipMemset(M, 1, n4 * 4));
  printf("I = %lld (column = %lld bytes)\n", (long long)I, (long long)I * 4);
  const double gb_r = n4 * 16e-9, gb_m = n4 * 4e-9;
#define WALK(LD, ST, MASK, INPLACE, JC) { \
    const int jc = JC; const int64_t jl = ((J + jc - 1) / jc + 63) / 64 * 64; const int jcs = (int)((J + jl - 1) / jl); \
    const float ms = timeit([&] { walk_k<LD, ST, MASK><<<dim3(jcs, K), 256, 0, s>>>(X, INPLACE ? X : Y, M, I4, J, jcs, jl, sink); }); \
    const double gb = gb_r + (MASK ? gb_m : 0) + (ST ? gb_r : 0); \
    printf("walk  ld=%s st=%-5s mask=%d %s chunks=%d : %.3f ms  %.2f TB/s\n", LD ? "nt" : "plain", ST == 0 ? "none" : ST == 1 ? "plain" : "nt", (int)MASK, INPLACE ? "in place " : "to second", jcs, ms, gb / ms); }
  WALK(1, 0, false, true, 6)
  WALK(1, 0, true, true, 6)
  WALK(0, 0, true, true, 6)
  WALK(1, 2, true, true, 6)
  WALK(1, 2, true, true, 1)
  WALK(1, 2, true, true, 16)
  WALK(1, 1, true, true, 6)
  WALK(0, 1, true, true, 6)
  WALK(0, 2, true, true, 6)
  WALK(1, 2, true, false, 6)
  WALK(1, 1, true, false, 6)
  WALK(1, 2, false, true, 6)
  WALK(1, 2, false, false, 6)
#define FLAT(LD, ST, INPLACE, G) { \
    const float ms = timeit([&] { flat_k<LD, ST><<<G, 256, 0, s>>>(X, INPLACE ? X : Y, M, n4, sink); }); \
    printf("flat  ld=%s st=%-5s %s grid=%d : %.3f ms  %.2f TB/s\n", LD ? "nt" : "plain", ST == 1 ? "plain" : "nt", INPLACE ? "in place " : "to second", G, ms, (2 * gb_r + gb_m) / ms); }
  FLAT(1, 2, true, 1024)
  FLAT(1, 2, true, 2048)
  FLAT(1, 2, true, 4096)
  FLAT(1, 2, true, 8192)
  FLAT(1, 2, true, 16384)
  FLAT(1, 2, true, 65536)
  FLAT(0, 2, true, 8192)
  FLAT(0, 1, true, 8192)
  FLAT(1, 1, true, 8192)
  FLAT(1, 2, false, 2048)
  FLAT(0, 1, true, 2048)
  FLAT(0, 1, false, 2048)
#define CYC(LD, ST, G) { \
    const float ms = timeit([&] { cyc_k<LD, ST><<<G, 256, 0, s>>>(X, M, I4, J * K); }); \
    printf("cyclic columns ld=%s st=%-5s grid=%d : %.3f ms  %.2f TB/s\n", LD ? "nt" : "plain", ST == 1 ? "plain" : "nt", G, ms, (2 * gb_r + gb_m) / ms); }
  CYC(1, 2, 1024)
  CYC(1, 2, 2048)
  CYC(1, 2, 4096)
  CYC(1, 2, 8192)
  CYC(0, 1, 2048)
  CYC(0, 2, 2048)
  return 0;
}
