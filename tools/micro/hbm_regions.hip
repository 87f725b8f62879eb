// Development probe (round 2): is the speed of a streaming pass a property of the PHYSICAL HBM region the bytes live in,
// and can it be chosen?  Creates the device memory as 1 GiB physical handles (hipMemCreate), maps each one, times a
// streaming read of every handle (several rounds), then maps the fastest / slowest 30-handle sets back to back and
// times the real 2000^3 fp32 contraction pass on them.
#include <algorithm>
#include <cstdio>
#include <vector>
#include "../../matlab-code_amd/csrc/contract.h"
using namespace aoadmm;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void read_k(const f32x4* __restrict__ p, int64_t n4, float* out) {
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4 v = __builtin_nontemporal_load(p + i);
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456f) out[0] = acc;
}

static hipStream_t s;
static float time_read(const void* p, size_t bytes, float* sink) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, s);
  read_k<<<2048, 256, 0, s>>>((const f32x4*)p, (int64_t)(bytes / 16), sink);
  (void)hipEventRecord(e1, s);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return ms;
}
static DevBuf T, frag, F;
static float time_pass(const void* X, const ContractPlan& pl) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  launch_contract(X, AOADMM_PREC_F32, pl, F.d(), pl.C, frag.p, T.p, s, e0, e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return ms;
}

int main() {
  CK(hipSetDevice(0));
  CK(hipStreamCreate(&s));
  const int64_t n = 2000, M = n * n;
  const int R = 20;
  ContractPlan full = make_plan(1, 0, M, M, n, R, AOADMM_PREC_F32);
  T.alloc(full.t_bytes()); frag.alloc(full.frag_bytes(AOADMM_PREC_F32)); F.alloc((size_t)n * R * 8);
  DevBuf sink; sink.alloc(64);
  CK(hipMemset(F.p, 0, (size_t)n * R * 8));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  size_t fr = 0, tot = 0;
  CK(hipMemGetInfo(&fr, &tot));
  const size_t chunk = (size_t)1 << 30;
  const int N = (int)((fr - ((size_t)12 << 30)) / chunk);
  printf("granularity %zu, free %.1f GiB, %d handles of 1 GiB\n", gran, fr / 1073741824.0, N);
  std::vector<hipMemGenericAllocationHandle_t> h(N);
  for (int i = 0; i < N; ++i) CK(hipMemCreate(&h[i], chunk, &prop, 0));
  void* va = nullptr;
  CK(hipMemAddressReserve(&va, (size_t)N * chunk, chunk, nullptr, 0));
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  for (int i = 0; i < N; ++i) CK(hipMemMap((char*)va + (size_t)i * chunk, chunk, 0, h[i], 0));
  CK(hipMemSetAccess(va, (size_t)N * chunk, &acc, 1));
  CK(hipMemsetAsync(va, 0, (size_t)N * chunk, s));
  CK(hipStreamSynchronize(s));
  std::vector<float> best(N, 1e9f);
  for (int round = 0; round < 4; ++round) {
    printf("round %d (us per GiB):", round);
    for (int i = 0; i < N; ++i) {
      const float ms = time_read((char*)va + (size_t)i * chunk, chunk, (float*)sink.p);
      if (round > 0) best[i] = std::min(best[i], ms);
      printf(" %.0f", ms * 1e3);
    }
    printf("\n");
  }
  std::vector<int> order(N);
  for (int i = 0; i < N; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](int a, int b) { return best[a] < best[b]; });
  printf("fastest %.0f us, median %.0f us, slowest %.0f us per GiB\n", best[order[0]] * 1e3, best[order[N / 2]] * 1e3, best[order[N - 1]] * 1e3);
  // remap: three tensors of 30 GiB from the fastest 90 handles, three from the slowest 90
  CK(hipMemUnmap(va, (size_t)N * chunk));
  const int per = 30;
  for (int which = 0; which < 2; ++which) {
    for (int t = 0; t < 3; ++t)
      for (int k = 0; k < per; ++k) {
        const int idx = which == 0 ? order[t * per + k] : order[N - 1 - (t * per + k)];
        CK(hipMemMap((char*)va + ((size_t)t * per + k) * chunk, chunk, 0, h[idx], 0));
      }
    CK(hipMemSetAccess(va, (size_t)3 * per * chunk, &acc, 1));
    for (int t = 0; t < 3; ++t) {
      void* X = (char*)va + (size_t)t * per * chunk;
      (void)time_pass(X, full);
      float a = time_pass(X, full), b = time_pass(X, full), c = time_pass(X, full);
      printf("%s handles, tensor %d: contraction pass %.3f %.3f %.3f ms\n", which == 0 ? "fastest" : "slowest", t, a, b, c);
    }
    CK(hipMemUnmap(va, (size_t)3 * per * chunk));
  }
  for (int i = 0; i < N; ++i) CK(hipMemRelease(h[i]));
  CK(hipMemAddressFree(va, (size_t)N * chunk));
  printf("done\n");
  return 0;
}
