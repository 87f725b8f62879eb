// Development probe: does the duration of the streaming contraction depend on WHERE in HBM the tensor copy lives?
// (round 2: in the solver the pass on X takes 5.25 ms, the passes on the two permuted copies 5.46-5.50 ms.)
// Times aoadmm::launch_contract (2000^2 rows x 250 columns = 4 GB sub-blocks) along buffers allocated in different ways.
// build: tools/micro/Makefile; run on the GPU box.
#include <cstdio>
#include <string>
#include <vector>
#include "../../matlab-code_amd/csrc/contract.h"
using namespace aoadmm;

__global__ void fill_k(float* p, int64_t n, unsigned seed) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)(i * 2654435761u) ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (float)(h & 0xffff) * (1.0f / 65536.0f) * 1e-4f;
  }
}

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
static int g_read_blocks = 2048;
static float time_read(const void* p, size_t bytes, float* sink) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, s);
  read_k<<<g_read_blocks, 256, 0, s>>>((const f32x4*)p, (int64_t)(bytes / 16), sink);
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

int main(int argc, char** argv) {
  const int64_t n = 2000;
  const int R = 20;
  const int64_t M = n * n;
  (void)hipStreamCreate(&s);
  ContractPlan full = make_plan(1, 0, M, M, n, R, AOADMM_PREC_F32);
  ContractPlan part = make_plan(1, 0, M, M, 250, R, AOADMM_PREC_F32);       // 4 GB of tensor per pass
  T.alloc(full.t_bytes()); frag.alloc(full.frag_bytes(AOADMM_PREC_F32)); F.alloc((size_t)n * R * 8);
  std::vector<double> fh((size_t)n * R);
  for (size_t i = 0; i < fh.size(); ++i) fh[i] = (double)((i * 7919u) % 1000) / 1000.0;
  (void)hipMemcpy(F.p, fh.data(), fh.size() * 8, hipMemcpyHostToDevice);
  const size_t bytes = (size_t)M * n * 4, sub = (size_t)M * 250 * 4;
  auto profile = [&](const char* what, void* base, size_t total) {
    fill_k<<<8192, 256, 0, s>>>((float*)base, (int64_t)(total / 4), 17u);
    (void)hipStreamSynchronize(s);
    for (int var : {0, 1, 2, 3, 4}) {
      g_store_policy = var;
      printf("%-32s store policy %d %p :", what, var, base);
      (void)time_pass(base, part);
      for (size_t off = 0; off + sub <= total; off += sub) {
        float a = time_pass((char*)base + off, part), b = time_pass((char*)base + off, part);
        printf(" %.0f", 1e3 * (a < b ? a : b));          // us per 4 GB
      }
      for (size_t off = 0; off + bytes <= total; off += bytes) {
        (void)time_pass((char*)base + off, full);
        float a = time_pass((char*)base + off, full), b = time_pass((char*)base + off, full);
        printf(" | full %.3f ms", a < b ? a : b);
        ContractPlan bp = make_plan(M / 512, 512 * n, 512, 512, n, R, AOADMM_PREC_F32);
        (void)time_pass((char*)base + off, bp);
        a = time_pass((char*)base + off, bp); b = time_pass((char*)base + off, bp);
        printf(" (blocked512 %.3f)", a < b ? a : b);
      }
      printf("\n");
    }
    fflush(stdout);
  };
  if (argc > 1 && std::string(argv[1]) == "pmc") {
    // one pass per 4 GB piece of a 96e9-byte allocation, in order: run under rocprofv3 --pmc and match the
    // per-dispatch counters of contract16_f32 with the durations printed here (dispatch 0 is a warm-up on piece 0)
    DevBuf arena;
    arena.alloc(3 * bytes);
    fill_k<<<8192, 256, 0, s>>>((float*)arena.p, (int64_t)(3 * bytes / 4), 17u);
    (void)hipStreamSynchronize(s);
    (void)time_pass(arena.p, part);
    printf("piece_us:");
    for (size_t off = 0; off + sub <= 3 * bytes; off += sub) printf(" %.0f", 1e3 * time_pass((char*)arena.p + off, part));
    printf("\n");
    return 0;
  }
  size_t fr = 0, tot = 0;
  (void)hipMemGetInfo(&fr, &tot);
  printf("free %.1f GB of %.1f GB\n", fr / 1e9, tot / 1e9);
  for (int trial = 0; trial < 2; ++trial) {
    DevBuf a, b, c;
    a.alloc(bytes); b.alloc(bytes); c.alloc(bytes);
    profile("separate 32e9 B allocation #0", a.p, bytes);
    profile("separate 32e9 B allocation #1", b.p, bytes);
    profile("separate 32e9 B allocation #2", c.p, bytes);
  }
  for (int trial = 0; trial < 2; ++trial) {
    DevBuf arena;
    arena.alloc(3 * bytes);
    profile("one 96e9 B allocation", arena.p, 3 * bytes);
  }
  return 0;
}
