// Development probe: 2000^3 fp64 contraction pass (R = 20) in the variants of contract_f64 (g_f64_mode).
#include <cstdio>
#include <vector>
#include "../../matlab-code_amd/csrc/contract.h"
using namespace aoadmm;
__global__ void fill_k(double* p, int64_t n, unsigned seed) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)(i * 2654435761u) ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (double)(h & 0xffff) * (1.0 / 65536.0) * 1e-4;
  }
}
int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 2000;
  const int R = argc > 2 ? atoi(argv[2]) : 20;
  const int64_t M = n * n;
  hipStream_t s;
  (void)hipStreamCreate(&s);
  ContractPlan pl = make_plan(1, 0, M, M, n, R, AOADMM_PREC_F64);
  DevBuf T, frag, F, X[2];
  T.alloc(pl.t_bytes()); frag.alloc(pl.frag_bytes(AOADMM_PREC_F64)); F.alloc((size_t)n * R * 8);
  (void)hipMemset(F.p, 0, (size_t)n * R * 8);
  for (int b = 0; b < 2; ++b) { X[b].alloc((size_t)M * n * 8); fill_k<<<8192, 256, 0, s>>>(X[b].d(), M * n, 5u + b); }
  (void)hipStreamSynchronize(s);
  for (int mode : {0}) {
    for (int b = 0; b < 2; ++b) {
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        launch_contract(X[b].p, AOADMM_PREC_F64, pl, F.d(), n, frag.p, T.p, s, e0, e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
      }
      printf("mode %d (%s) buffer %d: %.3f ms  %.2f TB/s\n", mode, "shipped kernel",
             b, best, pl.algorithmic_bytes(AOADMM_PREC_F64) / best / 1e9);
      fflush(stdout);
    }
  }
  return 0;
}
