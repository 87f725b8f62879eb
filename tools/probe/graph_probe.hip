// Probe: cost per dependent tiny kernel, plain stream launches vs a captured hipGraph (gfx950).
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe/graph_probe.hip -o tools/probe/graph_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void tiny(double* p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0000001 + 1e-9;
}
int main() {
  const int n = 2000 * 20, per = 200, reps = 50;
  double* d; CK(hipMalloc(&d, n * sizeof(double))); CK(hipMemset(d, 0, n * sizeof(double)));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const int blocks = (n + 255) / 256;
  for (int w = 0; w < 3; ++w) { for (int k = 0; k < per; ++k) tiny<<<blocks, 256, 0, s>>>(d, n); CK(hipStreamSynchronize(s)); }
  auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r) for (int k = 0; k < per; ++k) tiny<<<blocks, 256, 0, s>>>(d, n);
  CK(hipStreamSynchronize(s));
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  printf("stream launches : %.2f us per kernel (%d dependent kernels)\n", us / (reps * per), reps * per);
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int k = 0; k < per; ++k) tiny<<<blocks, 256, 0, s>>>(d, n);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int w = 0; w < 3; ++w) { CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); }
  t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  printf("graph launches  : %.2f us per kernel (%d-kernel graph x %d)\n", us / (reps * per), per, reps);
  return 0;
}
