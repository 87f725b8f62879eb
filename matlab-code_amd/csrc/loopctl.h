// Device-side loop control of the ADMM inner loops, shared by the kernels that evaluate it (admm.hip) and the
// Gram kernel that closes a loop (small.hip).
#pragma once
#include "small.h"
#include "device_utils.h"

namespace aoadmm {

static constexpr int kMaxParts = 1024;     // partial-sum slots per parity

__device__ __forceinline__ bool admm_continue(const double* part_prev, int nparts, int it, int max_inner,
                                              double tol_pr, double tol_du, AdmmCtl* ctl, bool writer) {
  const int active = ctl->active;
  if (it == 0) return active != 0;
  const int lane = threadIdx.x & 63;
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  for (int b = lane; b < nparts; b += 64) {
    const double* pb = part_prev + (int64_t)b * 4;
    s0 += pb[0]; s1 += pb[1]; s2 += pb[2]; s3 += pb[3];
  }
  s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3);   // DPP tree: the same total in every lane
  const double pr = sqrt(s0) / sqrt(s1);                                   // :1085
  const double sc = sqrt(s2);
  const double du = sc > 0 ? sqrt(s3) / sc : sqrt(s3);                     // :1087-1092
  const bool cont = it < max_inner && (pr > tol_pr || du > tol_du);        // :600
  if (writer && active) {
    ctl->res[1] = pr;
    ctl->res[3] = du;
    ctl->iters = it;
    if (!cont) ctl->active = 0;
  }
  return active != 0 && cont;
}


// closing record of a loop whose last iteration's residuals have not been evaluated yet (see admm_loop_end_k)
struct LoopEnd {
  const double* part = nullptr;
  int nparts = 0, max_inner = 0;
  double tol_pr = 0, tol_du = 0;
  AdmmCtl* ctl = nullptr;          // null: nothing to close
};
__device__ __forceinline__ void loop_end_eval(const LoopEnd& le) {
  (void)admm_continue(le.part + (int64_t)((le.max_inner + 1) & 1) * kMaxParts * 4, le.nparts, le.max_inner, le.max_inner,
                      le.tol_pr, le.tol_du, le.ctl, (threadIdx.x & 63) == 0);
}

}  // namespace aoadmm
