// Device-side pieces of the element-wise / row-wise proximal operators (functions/constraints_to_prox.m), shared by
// the ADMM kernels of admm.hip and the one-workgroup coupled loop of solver.hip.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/aoadmm_hip.h"

namespace aoadmm {

__device__ __forceinline__ double prox_elem(int type, double v, double p0, double p1, double rho) {
  switch (type) {
    case AOADMM_C_NONNEG: return fmax(v, 0.0);                          // project_box(x,0,inf)  (:14)
    case AOADMM_C_BOX: return fmin(fmax(v, p0), p1);                    // (:18)
    case AOADMM_C_L1_REG: {                                             // prox_abs(x,eta/rho) (:48)
      const double g = p0 / rho;
      const double m = fabs(v) - g;
      return m > 0.0 ? copysign(m, v) : 0.0;
    }
    case AOADMM_C_L0_REG: {                                             // prox_zero (:52)
      const double g = p0 / rho;
      return v * v > 2.0 * g ? v : 0.0;
    }
    case AOADMM_C_RIDGE: return 1.0 / (2.0 * (p0 / rho) + 1.0) * v;     // (:60)
    default: return v;
  }
}

// exact projection of v[0..R) onto {x >= 0, sum x = eta}: fixed point of
// tau <- (sum_{v_i > tau} v_i - eta) / #{v_i > tau}  (nested active sets, finite termination)
template <int RMAX>
__device__ __forceinline__ void simplex_regs(double (&v)[RMAX], int R, double eta) {
  double tau = -INFINITY;
  int cnt_prev = -1;
  for (int it = 0; it <= RMAX; ++it) {
    double sum = 0.0;
    int cnt = 0;
#pragma unroll
    for (int r = 0; r < RMAX; ++r)
      if (r < R && v[r] > tau) { sum += v[r]; ++cnt; }
    if (cnt == cnt_prev || cnt == 0) break;
    cnt_prev = cnt;
    tau = (sum - eta) / cnt;
  }
#pragma unroll
  for (int r = 0; r < RMAX; ++r)
    if (r < R) v[r] = fmax(v[r] - tau, 0.0);
}

// branch-free form of the element-wise prox catalogue (same arithmetic per constraint as prox_elem):
//   z = scale * clamp(hard(soft(v, g), thr0), lo, hi)
struct ElemProx { double g, thr0, lo, hi, scale; };
__device__ __forceinline__ ElemProx elem_prox_of(int type, double p0, double p1, double rho) {
  ElemProx e{0.0, -1.0, -INFINITY, INFINITY, 1.0};
  switch (type) {
    case AOADMM_C_NONNEG: e.lo = 0.0; break;                              // project_box(x,0,inf)  (:14)
    case AOADMM_C_BOX: e.lo = p0; e.hi = p1; break;                       // (:18)
    case AOADMM_C_L1_REG: e.g = p0 / rho; break;                          // prox_abs(x,eta/rho) (:48)
    case AOADMM_C_L0_REG: e.thr0 = 2.0 * (p0 / rho); break;               // prox_zero (:52)
    case AOADMM_C_RIDGE: e.scale = 1.0 / (2.0 * (p0 / rho) + 1.0); break; // (:60)
    default: break;
  }
  return e;
}
__device__ __forceinline__ double elem_prox(const ElemProx& e, double v) {
  double s = v;
  if (e.g != 0.0) { const double m = fabs(v) - e.g; s = m > 0.0 ? copysign(m, v) : 0.0; }
  if (!(v * v > e.thr0)) s = 0.0;
  return e.scale * fmin(fmax(s, e.lo), e.hi);
}

}  // namespace aoadmm
