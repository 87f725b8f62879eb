// Host-side symmetric eigendecomposition (cyclic Jacobi, fp64), used once per model for fixed user
// matrices: the Laplacian of 'quadratic regularization' (constraints_to_prox.m:62-67) and H'*H of the
// transformed couplings (cmtf_fun_AOADMM.m:288-293, :377-382).  A = U diag(w) U'.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace aoadmm {

// A: n x n column-major symmetric (overwritten); w: n eigenvalues; U: n x n column-major eigenvectors.
// Returns the number of sweeps used (negative: not converged in 60 sweeps).
inline int host_sym_eig(int64_t n, std::vector<double>& A, std::vector<double>& w, std::vector<double>& U) {
  U.assign((size_t)n * n, 0.0);
  for (int64_t i = 0; i < n; ++i) U[(size_t)i + (size_t)n * i] = 1.0;
  w.assign((size_t)n, 0.0);
  auto a = [&](int64_t i, int64_t j) -> double& { return A[(size_t)i + (size_t)n * j]; };
  double scale = 0.0;
  for (int64_t j = 0; j < n; ++j)
    for (int64_t i = 0; i < n; ++i) scale += a(i, j) * a(i, j);
  scale = std::sqrt(scale);
  int sweeps = -1;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
    for (int64_t j = 0; j < n; ++j)
      for (int64_t i = 0; i < j; ++i) off += a(i, j) * a(i, j);
    if (std::sqrt(2.0 * off) <= 1e-15 * scale) { sweeps = sweep; break; }
    for (int64_t p = 0; p < n - 1; ++p)
      for (int64_t q = p + 1; q < n; ++q) {
        const double apq = a(p, q);
        if (apq == 0.0) continue;
        const double theta = (a(q, q) - a(p, p)) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int64_t k = 0; k < n; ++k) {          // columns p, q
          const double akp = a(k, p), akq = a(k, q);
          a(k, p) = c * akp - s * akq;
          a(k, q) = s * akp + c * akq;
        }
        for (int64_t k = 0; k < n; ++k) {          // rows p, q
          const double apk = a(p, k), aqk = a(q, k);
          a(p, k) = c * apk - s * aqk;
          a(q, k) = s * apk + c * aqk;
        }
        for (int64_t k = 0; k < n; ++k) {
          const double ukp = U[(size_t)k + (size_t)n * p], ukq = U[(size_t)k + (size_t)n * q];
          U[(size_t)k + (size_t)n * p] = c * ukp - s * ukq;
          U[(size_t)k + (size_t)n * q] = s * ukp + c * ukq;
        }
      }
  }
  for (int64_t i = 0; i < n; ++i) w[(size_t)i] = a(i, i);
  return sweeps;
}

}  // namespace aoadmm
