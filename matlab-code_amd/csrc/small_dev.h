// Device side of sys_build (small.h): the R x R system of one mode update in ONE wave with every matrix in registers.
// A header so that it can ride in other kernels' launches as one extra workgroup (the reductions over T in
// contract.hip: the system depends on Gram matrices only, not on the MTTKRP they finish, so it is built beside it
// instead of 10-12 us behind it).
#pragma once
#include "small.h"

namespace aoadmm {

__device__ __forceinline__ double readlane_d(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

template <int RMAX, bool EXACT>          // EXACT: R == RMAX, every rank test folds away (a third of the instructions)
__device__ __forceinline__ void sys_build_dev(const SysBuild& sb) {   // one full wave, t = threadIdx.x in [0, 64)
  const int R = EXACT ? RMAX : sb.R, t = threadIdx.x;
  const bool mine = t < R;
  double row[RMAX];                              // row t of C, then of B + nrho*rho/2*I, then of L
  // all loads of one Gram matrix are issued together on clamped (always valid) addresses: one memory
  // round trip per matrix instead of one per entry
  {
    const double* g0 = sb.ngram == 0 ? sb.Cpre : sb.grams[0];
#pragma unroll
    for (int c = 0; c < RMAX; ++c) row[c] = g0[(mine ? t : 0) + R * (c < R ? c : 0)];
    // C = ones .* G_transp_G{j}...  (:98-103).  The second matrix (3-way blocks have two) is loaded in the same round
    // trip as the first: its loads are issued before the first product needs the first's values.
    if (sb.ngram >= 2) {
      const double* g1 = sb.grams[1];
      double tmp[RMAX];
#pragma unroll
      for (int c = 0; c < RMAX; ++c) tmp[c] = g1[(mine ? t : 0) + R * (c < R ? c : 0)];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < RMAX; ++c) row[c] *= tmp[c];
    }
    for (int k = 2; k < sb.ngram; ++k) {
      const double* g = sb.grams[k];
      double tmp[RMAX];
#pragma unroll
      for (int c = 0; c < RMAX; ++c) tmp[c] = g[(mine ? t : 0) + R * (c < R ? c : 0)];
#pragma unroll
      for (int c = 0; c < RMAX; ++c) row[c] *= tmp[c];
    }
#pragma unroll
    for (int c = 0; c < RMAX; ++c) {
      if (mine && c < R) sb.C[t + R * c] = row[c];
      else row[c] = 0.0;
    }
  }
  double tr = 0.0;                               // diagonal summed in index order by every lane
#pragma unroll
  for (int r = 0; r < RMAX; ++r)
    if (r < R) tr += readlane_d(row[r], r);
  const double rho = sb.rho_scale * (tr / R);    // rho = trace(C)/size(C,1)  (:115)
  if (t == 0) sb.rho[0] = rho;
#pragma unroll
  for (int c = 0; c < RMAX; ++c) {
    double b = sb.w * row[c];                    // B = w*C (:116)
    if (c == t) b += sb.ridge + sb.bsum_half;    // :117-119, :126
    if (mine && c < R) sb.Bsys[t + R * c] = b;
    if (c == t) b += sb.nrho * (rho / 2);        // :141 / :269-271
    if (sb.Madd && mine && c < R) b += (rho / 2) * sb.Madd[t + R * c];   // :314
    row[c] = (mine && c < R) ? b : 0.0;
  }
  // chol(B','lower') (:142), right-looking.  Lane t keeps L(t, 0..t); its entries right of the diagonal
  // are never read by another lane.
  // One reciprocal square root per column instead of a square root and a division (each ~30 dependent fp64
  // instructions on the kernel's single wave); the reciprocal diagonal is kept for the substitutions below.
  bool ok = true;
  double invd[RMAX];
#pragma unroll
  for (int j = 0; j < RMAX; ++j) {
    invd[j] = 0.0;
    if (j < R && ok) {
      const double d = readlane_d(row[j], j);
      if (!(d > 0.0)) {
        ok = false;
      } else {
        const double inv = rsqrt(d);
        const double dj = d * inv;
        invd[j] = inv;
        const double lij = row[j] * inv;
        row[j] = (t == j) ? dj : lij;
#pragma unroll
        for (int k = j + 1; k < RMAX; ++k)
          if (k < R) row[k] -= lij * readlane_d(lij, k);
      }
    }
  }
  if (ok && mine) {
#pragma unroll
    for (int k = 0; k < RMAX; ++k)
      if (k < R) sb.L[t + R * k] = (k <= t) ? row[k] : 0.0;
  }
  if (ok && sb.Binv) {
    // inv(L*L'): the system matrix of an ADMM mode carries +rho/2*I with rho = trace(C)/R, so
    // cond(B) <= 2R+1: the explicit inverse loses nothing measurable and turns the per-row solve into R
    // independent dot products.  Lane t owns column t of X = inv(L) (forward substitution on e_t; entries
    // above the diagonal come out as exact zeros), then Binv(i,t) = sum_k X(k,i)*X(k,t).
    double xc[RMAX];
#pragma unroll
    for (int i = 0; i < RMAX; ++i) {
      xc[i] = 0.0;
      if (i < R) {
        double v = (i == t) ? 1.0 : 0.0;
#pragma unroll
        for (int q = 0; q < i; ++q) v -= readlane_d(row[q], i) * xc[q];
        xc[i] = v * invd[i];
      }
    }
#pragma unroll
    for (int i = 0; i < RMAX; ++i) {
      if (i < R) {
        double acc = 0.0;
#pragma unroll
        for (int k = i; k < RMAX; ++k)            // X(k,i) = 0 for k < i
          if (k < R) acc += readlane_d(xc[k], i) * xc[k];
        if (mine) sb.Binv[t + R * i] = acc;       // symmetric: stored as Binv(t,i), coalesced
      }
    }
  }
  if (t == 0 && sb.ctl) {
    sb.ctl->active = 1;
    sb.ctl->iters = 0;
    if (!ok) sb.ctl->notpd = 1;
    sb.ctl->res[0] = sb.ctl->res[1] = sb.ctl->res[2] = sb.ctl->res[3] = 0.0;
  }
}

// the rider: called by every thread of the extra workgroup; wave 0 builds the system (R <= 20 classes only: their
// register footprint, <= 104 VGPRs, is below that of the host kernels)
__device__ __forceinline__ void sys_build_rider(const SysBuild& sb) {
  if (threadIdx.x >= 64) return;
  // one wave with a long chain of dependent instructions among the host kernel's bandwidth-bound waves: raised issue
  // priority, and the rank-exact instantiation when there is one (a third fewer instructions) -- beside a 14-us
  // reduction the rider took 20 us without them (10 us as a kernel of its own)
  __builtin_amdgcn_s_setprio(3);
  if (sb.R == 20) sys_build_dev<20, true>(sb);         // (every rank class in both forms crashed the compiler's register coalescer)
  else if (sb.R <= 4) sys_build_dev<4, false>(sb);
  else if (sb.R <= 8) sys_build_dev<8, false>(sb);
  else if (sb.R <= 12) sys_build_dev<12, false>(sb);
  else if (sb.R <= 16) sys_build_dev<16, false>(sb);
  else sys_build_dev<20, false>(sb);
}
constexpr int kSysRiderMaxR = 20;

}  // namespace aoadmm
