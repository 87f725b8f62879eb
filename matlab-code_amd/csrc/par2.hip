// PARAFAC2 slab kernels -- see par2.h.  Reference lines are cited at each kernel.
#include "par2.h"

#include "device_utils.h"

namespace aoadmm {

#define CTL_GUARD(ctl) \
  if ((ctl) != nullptr && (ctl)->active == 0) return;

static constexpr int kP2Threads = 64;

// ---------------------------------------------------------------------------
__global__ void par2_xkb_k(const double* X, const double* B, P2Dims d, double* T1) {
  const int k = d.k0 + blockIdx.x;
  const int64_t o = d.off[k];
  const int Jk = (int)(d.off[k + 1] - o);
  const double* Xk = X + (int64_t)d.I * o;
  const double* Bk = B + o * d.R;
  for (int e = threadIdx.x; e < d.I * d.R; e += blockDim.x) {
    const int i = e % d.I, r = e / d.I;
    // four independent partial sums: the loads of four columns are in flight together (one accumulator made every
    // step wait for its own load: J_k dependent L2 round trips, 15 us per launch at J_k ~ 90)
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int j = 0;
    for (; j + 3 < Jk; j += 4) {
      const double x0 = Xk[i + (int64_t)d.I * j], x1 = Xk[i + (int64_t)d.I * (j + 1)];
      const double x2 = Xk[i + (int64_t)d.I * (j + 2)], x3 = Xk[i + (int64_t)d.I * (j + 3)];
      const double b0 = Bk[j + Jk * r], b1 = Bk[j + 1 + Jk * r], b2 = Bk[j + 2 + Jk * r], b3 = Bk[j + 3 + Jk * r];
      a0 += x0 * b0; a1 += x1 * b1; a2 += x2 * b2; a3 += x3 * b3;
    }
    for (; j < Jk; ++j) a0 += Xk[i + (int64_t)d.I * j] * Bk[j + Jk * r];
    T1[(int64_t)k * d.I * d.R + e] = (a0 + a1) + (a2 + a3);
  }
}
void par2_xkb(const double* X, const double* B, const P2Dims& d, double* T1, hipStream_t s) {
  par2_xkb_k<<<d.k1 - d.k0, 128, 0, s>>>(X, B, d, T1);
  AO_KERNEL_CHECK();
}

__global__ void par2_gram_k(const double* B, P2Dims d, double* GB) {
  const int k = d.k0 + blockIdx.x;
  const int64_t o = d.off[k];
  const int Jk = (int)(d.off[k + 1] - o);
  const double* Bk = B + o * d.R;
  for (int e = threadIdx.x; e < d.R * d.R; e += blockDim.x) {
    const int r = e % d.R, q = e / d.R;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;   // independent partial sums: four pairs of loads in flight
    int j = 0;
    for (; j + 3 < Jk; j += 4) {
      a0 += Bk[j + Jk * r] * Bk[j + Jk * q]; a1 += Bk[j + 1 + Jk * r] * Bk[j + 1 + Jk * q];
      a2 += Bk[j + 2 + Jk * r] * Bk[j + 2 + Jk * q]; a3 += Bk[j + 3 + Jk * r] * Bk[j + 3 + Jk * q];
    }
    for (; j < Jk; ++j) a0 += Bk[j + Jk * r] * Bk[j + Jk * q];
    GB[(int64_t)k * d.R * d.R + e] = (a0 + a1) + (a2 + a3);
  }
}
void par2_gram(const double* B, const P2Dims& d, double* GB, hipStream_t s) {
  par2_gram_k<<<d.k1 - d.k0, kP2Threads, 0, s>>>(B, d, GB);
  AO_KERNEL_CHECK();
}

// Sum over the K slabs of f(k) for a tile of `ew` consecutive outputs: 256 threads = ew outputs x ng = 256/ew slab
// groups.  Every group adds its slabs in order, then the groups are folded by a fixed pairwise tree, so the result
// does not depend on scheduling.  The value is returned in the threads of group 0.
constexpr int kKsumThreads = 256;
template <class F>
__device__ inline double ksum_tile(bool live, int g, int ng, int ew, int k0, int k1, double* red, F f) {
  double acc = 0.0;
  if (live)
    for (int k = k0 + g; k < k1; k += ng) acc += f(k);
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = ng >> 1; s > 0; s >>= 1) {
    if (g < s) red[threadIdx.x] += red[threadIdx.x + s * ew];
    __syncthreads();
  }
  return red[threadIdx.x];
}
static inline int ksum_tile_width(int64_t n) { return n >= 16384 ? 64 : (n >= 2048 ? 16 : 4); }

// A{m} = sum_k X_k B_k diag(C(k,:)) ; C{m} = sum_k diag(C(k,:)) (B_k'B_k) diag(C(k,:))   (:163-164)
__global__ __launch_bounds__(kKsumThreads) void par2_modeA_combine_k(const double* T1, const double* Cfac,
                                                                      const double* GB, P2Dims d, double* Amt,
                                                                      double* Csys, int ew) {
  __shared__ double red[kKsumThreads];
  const int nA = d.I * d.R, nC = d.R * d.R, K = d.K;
  const int ng = kKsumThreads / ew, ex = threadIdx.x % ew, g = threadIdx.x / ew;
  const int e = blockIdx.x * ew + ex;
  double v;
  if (blockIdx.x * ew < nA) {            // tiles never straddle nA: the host rounds nA up to a tile boundary
    const bool lv = e < nA;
    const int r = lv ? e / d.I : 0;
    v = ksum_tile(lv, g, ng, ew, d.k0, d.k1, red, [&](int k) { return T1[(int64_t)k * nA + e] * Cfac[k + K * r]; });
    if (lv && g == 0) Amt[e] = v;
  } else {
    const int f = e - (int)((nA + ew - 1) / ew) * ew;
    const bool lv = f < nC;
    const int r = lv ? f % d.R : 0, q = lv ? f / d.R : 0;
    v = ksum_tile(lv, g, ng, ew, d.k0, d.k1, red,
                  [&](int k) { return Cfac[k + K * r] * GB[(int64_t)k * nC + f] * Cfac[k + K * q]; });
    if (lv && g == 0) Csys[f] = v;
  }
}
void par2_modeA_combine(const double* T1, const double* Cfac, const double* GB, const P2Dims& d, double* Amt,
                        double* Csys, hipStream_t s) {
  const int64_t nA = (int64_t)d.I * d.R, nC = (int64_t)d.R * d.R;
  const int ew = ksum_tile_width(nA + nC);
  const unsigned nb = (unsigned)(cdiv(nA, ew) + cdiv(nC, ew));
  par2_modeA_combine_k<<<nb, kKsumThreads, 0, s>>>(T1, Cfac, GB, d, Amt, Csys, ew);
  AO_KERNEL_CHECK();
}

__global__ void par2_xta_k(const double* X, const double* A, const double* Cfac, double w, P2Dims d, double* Ak) {
  const int k = d.k0 + blockIdx.x;
  const int64_t o = d.off[k];
  const int Jk = (int)(d.off[k + 1] - o);
  const double* Xk = X + (int64_t)d.I * o;
  double* out = Ak + o * d.R;
  for (int e = threadIdx.x; e < Jk * d.R; e += blockDim.x) {
    const int j = e % Jk, r = e / Jk;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;   // independent partial sums (see par2_xkb_k)
    int i = 0;
    for (; i + 3 < d.I; i += 4) {
      const double* x = Xk + i + (int64_t)d.I * j;
      const double* a = A + i + d.I * r;
      a0 += x[0] * a[0]; a1 += x[1] * a[1]; a2 += x[2] * a[2]; a3 += x[3] * a[3];
    }
    for (; i < d.I; ++i) a0 += Xk[i + (int64_t)d.I * j] * A[i + d.I * r];
    out[e] = w * ((a0 + a1) + (a2 + a3)) * Cfac[k + d.K * r];           // w * X_k' * A * diag(C(k,:))   (:193)
  }
}
void par2_xta(const double* X, const double* A, const double* Cfac, double w, const P2Dims& d, double* Ak,
              hipStream_t s) {
  par2_xta_k<<<d.k1 - d.k0, 128, 0, s>>>(X, A, Cfac, w, d, Ak);
  AO_KERNEL_CHECK();
}

__global__ void par2_b_system_k(const double* GA, const double* Cfac, double w, double ridge, double bsum_half,
                                double rho_scale, int nrho, P2Dims d, double* rho, double* L, AdmmCtl* ctl) {
  extern __shared__ double sh[];
  __shared__ double rk;
  const int k = d.k0 + blockIdx.x, R = d.R;
  if (blockIdx.x == 0 && threadIdx.x == 0 && ctl) {                    // opens the B_k loop (what ctl_reset does; notpd stays)
    ctl->active = 1;
    ctl->iters = 0;
    ctl->res[0] = ctl->res[1] = ctl->res[2] = ctl->res[3] = 0.0;
  }
  for (int e = threadIdx.x; e < R * R; e += blockDim.x) {
    const int r = e % R, q = e / R;
    sh[e] = Cfac[k + d.K * r] * GA[e] * Cfac[k + d.K * q];           // C_k = D_k (A'A) D_k   (:194)
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int r = 0; r < R; ++r) t += sh[r + R * r];
    rk = rho_scale * (t / R);                                          // :195-198
    rho[k] = rk;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < R * R; e += blockDim.x) {
    double b = w * sh[e];
    if (e % R == e / R) b += nrho * (rk / 2) + ridge + bsum_half;      // :199-211
    sh[e] = b;
  }
  __syncthreads();
  const bool ok = chol_lds(sh, R);                                     // chol(B,'lower')  (:212)
  if (ok)
    for (int e = threadIdx.x; e < R * R; e += blockDim.x) L[(int64_t)k * R * R + e] = sh[e];
  else if (threadIdx.x == 0 && ctl) ctl->notpd = 1;
}
void par2_b_system(const double* GA, const double* Cfac, double w, double ridge, double bsum_half, double rho_scale,
                   int nrho, const P2Dims& d, double* rho, double* L, AdmmCtl* ctl, hipStream_t s) {
  par2_b_system_k<<<d.k1 - d.k0, kP2Threads, (size_t)d.R * d.R * sizeof(double), s>>>(GA, Cfac, w, ridge, bsum_half, rho_scale,
                                                                            nrho, d, rho, L, ctl);
  AO_KERNEL_CHECK();
}

// ---------------------------------------------------------------------------
// B-mode inner iteration
// ---------------------------------------------------------------------------
// One inner iteration of a slab up to the sum over the slabs is ONE kernel (par2_b_slab_k below): the three steps are
// device functions that hand their results on through global memory + a workgroup barrier.
// per slab: B_k update (:526-530), W = (B_k + mu_k) * DeltaB' (:532), Pold = P
// RMAX >= R is a compile-time bound: with it the per-row vectors x, bm live in registers (fully unrolled, statically
// indexed) instead of scratch memory, which the runtime-R version needed 1 KB per lane of.
template <int RMAX>
__device__ __forceinline__ void par2_b_primal_dev(const P2BArgs& a, const P2Dims& d, int k, double* sh) {
  const int R = d.R;                      // sh: L_k (R*R) then DeltaB (R*R)
  const int64_t o = d.off[k];
  const int Jk = (int)(d.off[k + 1] - o);
  double* Lsh = sh;
  double* Dsh = sh + R * R;
  for (int e = threadIdx.x; e < R * R; e += blockDim.x) {
    Lsh[e] = a.L[(int64_t)k * R * R + e];
    Dsh[e] = a.DeltaB[e];
  }
  __syncthreads();
  const double rh = a.rho[k] / 2;
  const int64_t base = o * R;
  for (int j = threadIdx.x; j < Jk; j += blockDim.x) {
    double x[RMAX], bm[RMAX], pv[RMAX];
#pragma unroll
    for (int q = 0; q < RMAX; ++q) pv[q] = q < R ? a.P[base + j + Jk * q] : 0.0;
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
      x[r] = 0.0;
      if (r < R) {
        double pd = 0.0;
#pragma unroll
        for (int q = 0; q < RMAX; ++q)
          if (q < R) pd += pv[q] * Dsh[q + R * r];                                       // (P_k*DeltaB)(j,r)
        double v = a.Ak[base + j + Jk * r] + rh * (pd - a.mu[base + j + Jk * r]);
        if (a.use_constr) v += rh * (a.Z[base + j + Jk * r] - a.muZ[base + j + Jk * r]);    // :527-529
        x[r] = v;
      }
    }
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {                    // forward: x*L' = rhs
      if (r < R) {
        double v = x[r];
#pragma unroll
        for (int q = 0; q < RMAX; ++q)
          if (q < r) v -= Lsh[r + R * q] * x[q];
        x[r] = v / Lsh[r + R * r];
      }
    }
#pragma unroll
    for (int r = RMAX - 1; r >= 0; --r) {               // backward: x*L = y
      if (r < R) {
        double v = x[r];
#pragma unroll
        for (int q = 0; q < RMAX; ++q)
          if (q > r && q < R) v -= Lsh[q + R * r] * x[q];
        x[r] = v / Lsh[r + R * r];
      }
    }
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
      bm[r] = 0.0;
      if (r < R) {
        a.B[base + j + Jk * r] = x[r];
        bm[r] = x[r] + a.mu[base + j + Jk * r];
        a.Pold[base + j + Jk * r] = pv[r];
      }
    }
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {                    // W(j,r) = sum_q (B+mu)(j,q) * DeltaB(r,q)
      if (r < R) {
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < RMAX; ++q)
          if (q < R) v += bm[q] * Dsh[r + R * q];
        a.W[base + j + Jk * r] = v;
      }
    }
  }
}

// P_k = U*V' of svd(W_k,'econ') (:532-534) by one-sided (Hestenes) Jacobi: W*Jrot has orthogonal
// columns, U = W*Jrot/sigma, V = Jrot.  One wave per slab: the three column products of a pair are reduced by a
// butterfly over the 64 lanes, which leaves bit-identical sums in every lane, so each lane derives the rotation
// itself and nothing is exchanged through memory.  W_k is staged in LDS when it fits (in_lds), else rotated in place.
static_assert(kP2Threads == 64, "par2_polar_dev reduces over exactly one wavefront");
__device__ inline double wave_sum64(double v) { return wave_sum(v); }   // DPP tree (device_utils.h), the same bits in every lane
__device__ __forceinline__ void par2_polar_dev(double* W, double* P, const P2Dims& d, int k, int in_lds, double* Jr,
                                              double* Jwarm = nullptr, int warm_valid = 0) {
  const int R = d.R, lane = threadIdx.x;  // Jr: R*R, then W_k (n*R) when in_lds
  const int64_t o = d.off[k];
  const int n = (int)(d.off[k + 1] - o);
  double* Pk = P + o * R;
  double* Wk = W + o * R;
  if (in_lds) {
    double* Wl = Jr + R * R;
    for (int e = lane; e < n * R; e += 64) Wl[e] = Wk[e];
    Wk = Wl;
  }
  const bool warm = Jwarm != nullptr && warm_valid && in_lds;   // needs the untouched copy in global memory as source
  for (int e = lane; e < R * R; e += 64) Jr[e] = warm ? Jwarm[(int64_t)k * R * R + e] : ((e % R == e / R) ? 1.0 : 0.0);
  __syncthreads();
  if (warm) {
    // W <- W * J_prev, row by row (each lane owns its rows): the columns are almost orthogonal before the first sweep
    const double* Wg = W + o * R;                      // the slab as the primal step left it
    for (int e = lane; e < n * R; e += 64) {
      const int i = e % n, p = e / n;
      double acc = 0.0;
      for (int q = 0; q < R; ++q) acc += Wg[i + n * q] * Jr[q + R * p];
      Wk[e] = acc;
    }
    __syncthreads();
  }
  for (int sweep = 0; sweep < 60; ++sweep) {
    bool rotated = false;
    for (int p = 0; p < R - 1; ++p)
      for (int q = p + 1; q < R; ++q) {
        double* wp = Wk + n * p;
        double* wq = Wk + n * q;
        double al = 0, be = 0, ga = 0;
        for (int i = lane; i < n; i += 64) { al += wp[i] * wp[i]; be += wq[i] * wq[i]; ga += wp[i] * wq[i]; }
        al = wave_sum64(al); be = wave_sum64(be); ga = wave_sum64(ga);
        if (ga != 0.0 && fabs(ga) > 1e-15 * sqrt(al * be)) {           // uniform over the wave
          const double zeta = (be - al) / (2.0 * ga);
          const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
          const double c = 1.0 / sqrt(1.0 + t * t);
          const double sn = c * t;
          rotated = true;
          for (int i = lane; i < n; i += 64) {
            const double x = wp[i], y = wq[i];
            wp[i] = c * x - sn * y;
            wq[i] = sn * x + c * y;
          }
          for (int i = lane; i < R; i += 64) {
            const double x = Jr[i + R * p], y = Jr[i + R * q];
            Jr[i + R * p] = c * x - sn * y;
            Jr[i + R * q] = sn * x + c * y;
          }
        }
      }
    if (!rotated) break;
  }
  for (int p = 0; p < R; ++p) {
    double* wp = Wk + n * p;
    double al = 0;
    for (int i = lane; i < n; i += 64) al += wp[i] * wp[i];
    const double sg = sqrt(wave_sum64(al));
    for (int i = lane; i < n; i += 64) wp[i] = sg > 0 ? wp[i] / sg : 0.0;
  }
  __syncthreads();
  if (Jwarm != nullptr)
    for (int e = lane; e < R * R; e += 64) Jwarm[(int64_t)k * R * R + e] = Jr[e];
  for (int e = lane; e < n * R; e += 64) {
    const int i = e % n, r = e / n;
    double acc = 0.0;
    for (int q = 0; q < R; ++q) acc += Wk[i + n * q] * Jr[r + R * q];
    Pk[e] = acc;
  }
}

// part[k] = rho_k * P_k' * (B_k + mu_k)   (:541).  With R*R < 64 outputs the rows are split over ng = 64/(R*R) groups
// of lanes whose partial sums are added in group order (LDS), so a rank-3 block keeps 63 lanes busy instead of 9.
__device__ __forceinline__ void par2_deltab_part_dev(const P2BArgs& a, const P2Dims& d, int k, double* sh) {
  const int R = d.R, RR = R * R;
  const int64_t o = d.off[k];
  const int Jk = (int)(d.off[k + 1] - o);
  const int64_t base = o * R;
  const int ng = RR < 64 ? 64 / RR : 1;
  if (ng > 1) {
    const int g = threadIdx.x / RR, e = threadIdx.x % RR;
    if (g < ng) {
      const int r = e % R, q = e / R;
      double acc = 0.0, acc1 = 0.0;                   // two rows per step: six loads in flight instead of three
      int j = g;
      for (; j + ng < Jk; j += 2 * ng) {
        const double p0 = a.P[base + j + Jk * r], p1 = a.P[base + j + ng + Jk * r];
        const double b0 = a.B[base + j + Jk * q] + a.mu[base + j + Jk * q];
        const double b1 = a.B[base + j + ng + Jk * q] + a.mu[base + j + ng + Jk * q];
        acc += p0 * b0; acc1 += p1 * b1;
      }
      for (; j < Jk; j += ng) acc += a.P[base + j + Jk * r] * (a.B[base + j + Jk * q] + a.mu[base + j + Jk * q]);
      sh[g * RR + e] = acc + acc1;
    }
    __syncthreads();
    if ((int)threadIdx.x < RR) {
      double tot = 0.0;
      for (int g2 = 0; g2 < ng; ++g2) tot += sh[g2 * RR + threadIdx.x];
      a.part[(int64_t)k * RR + threadIdx.x] = a.rho[k] * tot;
    }
    return;
  }
  for (int e = threadIdx.x; e < RR; e += blockDim.x) {
    const int r = e % R, q = e / R;
    double acc = 0.0;
    for (int j = 0; j < Jk; ++j) acc += a.P[base + j + Jk * r] * (a.B[base + j + Jk * q] + a.mu[base + j + Jk * q]);
    a.part[(int64_t)k * RR + e] = a.rho[k] * acc;
  }
}
// The same three steps for SHORT slabs (J_k <= 64 * NR rows, R <= RMAX <= 4) with the slab in registers: lane l owns rows
// l, l + 64, ...; B_k, mu_k, P_k, W_k of those rows, the slab's Cholesky factor, DeltaB and the Jacobi rotation never
// leave the register file between the steps.  The version above hands W, B, P from step to step through global memory
// and LDS (three dependent memory round trips and two workgroup barriers per call, the column products of every
// rotation read from LDS): 16-24 us per call at R = 3, 44 % of an outer iteration of BASELINE config 4.  Here a rotation
// is three lane-local products, three DPP wave sums (bit-identical in every lane, so every lane derives the same
// rotation) and a lane-local update.  Same arithmetic in the same order as par2_polar_dev (sums over a lane's rows first,
// then the wave), hence the same polar factor to the last bit; the DeltaB contribution sums in wave order instead of
// lane groups (a rounding-level difference).
template <int RMAX, int NR>
__device__ __forceinline__ void par2_b_slab_regs_dev(const P2BArgs& a, const P2Dims& d, int k) {
  const int R = d.R, lane = threadIdx.x;
  const int64_t o = d.off[k];
  const int Jk = (int)(d.off[k + 1] - o);
  const int64_t base = o * R;
  // ---- every global load of the call in one round trip (clamped addresses, no exec-mask branches in between)
  double Lr[RMAX][RMAX], Dr[RMAX][RMAX], Jr[RMAX][RMAX];
  const bool warm = a.Jrot != nullptr && a.jrot_valid;
#pragma unroll
  for (int c = 0; c < RMAX; ++c)
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
      const bool in = r < R && c < R;
      const int e = in ? r + R * c : 0;
      Lr[r][c] = a.L[(int64_t)k * R * R + e];
      Dr[r][c] = a.DeltaB[e];
      const double jw = warm ? a.Jrot[(int64_t)k * R * R + e] : ((r == c) ? 1.0 : 0.0);
      Jr[r][c] = in ? jw : ((r == c) ? 1.0 : 0.0);
    }
  const double rhok = a.rho[k];
  double pv[NR][RMAX], ak[NR][RMAX], mu[NR][RMAX], zc[NR][RMAX];
  bool have[NR];
#pragma unroll
  for (int u = 0; u < NR; ++u) {
    const int j = lane + 64 * u;
    have[u] = j < Jk;
    const int jc = have[u] ? j : 0;
#pragma unroll
    for (int q = 0; q < RMAX; ++q) {
      const int64_t at = base + jc + (int64_t)Jk * (q < R ? q : 0);
      pv[u][q] = a.P[at]; ak[u][q] = a.Ak[at]; mu[u][q] = a.mu[at];
      zc[u][q] = a.use_constr ? a.Z[at] - a.muZ[at] : 0.0;
    }
  }
  const double rh = rhok / 2;
  double bm[NR][RMAX], w[NR][RMAX];
#pragma unroll
  for (int u = 0; u < NR; ++u) {
    double x[RMAX];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
      x[r] = 0.0;
      if (r < R) {
        double pd = 0.0;
#pragma unroll
        for (int q = 0; q < RMAX; ++q)
          if (q < R) pd += pv[u][q] * Dr[q][r];                                           // (P_k*DeltaB)(j,r)
        double v = ak[u][r] + rh * (pd - mu[u][r]);
        if (a.use_constr) v += rh * zc[u][r];                                             // :527-529
        x[r] = v;
      }
    }
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {                    // forward: x*L' = rhs
      if (r < R) {
        double v = x[r];
#pragma unroll
        for (int q = 0; q < RMAX; ++q)
          if (q < r) v -= Lr[r][q] * x[q];
        x[r] = v / Lr[r][r];
      }
    }
#pragma unroll
    for (int r = RMAX - 1; r >= 0; --r) {               // backward: x*L = y
      if (r < R) {
        double v = x[r];
#pragma unroll
        for (int q = 0; q < RMAX; ++q)
          if (q > r && q < R) v -= Lr[q][r] * x[q];
        x[r] = v / Lr[r][r];
      }
    }
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
      bm[u][r] = 0.0;
      if (r < R) {
        bm[u][r] = have[u] ? x[r] + mu[u][r] : 0.0;
        if (have[u]) {
          const int64_t at = base + lane + 64 * u + (int64_t)Jk * r;
          a.B[at] = x[r];
          a.Pold[at] = pv[u][r];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {                    // W(j,r) = sum_q (B+mu)(j,q) * DeltaB(r,q); rows past J_k: zero
      double v = 0.0;
      if (r < R) {
#pragma unroll
        for (int q = 0; q < RMAX; ++q)
          if (q < R) v += bm[u][q] * Dr[r][q];
      }
      w[u][r] = v;
    }
  }
  // ---- polar factor of W_k by one-sided Jacobi (:532-534), as par2_polar_dev
  if (warm) {
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      double t[RMAX];
#pragma unroll
      for (int p = 0; p < RMAX; ++p) {
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < RMAX; ++q)
          if (q < R) acc += w[u][q] * Jr[q][p];
        t[p] = acc;
      }
#pragma unroll
      for (int p = 0; p < RMAX; ++p)
        if (p < R) w[u][p] = t[p];
    }
  }
  for (int sweep = 0; sweep < 60; ++sweep) {
    bool rotated = false;
#pragma unroll
    for (int p = 0; p < RMAX - 1; ++p)
#pragma unroll
      for (int q = p + 1; q < RMAX; ++q) {
        if (q < R) {                                   // uniform
          double al = 0, be = 0, ga = 0;
#pragma unroll
          for (int u = 0; u < NR; ++u) { al += w[u][p] * w[u][p]; be += w[u][q] * w[u][q]; ga += w[u][p] * w[u][q]; }
          al = wave_sum(al); be = wave_sum(be); ga = wave_sum(ga);
          if (ga != 0.0 && fabs(ga) > 1e-15 * sqrt(al * be)) {           // uniform over the wave
            const double zeta = (be - al) / (2.0 * ga);
            const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
            const double c = 1.0 / sqrt(1.0 + t * t);
            const double sn = c * t;
            rotated = true;
#pragma unroll
            for (int u = 0; u < NR; ++u) {
              const double x = w[u][p], y = w[u][q];
              w[u][p] = c * x - sn * y;
              w[u][q] = sn * x + c * y;
            }
#pragma unroll
            for (int i = 0; i < RMAX; ++i) {
              const double x = Jr[i][p], y = Jr[i][q];
              Jr[i][p] = c * x - sn * y;
              Jr[i][q] = sn * x + c * y;
            }
          }
        }
      }
    if (!rotated) break;
  }
#pragma unroll
  for (int p = 0; p < RMAX; ++p) {
    if (p < R) {
      double al = 0;
#pragma unroll
      for (int u = 0; u < NR; ++u) al += w[u][p] * w[u][p];
      const double sg = sqrt(wave_sum(al));
#pragma unroll
      for (int u = 0; u < NR; ++u) w[u][p] = sg > 0 ? w[u][p] / sg : 0.0;
    }
  }
  if (a.Jrot != nullptr && lane == 0) {
#pragma unroll
    for (int c = 0; c < RMAX; ++c)
#pragma unroll
      for (int r = 0; r < RMAX; ++r)
        if (r < R && c < R) a.Jrot[(int64_t)k * R * R + r + R * c] = Jr[r][c];
  }
  // P_k = U * Jr'  and this slab's DeltaB contribution  rho_k * P_k' * (B_k + mu_k)   (:541)
  double pn[NR][RMAX];
#pragma unroll
  for (int u = 0; u < NR; ++u)
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
      double acc = 0.0;
      if (r < R) {
#pragma unroll
        for (int q = 0; q < RMAX; ++q)
          if (q < R) acc += w[u][q] * Jr[r][q];
        if (have[u]) a.P[base + lane + 64 * u + (int64_t)Jk * r] = acc;
      }
      pn[u][r] = have[u] ? acc : 0.0;
    }
  double tot[RMAX][RMAX];
#pragma unroll
  for (int q = 0; q < RMAX; ++q)
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
      tot[r][q] = 0.0;
      if (r < R && q < R) {
        double acc = 0.0;
#pragma unroll
        for (int u = 0; u < NR; ++u) acc += pn[u][r] * bm[u][q];
        tot[r][q] = wave_sum(acc);
      }
    }
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < RMAX; ++q)
#pragma unroll
      for (int r = 0; r < RMAX; ++r)
        if (r < R && q < R) a.part[(int64_t)k * R * R + r + R * q] = rhok * tot[r][q];
  }
}
constexpr int kP2RegsMaxR = 4;
// rows per lane of the register form for a block whose longest slab has Jmax rows (0: not applicable)
static int par2_regs_rows(const P2Dims& d) {
  static const bool off = getenv("AOADMM_NO_PAR2_REGS") != nullptr;           // development switch
  if (off || d.R > kP2RegsMaxR) return 0;
  return d.Jmax <= 64 ? 1 : (d.Jmax <= 128 ? 2 : (d.Jmax <= 256 ? 4 : 0));
}
template <int NR>
__global__ __launch_bounds__(kP2Threads) void par2_b_slab_regs_k(P2BArgs a, P2Dims d, const AdmmCtl* ctl) {
  CTL_GUARD(ctl);
  par2_b_slab_regs_dev<kP2RegsMaxR, NR>(a, d, d.k0 + blockIdx.x);
}

// primal update, polar factor and DeltaB contribution of slab k in one launch
template <int RMAX>
__global__ __launch_bounds__(kP2Threads) void par2_b_slab_k(P2BArgs a, P2Dims d, const AdmmCtl* ctl, int in_lds) {
  CTL_GUARD(ctl);
  extern __shared__ double sh[];          // max(2*R*R, R*R + Jmax*R [in_lds], 64) doubles, reused by the three steps
  const int k = d.k0 + blockIdx.x;
  par2_b_primal_dev<RMAX>(a, d, k, sh);
  __syncthreads();                        // W, B, Pold of this slab are in global memory for the whole workgroup
  par2_polar_dev(a.W, a.P, d, k, in_lds, sh);
  __syncthreads();
  par2_deltab_part_dev(a, d, k, sh);
}
// DeltaB_old = DeltaB ; DeltaB = sum_k part[k] / sum_k rho_k   (:537-544).  With slabs sharded over ranks
// (psum != nullptr) the kernel leaves this rank's partial sums psum[0..R*R) and psum[R*R] = sum rho_k; after the
// all-reduce par2_deltab_apply_k finishes the division.
__global__ __launch_bounds__(kKsumThreads) void par2_deltab_combine_k(P2BArgs a, P2Dims d, const AdmmCtl* ctl, int ew,
                                                                       double* psum) {
  CTL_GUARD(ctl);
  __shared__ double red[kKsumThreads];
  __shared__ double red2[kKsumThreads];
  const int R = d.R;
  double sr = 0.0;
  for (int k = d.k0 + threadIdx.x; k < d.k1; k += kKsumThreads) sr += a.rho[k];
  sr = block_sum_pow2(sr, red2);
  const int ng = kKsumThreads / ew, ex = threadIdx.x % ew, g = threadIdx.x / ew;
  const int e = blockIdx.x * ew + ex;
  const bool lv = e < R * R;
  const double acc = ksum_tile(lv, g, ng, ew, d.k0, d.k1, red, [&](int k) { return a.part[(int64_t)k * R * R + e]; });
  if (psum) {
    if (lv && g == 0) psum[e] = acc;
    if (blockIdx.x == 0 && threadIdx.x == 0) psum[R * R] = sr;
  } else if (lv && g == 0) {
    a.DeltaBold[e] = a.DeltaB[e];
    a.DeltaB[e] = acc / sr;
  }
}
__global__ void par2_deltab_apply_k(P2BArgs a, int R, const double* psum, const AdmmCtl* ctl) {
  CTL_GUARD(ctl);
  for (int e = threadIdx.x; e < R * R; e += blockDim.x) {
    a.DeltaBold[e] = a.DeltaB[e];
    a.DeltaB[e] = psum[e] / psum[R * R];
  }
}
// mu_k += B_k - P_k*DeltaB (:546); norms[k] = ||B-P*D||^2, ||B||^2, ||Pold*Dold - P*D||^2, ||mu||^2 (:583-584)
__global__ __launch_bounds__(kP2Threads) void par2_b_dual_k(P2BArgs a, P2Dims d, const AdmmCtl* ctl) {
  CTL_GUARD(ctl);
  extern __shared__ double sh2[];         // DeltaB, DeltaBold
  __shared__ double red[kP2Threads];
  const int k = d.k0 + blockIdx.x, R = d.R;
  const int64_t o = d.off[k];
  const int Jk = (int)(d.off[k + 1] - o);
  const int64_t base = o * R;
  double* Dn = sh2;
  double* Do = sh2 + R * R;
  for (int e = threadIdx.x; e < R * R; e += blockDim.x) { Dn[e] = a.DeltaB[e]; Do[e] = a.DeltaBold[e]; }
  __syncthreads();
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  for (int e = threadIdx.x; e < Jk * R; e += blockDim.x) {
    const int j = e % Jk, r = e / Jk;
    double pd = 0.0, po = 0.0;
    for (int q = 0; q < R; ++q) {
      pd += a.P[base + j + Jk * q] * Dn[q + R * r];
      po += a.Pold[base + j + Jk * q] * Do[q + R * r];
    }
    const double b = a.B[base + e];
    const double m = a.mu[base + e] + b - pd;
    a.mu[base + e] = m;
    s0 += (b - pd) * (b - pd); s1 += b * b; s2 += (po - pd) * (po - pd); s3 += m * m;
  }
  s0 = block_sum_pow2(s0, red); s1 = block_sum_pow2(s1, red); s2 = block_sum_pow2(s2, red); s3 = block_sum_pow2(s3, red);
  if (threadIdx.x == 0) {
    double* nk = a.norms + (int64_t)k * 8;
    nk[0] = s0; nk[1] = s1; nk[2] = s2; nk[3] = s3;
  }
}

// ---------------------------------------------------------------------------
// Two launches per inner iteration instead of four (unconstrained B_k, slabs not sharded, R <= 8): the sum over the
// slabs that finishes DeltaB (:537-544) moves to the head of the dual kernel and the residual means with the while
// test (:520, :583-584) to the head of the NEXT iteration's slab kernel.  Every workgroup recomputes the same small sums
// from the same data in the same order, so all of them hold the same DeltaB and take the same decision; workgroup 0
// publishes them.  DeltaB alternates between the two buffers by iteration parity (a workgroup may still be reading the
// old one while workgroup 0 writes the new one); par2_b_close_k records the last iteration's residuals and moves the
// final pair into place.
// ---------------------------------------------------------------------------
struct P2Fold {
  int it, max_inner;
  double tpc, tpz, tdc, tdz;
};
// while test at the head of iteration `it` from the norms iteration it-1 left (kP2Threads = 64 lanes, all active)
__device__ __forceinline__ bool par2_b_head(const double* norms, const P2Dims& d, const P2Fold& f, AdmmCtl* ctl, bool writer) {
  const int active = ctl->active;
  if (f.it == 0) return active != 0;
  double pc = 0.0, dc = 0.0;
  for (int k = d.k0 + (int)threadIdx.x; k < d.k1; k += kP2Threads) {
    const double* nk = norms + (int64_t)k * 8;
    pc += sqrt(nk[0]) / sqrt(nk[1]) / d.K;                      // :583
    dc += sqrt(nk[2]) / sqrt(nk[3]) / d.K;                      // :584 (no zero check in the reference)
  }
  pc = wave_sum(pc); dc = wave_sum(dc);
  const bool cont = f.it < f.max_inner && (pc > f.tpc || 0.0 > f.tpz || dc > f.tdc || 0.0 > f.tdz);   // :520
  if (writer && active) {
    ctl->res[0] = pc; ctl->res[1] = 0.0; ctl->res[2] = dc; ctl->res[3] = 0.0;
    ctl->iters = f.it;
    if (!cont) ctl->active = 0;
  }
  return active != 0 && cont;
}
template <int RMAX>
__global__ __launch_bounds__(kP2Threads) void par2_b_slab_fold_k(P2BArgs a, P2Dims d, AdmmCtl* ctl, int in_lds, P2Fold f) {
  extern __shared__ double sh[];
  if (!par2_b_head(a.norms, d, f, ctl, blockIdx.x == 0 && threadIdx.x == 0)) return;   // the same in every workgroup
  const int k = d.k0 + blockIdx.x;
  par2_b_primal_dev<RMAX>(a, d, k, sh);
  __syncthreads();
  par2_polar_dev(a.W, a.P, d, k, in_lds, sh, a.Jrot, a.jrot_valid);
  __syncthreads();
  par2_deltab_part_dev(a, d, k, sh);
}
template <int NR>
__global__ __launch_bounds__(kP2Threads) void par2_b_slab_fold_regs_k(P2BArgs a, P2Dims d, AdmmCtl* ctl, P2Fold f) {
  if (!par2_b_head(a.norms, d, f, ctl, blockIdx.x == 0 && threadIdx.x == 0)) return;   // the same in every workgroup
  par2_b_slab_regs_dev<kP2RegsMaxR, NR>(a, d, d.k0 + blockIdx.x);
}
// a.DeltaB: DeltaB of this iteration (read); a.DeltaBold: receives the new one
template <int RRMAX>
__global__ __launch_bounds__(kP2Threads) void par2_b_dual_fold_k(P2BArgs a, P2Dims d, const AdmmCtl* ctl) {
  if (ctl->active == 0) return;
  extern __shared__ double sh2[];         // Dn (RR) | Do (RR) | lane partials [64][RR + 1]
  __shared__ double red[kP2Threads + 1];   // R*R + 1 totals: R*R = 64 at R = 8
  const int k = d.k0 + blockIdx.x, R = d.R, RR = R * R, W = RR + 1;
  const int lane = threadIdx.x;
  double* Dn = sh2;
  double* Do = sh2 + RR;
  double* lp = sh2 + 2 * RR;
  {
    double acc[RRMAX], sr = 0.0;
#pragma unroll
    for (int e = 0; e < RRMAX; ++e) acc[e] = 0.0;
    // UB slabs per step, their loads issued before the first addition (one slab per step was one L2 round trip per
    // 64 slabs, in sequence); added in slab order
    constexpr int UB = RRMAX <= 16 ? 4 : 2;
    for (int kk = d.k0 + lane; kk < d.k1; kk += UB * kP2Threads) {
      double v[UB][RRMAX], rv[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int ku = kk + u * kP2Threads;
        const double* pk = a.part + (int64_t)(ku < d.k1 ? ku : kk) * RR;
#pragma unroll
        for (int e = 0; e < RRMAX; ++e) v[u][e] = e < RR ? pk[e] : 0.0;
        rv[u] = a.rho[ku < d.k1 ? ku : kk];
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        if (kk + u * kP2Threads < d.k1) {
#pragma unroll
          for (int e = 0; e < RRMAX; ++e) acc[e] += v[u][e];
          sr += rv[u];
        }
      }
    }
#pragma unroll
    for (int e = 0; e < RRMAX; ++e)
      if (e < RR) lp[lane * W + e] = acc[e];
    lp[lane * W + RR] = sr;
    __syncthreads();
    for (int col = lane; col <= RR; col += kP2Threads) {   // lanes in order: the sum does not depend on scheduling
      double tot = 0.0;
      for (int q = 0; q < kP2Threads; ++q) tot += lp[q * W + col];
      red[col] = tot;
    }
    __syncthreads();
    if (lane < RR) {
      const double dn = red[lane] / red[RR];                                    // :544
      Dn[lane] = dn;
      Do[lane] = a.DeltaB[lane];
      if (blockIdx.x == 0) a.DeltaBold[lane] = dn;
    }
    __syncthreads();
  }
  const int64_t o = d.off[k];
  const int Jk = (int)(d.off[k + 1] - o);
  const int64_t base = o * R;
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  // two entries per step: the loads of both are in flight together
  const int ne = Jk * R, st = blockDim.x;
  for (int e = threadIdx.x; e < ne; e += 2 * st) {
    const int e1 = e + st;
    const bool two = e1 < ne;
    const int e1c = two ? e1 : e;
    const int j0 = e % Jk, r0 = e / Jk, j1 = e1c % Jk, r1 = e1c / Jk;
    double pd0 = 0.0, po0 = 0.0, pd1 = 0.0, po1 = 0.0;
    for (int q = 0; q < R; ++q) {
      const double p0 = a.P[base + j0 + Jk * q], o0 = a.Pold[base + j0 + Jk * q];
      const double p1 = a.P[base + j1 + Jk * q], o1 = a.Pold[base + j1 + Jk * q];
      pd0 += p0 * Dn[q + R * r0]; po0 += o0 * Do[q + R * r0];
      pd1 += p1 * Dn[q + R * r1]; po1 += o1 * Do[q + R * r1];
    }
    const double b0 = a.B[base + e], b1 = a.B[base + e1c];
    const double mo0 = a.mu[base + e], mo1 = a.mu[base + e1c];
    const double m0 = mo0 + b0 - pd0;                                            // :546
    a.mu[base + e] = m0;
    s0 += (b0 - pd0) * (b0 - pd0); s1 += b0 * b0; s2 += (po0 - pd0) * (po0 - pd0); s3 += m0 * m0;
    if (two) {
      const double m1 = mo1 + b1 - pd1;
      a.mu[base + e1] = m1;
      s0 += (b1 - pd1) * (b1 - pd1); s1 += b1 * b1; s2 += (po1 - pd1) * (po1 - pd1); s3 += m1 * m1;
    }
  }
  s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3);
  if (threadIdx.x == 0) {
    double* nk = a.norms + (int64_t)k * 8;
    nk[0] = s0; nk[1] = s1; nk[2] = s2; nk[3] = s3;
  }
}
// closes the loop: residuals of the last iteration, and the final DeltaB / its predecessor into DeltaB / DeltaBold
// (after c iterations the current one sits in buffer c & 1, buffer 0 being DeltaB)
__global__ __launch_bounds__(kP2Threads) void par2_b_close_k(P2BArgs a, P2Dims d, AdmmCtl* ctl, P2Fold f) {
  const int c = ctl->active ? f.max_inner : ctl->iters;      // iterations that ran (an earlier exit recorded its count)
  (void)par2_b_head(a.norms, d, f, ctl, threadIdx.x == 0);
  const int RR = d.R * d.R;
  if ((c & 1) == 0) return;
  for (int e = threadIdx.x; e < RR; e += kP2Threads) {
    const double cur = a.DeltaBold[e], old = a.DeltaB[e];
    a.DeltaB[e] = cur; a.DeltaBold[e] = old;
  }
}

bool par2_b_loop_folded_ok(const P2Dims& d, bool constrained, bool sharded) {
  static const bool off = getenv("AOADMM_NO_PAR2_FOLD") != nullptr;           // development switch
  return !off && !constrained && !sharded && d.R <= 8;
}

void par2_b_loop_folded(const P2BArgs& a0, const P2Dims& d, AdmmCtl* ctl, int max_inner, double tpc, double tpz, double tdc,
                        double tdz, hipStream_t s) {
  AO_REQUIRE(par2_b_loop_folded_ok(d, a0.use_constr != 0, false), "par2_b_loop_folded: not applicable");
  const int RR = d.R * d.R;
  const size_t rr = (size_t)RR * sizeof(double);
  const unsigned nk = (unsigned)(d.k1 - d.k0);
  const size_t wl = (size_t)d.Jmax * d.R * sizeof(double);
  const int in_lds = rr + wl <= 48 * 1024;
  const size_t lds = std::max<size_t>(std::max<size_t>(2 * rr, rr + (in_lds ? wl : 0)), 64 * sizeof(double));
  const size_t lds2 = (size_t)(2 * RR + kP2Threads * (RR + 1)) * sizeof(double);
  P2Fold f{0, max_inner, tpc, tpz, tdc, tdz};
  for (int it = 0; it < max_inner; ++it) {
    P2BArgs a = a0;
    a.DeltaB = (it & 1) ? a0.DeltaBold : a0.DeltaB;
    a.DeltaBold = (it & 1) ? a0.DeltaB : a0.DeltaBold;
    a.jrot_valid = it >= 1 ? 1 : 0;                  // the first inner iteration of every loop starts cold
    f.it = it;
    const int nr = par2_regs_rows(d);
    if (nr == 1) par2_b_slab_fold_regs_k<1><<<nk, kP2Threads, 0, s>>>(a, d, ctl, f);
    else if (nr == 2) par2_b_slab_fold_regs_k<2><<<nk, kP2Threads, 0, s>>>(a, d, ctl, f);
    else if (nr == 4) par2_b_slab_fold_regs_k<4><<<nk, kP2Threads, 0, s>>>(a, d, ctl, f);
    else if (d.R <= 4) par2_b_slab_fold_k<4><<<nk, kP2Threads, lds, s>>>(a, d, ctl, in_lds, f);
    else par2_b_slab_fold_k<8><<<nk, kP2Threads, lds, s>>>(a, d, ctl, in_lds, f);
    AO_KERNEL_CHECK();
    if (RR <= 16) par2_b_dual_fold_k<16><<<nk, kP2Threads, lds2, s>>>(a, d, ctl);
    else par2_b_dual_fold_k<64><<<nk, kP2Threads, lds2, s>>>(a, d, ctl);
    AO_KERNEL_CHECK();
  }
  f.it = max_inner;
  par2_b_close_k<<<1, kP2Threads, 0, s>>>(a0, d, ctl, f);
  AO_KERNEL_CHECK();
}

void par2_b_iteration(const P2BArgs& a, const P2Dims& d, const AdmmCtl* ctl, hipStream_t s, double* psum,
                      const P2AllReduce& allreduce) {
  const size_t rr = (size_t)d.R * d.R * sizeof(double);
  const unsigned nk = (unsigned)(d.k1 - d.k0);
  const size_t wl = (size_t)d.Jmax * d.R * sizeof(double);
  const int in_lds = rr + wl <= 48 * 1024;
  const size_t lds = std::max<size_t>(std::max<size_t>(2 * rr, rr + (in_lds ? wl : 0)), 64 * sizeof(double));
  const int nr = par2_regs_rows(d);
  if (nr == 1) par2_b_slab_regs_k<1><<<nk, kP2Threads, 0, s>>>(a, d, ctl);
  else if (nr == 2) par2_b_slab_regs_k<2><<<nk, kP2Threads, 0, s>>>(a, d, ctl);
  else if (nr == 4) par2_b_slab_regs_k<4><<<nk, kP2Threads, 0, s>>>(a, d, ctl);
  else if (d.R <= 4) par2_b_slab_k<4><<<nk, kP2Threads, lds, s>>>(a, d, ctl, in_lds);
  else if (d.R <= 8) par2_b_slab_k<8><<<nk, kP2Threads, lds, s>>>(a, d, ctl, in_lds);
  else if (d.R <= 16) par2_b_slab_k<16><<<nk, kP2Threads, lds, s>>>(a, d, ctl, in_lds);
  else par2_b_slab_k<kMaxRank><<<nk, kP2Threads, lds, s>>>(a, d, ctl, in_lds);
  AO_KERNEL_CHECK();
  const int ew = ksum_tile_width((int64_t)d.R * d.R);
  par2_deltab_combine_k<<<(unsigned)cdiv((int64_t)d.R * d.R, ew), kKsumThreads, 0, s>>>(a, d, ctl, ew, psum);
  AO_KERNEL_CHECK();
  if (psum) {
    allreduce(psum, (int64_t)d.R * d.R + 1);          // every rank, whatever ctl says: the ranks must stay in step
    par2_deltab_apply_k<<<1, 256, 0, s>>>(a, d.R, psum, ctl);
    AO_KERNEL_CHECK();
  }
  par2_b_dual_k<<<nk, kP2Threads, 2 * rr, s>>>(a, d, ctl);
  AO_KERNEL_CHECK();
}

// ---- constraint on B_k ------------------------------------------------------------------------
__global__ void par2_bz_pre_k(const double* B, const double* Z, const double* muZ, double* Zold, double* V, int64_t n,
                              const AdmmCtl* ctl) {
  CTL_GUARD(ctl);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    V[i] = B[i] + muZ[i];
    Zold[i] = Z[i];
  }
}
__global__ __launch_bounds__(kP2Threads) void par2_bz_post_k(const double* B, const double* Z, double* muZ,
                                                              const double* Zold, P2Dims d, double* norms,
                                                              const AdmmCtl* ctl) {
  CTL_GUARD(ctl);
  __shared__ double red[kP2Threads];
  const int k = d.k0 + blockIdx.x, R = d.R;
  const int64_t o = d.off[k];
  const int Jk = (int)(d.off[k + 1] - o);
  const int64_t base = o * R;
  double s0 = 0, s1 = 0, s2 = 0;
  for (int e = threadIdx.x; e < Jk * R; e += blockDim.x) {
    const double b = B[base + e], z = Z[base + e];
    const double m = muZ[base + e] + b - z;            // :569
    muZ[base + e] = m;
    s0 += (b - z) * (b - z); s1 += m * m; s2 += (Zold[base + e] - z) * (Zold[base + e] - z);
  }
  s0 = block_sum_pow2(s0, red); s1 = block_sum_pow2(s1, red); s2 = block_sum_pow2(s2, red);
  if (threadIdx.x == 0) {
    double* nk = norms + (int64_t)k * 8;
    nk[4] = s0; nk[5] = s1; nk[6] = s2;
  }
}
// t_smoothness_prox.m:1-58 -- for every entry (j,r): tridiagonal system over the K slabs
//   diag_k = 4*eta + rho_k (2*eta + rho_k at both ends), off-diagonals c = -2*eta, rhs_k = rho_k*V_k(j,r),
// solved by the same Gaussian elimination (no pivoting, :42-46) and back substitution (:49-56) as the
// reference.  The eliminated diagonal is the same for every entry; each thread keeps it in a local array.
constexpr int kTsmoothMaxK = 64;
__global__ void par2_tsmooth_k(const double* V, double* Z, const double* rho, double eta, P2Dims d, const AdmmCtl* ctl) {
  CTL_GUARD(ctl);
  const int64_t n = d.off[1] * d.R;                    // entries per slab (all slabs equal)
  const int K = d.K;
  const double c = -2.0 * eta;
  double dg[kTsmoothMaxK];
  for (int k = 0; k < K; ++k) {
    double a = 4.0 * eta + rho[k];                     // :25-34
    if (k == 0) a -= 2.0 * eta;                        // :37
    if (k == K - 1) a -= 2.0 * eta;                    // :38
    if (k > 0) a -= (c / dg[k - 1]) * c;               // :43-44
    dg[k] = a;
  }
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    double rprev = rho[0] * V[e];                      // :8-10
    Z[e] = rprev;                                      // eliminated rhs kept in Z until back substitution
    for (int k = 1; k < K; ++k) {
      const double m = c / dg[k - 1];                  // :43
      const double rk = rho[k] * V[(int64_t)k * n + e] - m * rprev;   // :45
      Z[(int64_t)k * n + e] = rk;
      rprev = rk;
    }
    double q = rprev / dg[K - 1];                      // :50
    Z[(int64_t)(K - 1) * n + e] = q;
    for (int k = K - 2; k >= 0; --k) {                 // :53-56
      q = (Z[(int64_t)k * n + e] - c * q) / dg[k];
      Z[(int64_t)k * n + e] = q;
    }
  }
}

void par2_b_constraint(const ProxSpec& ps, const double* B, double* Z, double* muZ, double* Zold, double* V,
                       const double* rho, const P2Dims& d, double* prox_ws, double* norms, const AdmmCtl* ctl,
                       hipStream_t s) {
  const int64_t* off = d.off_h;
  const int64_t e0 = off[d.k0] * d.R;                  // this rank's slabs are one contiguous range
  const int64_t n = (off[d.k1] - off[d.k0]) * d.R;
  int64_t nb = cdiv(n, 256);
  if (nb > 1024) nb = 1024;
  par2_bz_pre_k<<<(unsigned)nb, 256, 0, s>>>(B + e0, Z + e0, muZ + e0, Zold + e0, V + e0, n, ctl);
  AO_KERNEL_CHECK();
  // Z_k = prox(B_k + muZ_k, rho_k) slab by slab (:568): every catalogue entry goes through prox_apply
  if (ps.type == AOADMM_C_TPARAFAC2) {
    AO_REQUIRE(d.K <= kTsmoothMaxK, "tPARAFAC2 on the device supports up to %d slabs", kTsmoothMaxK);
    AO_REQUIRE(d.k0 == 0 && d.k1 == d.K, "tPARAFAC2 couples neighbouring slabs: the block cannot be slab-sharded");
    const int64_t ne = off[1] * d.R;
    par2_tsmooth_k<<<(unsigned)cdiv(ne, 128), 128, 0, s>>>(V, Z, rho, ps.p0, d, ctl);
    AO_KERNEL_CHECK();
  } else
  for (int k = d.k0; k < d.k1; ++k) {
    const int64_t Jk = off[k + 1] - off[k];
    prox_apply(ps, V + off[k] * d.R, Jk, Z + off[k] * d.R, Jk, Jk, d.R, rho + k, 1.0, prox_ws, ctl, s,
               Zold + off[k] * d.R, Jk);
  }
  par2_bz_post_k<<<d.k1 - d.k0, kP2Threads, 0, s>>>(B, Z, muZ, Zold, d, norms, ctl);
  AO_KERNEL_CHECK();
}

// Residual means over the slabs (:571-577, :583-584) and the loop condition (:520).  part4 != nullptr (slabs sharded
// over ranks): leave this rank's share of the four means in part4; par2_b_finalize_apply_k decides after the all-reduce.
__device__ inline void par2_b_decide(AdmmCtl* ctl, double pc, double pz, double dc, double dz, int max_inner, double tpc,
                                     double tpz, double tdc, double tdz) {
  ctl->res[0] = pc; ctl->res[1] = pz; ctl->res[2] = dc; ctl->res[3] = dz;
  const int it = ctl->iters + 1;
  ctl->iters = it;
  ctl->active = (it < max_inner && (pc > tpc || pz > tpz || dc > tdc || dz > tdz)) ? 1 : 0;   // :520
}
__global__ __launch_bounds__(kKsumThreads) void par2_b_finalize_k(const double* norms, int K, int k0, int k1,
                                                                   int use_constr, AdmmCtl* ctl, int max_inner,
                                                                   double tpc, double tpz, double tdc, double tdz,
                                                                   double* part4) {
  if (ctl->active == 0) return;
  __shared__ double red[kKsumThreads];
  double pc = 0, dc = 0, pz = 0, dz = 0;
  for (int k = k0 + threadIdx.x; k < k1; k += kKsumThreads) {
    const double* nk = norms + (int64_t)k * 8;
    const double nb = sqrt(nk[1]);
    pc += sqrt(nk[0]) / nb / K;                                 // :583
    dc += sqrt(nk[2]) / sqrt(nk[3]) / K;                        // :584 (no zero check in the reference)
    if (use_constr) {
      pz += sqrt(nk[4]) / nb / K;                               // :571
      const double sc = sqrt(nk[5]);
      dz += (sc > 0 ? sqrt(nk[6]) / sc : sqrt(nk[6])) / K;       // :572-577
    }
  }
  pc = block_sum_pow2(pc, red); dc = block_sum_pow2(dc, red);
  pz = block_sum_pow2(pz, red); dz = block_sum_pow2(dz, red);
  if (threadIdx.x != 0) return;
  if (part4) { part4[0] = pc; part4[1] = pz; part4[2] = dc; part4[3] = dz; return; }
  par2_b_decide(ctl, pc, pz, dc, dz, max_inner, tpc, tpz, tdc, tdz);
}
__global__ void par2_b_finalize_apply_k(const double* part4, AdmmCtl* ctl, int max_inner, double tpc, double tpz,
                                        double tdc, double tdz) {
  if (ctl->active == 0 || threadIdx.x != 0) return;
  par2_b_decide(ctl, part4[0], part4[1], part4[2], part4[3], max_inner, tpc, tpz, tdc, tdz);
}
void par2_b_finalize(const double* norms, const P2Dims& d, int use_constr, AdmmCtl* ctl, int max_inner,
                     double tol_pr_coupl, double tol_pr_constr, double tol_du_coupl, double tol_du_constr, hipStream_t s,
                     double* part4, const P2AllReduce& allreduce) {
  par2_b_finalize_k<<<1, kKsumThreads, 0, s>>>(norms, d.K, d.k0, d.k1, use_constr, ctl, max_inner, tol_pr_coupl,
                                               tol_pr_constr, tol_du_coupl, tol_du_constr, part4);
  AO_KERNEL_CHECK();
  if (part4) {
    allreduce(part4, 4);
    par2_b_finalize_apply_k<<<1, 64, 0, s>>>(part4, ctl, max_inner, tol_pr_coupl, tol_pr_constr, tol_du_coupl,
                                             tol_du_constr);
    AO_KERNEL_CHECK();
  }
}

// ---------------------------------------------------------------------------
// C mode
// ---------------------------------------------------------------------------
__global__ void par2_c_system_k(const double* A, const double* T1, const double* GA, const double* GB, double w,
                                double ridge, double bsum_half, int nrho, int raw, const double* Madd, P2Dims d,
                                const double* Cfac, double* a, double* rho, double* L, AdmmCtl* ctl) {
  extern __shared__ double sh[];
  __shared__ double rk;
  const int k = d.k0 + blockIdx.x, R = d.R, I = d.I;
  for (int r = threadIdx.x; r < R; r += blockDim.x) {
    double acc = 0.0;
    for (int i = 0; i < I; ++i) acc += A[i + I * r] * T1[(int64_t)k * I * R + i + I * r];
    a[k + d.K * r] = w * acc + bsum_half * Cfac[k + d.K * r];          // w*diag(A' X_k B_k) (:221), bsum (:231)
  }
  for (int e = threadIdx.x; e < R * R; e += blockDim.x) sh[e] = GA[e] * GB[(int64_t)k * R * R + e];   // :222
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int r = 0; r < R; ++r) t += sh[r + R * r];
    rk = t / R;                                                        // :223
    rho[k] = rk;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < R * R; e += blockDim.x) {
    double b = w * sh[e];
    if (e % R == e / R) b += ridge + bsum_half + nrho * (rk / 2);   // :224-239 ; coupled: :262-264
    if (Madd) b += rk / 2 * Madd[e];                                 // coupling type 2: + rho_k/2 * H*H'  (:307)
    sh[e] = b;
  }
  __syncthreads();
  if (raw) {                                         // the (K*R)-system of coupling type 1 wants B_k itself (:286)
    for (int e = threadIdx.x; e < R * R; e += blockDim.x) L[(int64_t)k * R * R + e] = sh[e];
    return;
  }
  const bool ok = chol_lds(sh, R);
  if (ok)
    for (int e = threadIdx.x; e < R * R; e += blockDim.x) L[(int64_t)k * R * R + e] = sh[e];
  else if (threadIdx.x == 0 && ctl) ctl->notpd = 1;
}
__global__ __launch_bounds__(64) void par2_max_k(const double* x, int n, double* out, double* mean, double* sum) {
  double m = -INFINITY, t = 0.0;                                       // lane-strided, then a 64-lane butterfly
  for (int i = threadIdx.x; i < n; i += 64) { m = fmax(m, x[i]); t += x[i]; }
  for (int o = 32; o > 0; o >>= 1) { m = fmax(m, __shfl_xor(m, o, 64)); t += __shfl_xor(t, o, 64); }
  if (threadIdx.x == 0) {
    out[0] = m;                                                        // max(rho)  (:1424)
    if (sum) sum[0] = t;                                               // sum(rho)  (:736)
    if (mean) mean[0] = t / n;                                         // mean(rho) (:284, :712)
  }
}
void par2_c_system(const double* A, const double* T1, const double* GA, const double* GB, double w, double ridge,
                   double bsum_half, int nrho, int raw, const P2Dims& d, const double* Cfac, double* a, double* rho,
                   double* L, AdmmCtl* ctl, hipStream_t s, const double* Madd) {
  par2_c_system_k<<<d.k1 - d.k0, kP2Threads, (size_t)d.R * d.R * sizeof(double), s>>>(A, T1, GA, GB, w, ridge, bsum_half,
                                                                            nrho, raw, Madd, d, Cfac, a, rho, L, ctl);
  AO_KERNEL_CHECK();
}
void par2_rho_max(const double* rho, int K, double* rhomax, hipStream_t s, double* rhomean, double* rhosum) {
  par2_max_k<<<1, 64, 0, s>>>(rho, K, rhomax, rhomean, rhosum);
  AO_KERNEL_CHECK();
}

// M = blkdiag(B_1..B_K) + rhoC/2 * kron(H'H, I_R) (+ rhoC/2 * I if the mode is constrained), unknowns ordered as the
// rows of C back to back: index k*R + r   (cmtf_fun_AOADMM.m:283-293)
__global__ void par2_c_big_system_k(const double* Bk, const double* HtH, const double* rhoC, int constrained, int K,
                                    int R, double* M) {
  const int n = K * R;
  const double h = rhoC[0] / 2;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < (int64_t)n * n;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int u = (int)(e % n), v = (int)(e / n);
    const int k = u / R, r = u % R, k2 = v / R, q = v % R;
    double m = 0.0;
    if (k == k2) m = Bk[(int64_t)k * R * R + r + R * q];
    if (r == q) m += h * HtH[k + (int64_t)K * k2];
    if (constrained && u == v) m += h;
    M[e] = m;
  }
}
// diagonal H'H: L_k = chol(B_k + rhoC/2*(d_k + constrained)*I), in place over B_k
__global__ void par2_c_rowsys_diag_k(double* L, const double* d, const double* rhoC, int constrained, int R, AdmmCtl* ctl) {
  extern __shared__ double sh[];
  const int k = blockIdx.x;
  const double add = rhoC[0] / 2 * (d[k] + (constrained ? 1.0 : 0.0));
  for (int e = threadIdx.x; e < R * R; e += blockDim.x) sh[e] = L[(int64_t)k * R * R + e] + ((e % R == e / R) ? add : 0.0);
  __syncthreads();
  const bool ok = chol_lds(sh, R);
  if (ok)
    for (int e = threadIdx.x; e < R * R; e += blockDim.x) L[(int64_t)k * R * R + e] = sh[e];
  else if (threadIdx.x == 0 && ctl) ctl->notpd = 1;
}
void par2_c_rowsys_diag(double* L, const double* d, const double* rhoC, int constrained, int K, int R, AdmmCtl* ctl,
                        hipStream_t s) {
  par2_c_rowsys_diag_k<<<K, kP2Threads, (size_t)R * R * sizeof(double), s>>>(L, d, rhoC, constrained, R, ctl);
  AO_KERNEL_CHECK();
}
void par2_c_big_system(const double* Bk, const double* HtH, const double* rhoC, int constrained, int K, int R, double* M,
                       hipStream_t s) {
  const int64_t n2 = (int64_t)K * R * K * R;
  int64_t nb = cdiv(n2, 256);
  if (nb > 2048) nb = 2048;
  par2_c_big_system_k<<<(unsigned)nb, 256, 0, s>>>(Bk, HtH, rhoC, constrained, K, R, M);
  AO_KERNEL_CHECK();
}

template <int RMAX>
__global__ void par2_c_rowsolve_k(const double* a, const double* rho, const double* L, const double* Z,
                                  const double* mu, int use_admm, P2Dims d, double* Cfac, const AdmmCtl* ctl) {
  CTL_GUARD(ctl);
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= d.K) return;
  const int R = d.R, K = d.K;
  const double* Lk = L + (int64_t)k * R * R;
  double x[RMAX];                                      // registers for RMAX <= 16 (statically indexed)
#pragma unroll
  for (int r = 0; r < RMAX; ++r) {
    x[r] = 0.0;
    if (r < R) {
      double v = a[k + K * r];
      if (use_admm) v += rho[k] / 2 * (Z[k + K * r] - (mu ? mu[k + K * r] : 0.0));   // :604 ; coupled: Z holds the whole bracket (:640-643)
      x[r] = v;
    }
  }
#pragma unroll
  for (int r = 0; r < RMAX; ++r) {
    if (r < R) {
      double v = x[r];
#pragma unroll
      for (int q = 0; q < RMAX; ++q)
        if (q < r) v -= Lk[r + R * q] * x[q];
      x[r] = v / Lk[r + R * r];
    }
  }
#pragma unroll
  for (int r = RMAX - 1; r >= 0; --r) {
    if (r < R) {
      double v = x[r];
#pragma unroll
      for (int q = 0; q < RMAX; ++q)
        if (q > r && q < R) v -= Lk[q + R * r] * x[q];
      x[r] = v / Lk[r + R * r];
    }
  }
#pragma unroll
  for (int r = 0; r < RMAX; ++r)
    if (r < R) Cfac[k + K * r] = x[r];                                             // :236 / :605
}
void par2_c_rowsolve(const double* a, const double* rho, const double* L, const double* Z, const double* mu,
                     int use_admm, const P2Dims& d, double* Cfac, const AdmmCtl* ctl, hipStream_t s) {
  const unsigned nb = (unsigned)((d.K + 63) / 64);
  if (d.R <= 4) par2_c_rowsolve_k<4><<<nb, 64, 0, s>>>(a, rho, L, Z, mu, use_admm, d, Cfac, ctl);
  else if (d.R <= 8) par2_c_rowsolve_k<8><<<nb, 64, 0, s>>>(a, rho, L, Z, mu, use_admm, d, Cfac, ctl);
  else if (d.R <= 16) par2_c_rowsolve_k<16><<<nb, 64, 0, s>>>(a, rho, L, Z, mu, use_admm, d, Cfac, ctl);
  else par2_c_rowsolve_k<kMaxRank><<<nb, 64, 0, s>>>(a, rho, L, Z, mu, use_admm, d, Cfac, ctl);
  AO_KERNEL_CHECK();
}

// ---------------------------------------------------------------------------
// objective pieces
// ---------------------------------------------------------------------------
constexpr int kP2ResThreads = 256;        // I*J_k entries per slab: four waves
__global__ __launch_bounds__(kP2ResThreads) void par2_residual_k(const double* X, const double* A, const double* B,
                                                                  const double* Cfac, P2Dims d, double* res) {
  __shared__ double red[kP2ResThreads];
  const int k = d.k0 + blockIdx.x, R = d.R, I = d.I;
  const int64_t o = d.off[k];
  const int Jk = (int)(d.off[k + 1] - o);
  const double* Xk = X + (int64_t)I * o;
  const double* Bk = B + o * R;
  double acc = 0.0;
  // four entries per step, their tensor loads issued together (one entry per step waited for its own load every time)
  const int n = I * Jk, st = blockDim.x;
  int e = threadIdx.x;
  for (; e + 3 * st < n; e += 4 * st) {
    const double x0 = Xk[e], x1 = Xk[e + st], x2 = Xk[e + 2 * st], x3 = Xk[e + 3 * st];
    double m0 = 0.0, m1 = 0.0, m2 = 0.0, m3 = 0.0;
    const int i0 = e % I, j0 = e / I, i1 = (e + st) % I, j1 = (e + st) / I;
    const int i2 = (e + 2 * st) % I, j2 = (e + 2 * st) / I, i3 = (e + 3 * st) % I, j3 = (e + 3 * st) / I;
    for (int r = 0; r < R; ++r) {
      const double c = Cfac[k + d.K * r];
      m0 += A[i0 + I * r] * c * Bk[j0 + Jk * r]; m1 += A[i1 + I * r] * c * Bk[j1 + Jk * r];
      m2 += A[i2 + I * r] * c * Bk[j2 + Jk * r]; m3 += A[i3 + I * r] * c * Bk[j3 + Jk * r];
    }
    acc += (x0 - m0) * (x0 - m0); acc += (x1 - m1) * (x1 - m1); acc += (x2 - m2) * (x2 - m2); acc += (x3 - m3) * (x3 - m3);
  }
  for (; e < n; e += st) {
    const int i = e % I, j = e / I;
    double m = 0.0;
    for (int r = 0; r < R; ++r) m += A[i + I * r] * Cfac[k + d.K * r] * Bk[j + Jk * r];
    const double dlt = Xk[e] - m;
    acc += dlt * dlt;
  }
  acc = block_sum_pow2(acc, red);
  if (threadIdx.x == 0) res[k] = acc;
}
void par2_residual(const double* X, const double* A, const double* B, const double* Cfac, const P2Dims& d,
                   double* res, hipStream_t s) {
  par2_residual_k<<<d.k1 - d.k0, kP2ResThreads, 0, s>>>(X, A, B, Cfac, d, res);
  AO_KERNEL_CHECK();
}

__global__ __launch_bounds__(kP2Threads) void par2_b_gaps_k(const double* B, const double* P, const double* DeltaB,
                                                             const double* Z, P2Dims d, double* q) {
  __shared__ double red[kP2Threads];
  const int k = d.k0 + blockIdx.x, R = d.R;
  const int64_t o = d.off[k];
  const int Jk = (int)(d.off[k + 1] - o);
  const int64_t base = o * R;
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  const bool prev = k > 0 && (d.off[k] - d.off[k - 1]) == Jk;       // ||B_k - B_{k-1}||^2 (t_smoothness_penalty.m)
  const int64_t pbase = prev ? d.off[k - 1] * R : 0;
  for (int e = threadIdx.x; e < Jk * R; e += blockDim.x) {
    const int j = e % Jk, r = e / Jk;
    double pd = 0.0;
    for (int t = 0; t < R; ++t) pd += P[base + j + Jk * t] * DeltaB[t + R * r];
    const double b = B[base + e];
    s0 += (b - pd) * (b - pd);
    s1 += b * b;
    if (Z) s2 += (b - Z[base + e]) * (b - Z[base + e]);
    if (prev) s3 += (b - B[pbase + e]) * (b - B[pbase + e]);
  }
  s0 = block_sum_pow2(s0, red); s1 = block_sum_pow2(s1, red); s2 = block_sum_pow2(s2, red); s3 = block_sum_pow2(s3, red);
  if (threadIdx.x == 0) { q[(int64_t)k * 4] = s0; q[(int64_t)k * 4 + 1] = s1; q[(int64_t)k * 4 + 2] = s2; q[(int64_t)k * 4 + 3] = s3; }
}
// regv[k] = reg_func(B_k) for the regularisation-type constraints (constraints_to_prox.m:50,54,58,62,75,81;
// summed over the slabs at cmtf_fun_AOADMM.m:1279-1281).  One workgroup per slab, fixed summation order.
__global__ __launch_bounds__(kP2Threads) void par2_reg_k(const double* B, int type, double eta, P2Dims d, double* regv) {
  __shared__ double red[kP2Threads];
  const int k = d.k0 + blockIdx.x, R = d.R;
  const int64_t o = d.off[k];
  const int Jk = (int)(d.off[k + 1] - o);
  const double* Bk = B + o * R;
  double tot = 0.0;
  if (type == AOADMM_C_L2_REG) {                     // eta * sum_r ||B_k(:,r)||_2
    for (int r = 0; r < R; ++r) {
      double a = 0.0;
      for (int j = threadIdx.x; j < Jk; j += blockDim.x) { const double v = Bk[j + (int64_t)Jk * r]; a += v * v; }
      tot += sqrt(block_sum_pow2(a, red));
    }
  } else {
    double a = 0.0;
    for (int e = threadIdx.x; e < Jk * R; e += blockDim.x) {
      const int r = e / Jk, j = e - r * Jk;
      const double v = Bk[e];
      switch (type) {
        case AOADMM_C_L1_REG: a += fabs(v); break;
        case AOADMM_C_L0_REG: a += (v != 0.0) ? 1.0 : 0.0; break;
        case AOADMM_C_RIDGE: a += v * v; break;
        case AOADMM_C_TV: if (j + 1 < Jk) a += Bk[e + 1] - v; break;               // no abs(): quirk of :81
        case AOADMM_C_GL_SMOOTH: if (j + 1 < Jk) { const double dd = Bk[e + 1] - v; a += dd * dd; } break;
        default: break;
      }
    }
    tot = block_sum_pow2(a, red);
  }
  if (threadIdx.x == 0) regv[k] = eta * tot;
}
void par2_reg_values(const double* B, int type, double eta, const P2Dims& d, double* regv, hipStream_t s) {
  par2_reg_k<<<d.k1 - d.k0, kP2Threads, 0, s>>>(B, type, eta, d, regv);
  AO_KERNEL_CHECK();
}

__global__ void par2_collect_notpd_k(const AdmmCtl* a, const AdmmCtl* b, const AdmmCtl* c, double* out) {
  out[0] = (a->notpd || b->notpd || c->notpd) ? 1.0 : 0.0;
}
void par2_collect_notpd(const AdmmCtl* a, const AdmmCtl* b, const AdmmCtl* c, double* out, hipStream_t s) {
  par2_collect_notpd_k<<<1, 1, 0, s>>>(a, b, c, out);
  AO_KERNEL_CHECK();
}

void par2_b_gaps(const double* B, const double* P, const double* DeltaB, const double* Z, const P2Dims& d, double* q,
                 hipStream_t s) {
  par2_b_gaps_k<<<d.k1 - d.k0, kP2Threads, 0, s>>>(B, P, DeltaB, Z, d, q);
  AO_KERNEL_CHECK();
}

}  // namespace aoadmm
