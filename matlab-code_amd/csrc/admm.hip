// ADMM inner loop + proximal operators on gfx950.
// Reference: functions/cmtf_fun_AOADMM.m:591-623 (ADMM_constrained_only), :1420-1429
// (update_constraint), :1079-1096 (eval_res_ADMM_constr); functions/constraints_to_prox.m.
#include <type_traits>

#include "admm.h"
#include "device_utils.h"
#include "hosteig.h"
#include "loopctl.h"
#include "prox_dev.h"

namespace aoadmm {

#define CTL_GUARD(ctl) \
  if ((ctl) != nullptr && (ctl)->active == 0) return;

// ===========================================================================
// element-wise / row-wise prox (fusable)
// ===========================================================================
bool prox_is_fusable(int t) {
  return t == AOADMM_C_NONNEG || t == AOADMM_C_BOX || t == AOADMM_C_L1_REG || t == AOADMM_C_L0_REG ||
         t == AOADMM_C_RIDGE || t == AOADMM_C_SIMPLEX_ROW;
}

// ===========================================================================
// fused primal (+ dual) step, thread per row
// ===========================================================================
template <int RMAX>
__device__ __forceinline__ void row_solve_regs2(double (&x)[RMAX], const double* Lsh, int R) {
#pragma unroll
  for (int j = 0; j < RMAX; ++j) {
    if (j < R) {
      double v = x[j];
#pragma unroll
      for (int q = 0; q < j; ++q) v -= Lsh[j + R * q] * x[q];
      x[j] = v / Lsh[j + R * j];
    }
  }
#pragma unroll
  for (int j = RMAX - 1; j >= 0; --j) {
    if (j < R) {
      double v = x[j];
#pragma unroll
      for (int q = j + 1; q < RMAX; ++q)
        if (q < R) v -= Lsh[q + R * j] * x[q];
      x[j] = v / Lsh[j + R * j];
    }
  }
}

// ---------------------------------------------------------------------------
// Loop control without a kernel of its own.  The residuals of inner iteration k-1 are four sums that
// iteration k-1 leaves as per-block partials in part[(k-1)&1]; the first thing every block of iteration
// k's first kernel does is add them up (same data, same order => same decision in every block) and
// evaluate eval_res_ADMM_constr (:1079-1096) + the while condition (:600).  Block 0 records the
// outcome in ctl.  The later kernels of iteration k only look at ctl->active.
// ---------------------------------------------------------------------------
static constexpr int kRowThreads = 64;

struct FusedArgs {
  const double *A, *L, *rho, *Binv;
  double *fac, *Z, *mu, *V, *part;
  AdmmCtl* ctl;
  int64_t rows;
  int R, fused, ptype;
  int it, max_inner, nparts_prev;
  double p0, p1, tol_pr, tol_du;
};

// One inner iteration of ADMM_constrained_only for 64 rows per 256-thread block: wave w owns output
// columns [w*CPW, (w+1)*CPW) of those rows (lane = row), so the per-row solve fac = A_inner*inv(L*L')
// is spread over four waves and no thread carries more than R + 5*CPW operands in registers.
// A_inner (:608) is exchanged through LDS, inv(L*L') is broadcast from LDS.  With an element-/row-wise
// prox the kernel also does update_constraint (:1420-1429) and the residual partial sums; otherwise it
// writes fac and V = fac + mu for the column prox kernel.
// The kernel is latency-bound (2000 x 20 operands): every global load it needs -- operands, system
// matrix, the previous iteration's partial sums -- is issued in ONE round trip before the loop decision.
static constexpr int kRowBlock = 256;
template <int CPW, bool EXACT>
__global__ __launch_bounds__(kRowBlock) void admm_rows_k(FusedArgs a) {
  constexpr int RMAX = 4 * CPW;
  extern __shared__ double lds[];                   // Binv[R*R] | rhs[64][RP] | red[4][4]
  const int R = EXACT ? RMAX : a.R;
  const int RP = R | 1;
  double* Bsh = lds;
  double* rsh = lds + R * R;
  double* red = rsh + 64 * RP;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int64_t stride = (int64_t)gridDim.x * 64;
  int64_t i = (int64_t)blockIdx.x * 64 + lane;
  bool have = i < a.rows;
  // loads are unconditional on clamped (always valid) addresses: no exec-mask branches in the prologue,
  // values of padding lanes / padding columns are never stored
  double av[CPW], mu[CPW], zo[CPW];
#pragma unroll
  for (int k = 0; k < CPW; ++k) {
    const int c = w * CPW + k;
    const int64_t o = (have ? i : a.rows - 1) + a.rows * (c < R ? c : 0);
    zo[k] = a.Z[o]; mu[k] = a.mu[o]; av[k] = a.A[o];
  }
  constexpr int NL = (RMAX * RMAX + kRowBlock - 1) / kRowBlock;
  double lb[NL];
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const int e = t + kRowBlock * k;
    lb[k] = a.Binv[e < R * R ? e : 0];
  }
  const double rho = a.rho[0];
  const bool go = admm_continue(a.part + (int64_t)((a.it + 1) & 1) * kMaxParts * 4, a.nparts_prev, a.it, a.max_inner,
                                a.tol_pr, a.tol_du, a.ctl, blockIdx.x == 0 && t == 0);
  if (!go) return;                                  // block-uniform
#pragma unroll
  for (int k = 0; k < NL; ++k) {
    const int e = t + kRowBlock * k;
    if (e < R * R) Bsh[e] = lb[k];
  }
  const double rh = rho / 2;
  const ElemProx ep = elem_prox_of(a.ptype, a.p0, a.p1, rho);
  double s1 = 0, s2 = 0, s3 = 0, s4 = 0;
  for (;;) {
#pragma unroll
    for (int k = 0; k < CPW; ++k) {
      const int c = w * CPW + k;
      if (c < R) rsh[lane * RP + c] = av[k] + rh * (zo[k] - mu[k]);       // A_inner = A + rho/2*(Z - mu)   (:608)
    }
    __syncthreads();
    double rhs[RMAX];
#pragma unroll
    for (int q = 0; q < RMAX; ++q) rhs[q] = q < R ? rsh[lane * RP + q] : 0.0;
    double x[CPW];
#pragma unroll
    for (int k = 0; k < CPW; ++k) {                  // fac = A_inner * inv(L*L')       (:609)
      const int c = w * CPW + k;
      double acc = 0.0;
      if (c < R) {
#pragma unroll
        for (int q = 0; q < RMAX; ++q)
          if (q < R) acc += rhs[q] * Bsh[q + R * c];
      }
      x[k] = acc;
    }
    if (a.fused) {
      double z[CPW];
#pragma unroll
      for (int k = 0; k < CPW; ++k) z[k] = x[k] + mu[k];
      if (a.ptype == AOADMM_C_SIMPLEX_ROW) {
        // the projection needs the whole row: exchange fac + mu through LDS, every thread of a row finds
        // the same threshold (exact nested-active-set iteration) and applies it to its own columns
        __syncthreads();
#pragma unroll
        for (int k = 0; k < CPW; ++k) {
          const int c = w * CPW + k;
          if (c < R) rsh[lane * RP + c] = z[k];
        }
        __syncthreads();
        double v[RMAX];
#pragma unroll
        for (int q = 0; q < RMAX; ++q) v[q] = q < R ? rsh[lane * RP + q] : 0.0;
        simplex_regs<RMAX>(v, R, a.p0);
#pragma unroll
        for (int k = 0; k < CPW; ++k) {
          const int c = w * CPW + k;
          z[k] = 0.0;
#pragma unroll
          for (int q = 0; q < RMAX; ++q)
            if (q == c) z[k] = v[q];
        }
      } else {
#pragma unroll
        for (int k = 0; k < CPW; ++k) z[k] = elem_prox(ep, z[k]);
      }
      if (have) {
#pragma unroll
        for (int k = 0; k < CPW; ++k) {
          const int c = w * CPW + k;
          if (c < R) {
            const int64_t o = i + a.rows * c;
            const double mn = mu[k] + x[k] - z[k];     // mu = mu + fac - Z              (:1428)
            a.fac[o] = x[k];
            a.Z[o] = z[k];
            a.mu[o] = mn;
            const double d = x[k] - z[k];
            s1 += d * d;
            s2 += x[k] * x[k];
            s3 += mn * mn;
            const double e = z[k] - zo[k];
            s4 += e * e;
          }
        }
      }
    } else if (have) {
#pragma unroll
      for (int k = 0; k < CPW; ++k) {
        const int c = w * CPW + k;
        if (c < R) {
          const int64_t o = i + a.rows * c;
          a.fac[o] = x[k];
          a.V[o] = x[k] + mu[k];
        }
      }
    }
    i += stride;
    if (i - lane >= a.rows) break;                    // block-uniform
    have = i < a.rows;
#pragma unroll
    for (int k = 0; k < CPW; ++k) {
      const int c = w * CPW + k;
      const int64_t o = (have ? i : a.rows - 1) + a.rows * (c < R ? c : 0);
      zo[k] = a.Z[o]; mu[k] = a.mu[o]; av[k] = a.A[o];
    }
    __syncthreads();                                  // rsh is rewritten at the top of the loop
  }
  if (a.fused) {
    s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3); s4 = wave_sum(s4);   // fixed-order DPP tree
    if (lane == 0) { red[w * 4 + 0] = s1; red[w * 4 + 1] = s2; red[w * 4 + 2] = s3; red[w * 4 + 3] = s4; }
    __syncthreads();
    if (t < 4) {
      a.part[((int64_t)(a.it & 1) * kMaxParts + blockIdx.x) * 4 + t] = red[t] + red[4 + t] + red[8 + t] + red[12 + t];
    }
  }
}

// ---------------------------------------------------------------------------
// The whole inner loop of an element-wise-prox mode in TWO launches instead of MaxInnerIters, on the matrix cores.
// With an element-wise prox every row of (fac, Z, mu) evolves on its own; only the residual test of the while
// condition (:600) couples the rows.  Pass 1 (FINAL = false) runs all MaxInnerIters iterations for its 64 rows in
// registers and stores nothing but the four residual partial sums of every iteration.  Pass 2 (FINAL = true) adds
// the partial sums up (same data, same order in every block: same decision everywhere), finds the iteration k* after
// which the reference loop stops, repeats exactly k* iterations -- the same instructions on the same operands, hence
// the same bits -- and stores fac, Z, mu.  Block 0 records k* and the residuals of iteration k* in ctl.  Pass 2 also
// leaves what the Gram kernel would compute next: the block's partial Gram matrix F_b'F_b and the row-major copy of
// its rows of the factor (the caller adds the partials with atb_fin).
//
// The row solve fac = A_inner * inv(L*L') (:609) is computed transposed, fac' = Binv * A_inner' (Binv is symmetric),
// with v_mfma_f64_16x16x4_f64: a wave owns 16 rows; lane (r = l&15, q = l>>4) supplies B[k = 4s+q][n = r] =
// A_inner(row r, column 4s+q) and receives D[m = q+4i (+16mt)][n = r] = fac(row r, column 4(4mt+i)+q) -- the SAME
// columns {4s+q} it supplies.  So a lane keeps A, Z, mu, fac of its row for the columns c = q (mod 4) in registers
// for the whole loop and the element-wise prox and dual update are lane-local: no LDS exchange, no barrier inside
// an iteration (the VALU form staged A_inner through LDS and read 20 + 100 LDS words per thread and iteration:
// ~4 us per iteration at 2000 x 20 against ~0.4 us here).
struct SpecExtra {
  double* gram_ws;      // [gridDim.x][R*R] partial Gram matrices (FINAL only; may be null)
  double* At;           // rows x R row-major copy of fac (FINAL only; may be null)
};
typedef double f64x4_t __attribute__((ext_vector_type(4)));
static constexpr int kSpecMaxInner = 10;      // beyond this the work thrown away after an early exit could matter
static constexpr int kSpecThreads = 64;       // one wave = 16 rows per workgroup: no barrier anywhere in the kernel
static constexpr int kSpecPartJ = 4;          // a lane adds up to 4 partial sums per iteration: <= 256 workgroups
template <int KS, bool FINAL>
__global__ __launch_bounds__(kSpecThreads) void admm_rows_mfma_k(FusedArgs a, SpecExtra ex) {
  constexpr int MT = (KS + 3) / 4;
  extern __shared__ double rsh[];                   // FINAL: [16][RP] tile of the new factor for the Gram partial
  const int R = a.R;
  const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
  const int64_t i = (int64_t)blockIdx.x * 16 + r;
  const bool have = i < a.rows;
  const int nparts = gridDim.x;
  double av[KS], mu[KS], zo[KS], x[KS], bi[MT][KS];
  // every global load of the kernel in one round trip, on clamped (always valid) addresses
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int c = 4 * s + q;
    const int64_t o = (have ? i : a.rows - 1) + a.rows * (c < R ? c : 0);
    zo[s] = a.Z[o]; mu[s] = a.mu[o]; av[s] = a.A[o];
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int m = 16 * mt + r, k = 4 * s + q;
      bi[mt][s] = a.Binv[(m < R ? m : 0) + R * (k < R ? k : 0)];
    }
  const double rho = a.rho[0];
  int niter = a.max_inner;
  if (FINAL) {
    // k*: the first iteration whose residuals end the loop (eval_res_ADMM_constr :1079-1096, while condition :600).
    // Branch-free loads (clamped indices, masked adds) so that all of them are in flight together.
    double S[kSpecMaxInner][4];
#pragma unroll
    for (int it = 0; it < kSpecMaxInner; ++it) {
      const int itc = it < a.max_inner ? it : a.max_inner - 1;
      double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
      for (int j = 0; j < kSpecPartJ; ++j) {
        const int b = lane + 64 * j;
        const bool ok = b < nparts;
        const f64x4_t v = *reinterpret_cast<const f64x4_t*>(a.part + ((int64_t)itc * nparts + (ok ? b : 0)) * 4);
        s0 += ok ? v[0] : 0.0; s1 += ok ? v[1] : 0.0; s2 += ok ? v[2] : 0.0; s3 += ok ? v[3] : 0.0;
      }
      S[it][0] = s0; S[it][1] = s1; S[it][2] = s2; S[it][3] = s3;
    }
    double pr = 0.0, du = 0.0;
    int ks = a.max_inner;
    bool found = false;
#pragma unroll
    for (int it = 1; it <= kSpecMaxInner; ++it) {    // (no early break: with one the array S went to scratch memory)
      double s0 = S[it - 1][0], s1 = S[it - 1][1], s2 = S[it - 1][2], s3 = S[it - 1][3];
      s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3);   // the same total in every lane
      if (!found && it <= a.max_inner) {
        pr = sqrt(s0) / sqrt(s1);                                            // :1085
        const double sc = sqrt(s2);
        du = sc > 0 ? sqrt(s3) / sc : sqrt(s3);                              // :1087-1092
        if (!(it < a.max_inner && (pr > a.tol_pr || du > a.tol_du))) { ks = it; found = true; }
      }
    }
    niter = ks;
    if (blockIdx.x == 0 && lane == 0) {
      a.ctl->res[1] = pr;
      a.ctl->res[3] = du;
      a.ctl->iters = ks;
      a.ctl->active = 0;
    }
  }
  // padding columns (4s+q >= R) and padding rows of Binv carry zeros: they add nothing to the products
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    if (4 * s + q >= R) { av[s] = 0.0; zo[s] = 0.0; mu[s] = 0.0; }
    x[s] = 0.0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      if (16 * mt + r >= R || 4 * s + q >= R) bi[mt][s] = 0.0;
  }
  const double rh = rho / 2;
  const ElemProx ep = elem_prox_of(a.ptype, a.p0, a.p1, rho);
  for (int it = 0; it < niter; ++it) {
    double ain[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) ain[s] = av[s] + rh * (zo[s] - mu[s]);       // A_inner = A + rho/2*(Z - mu)   (:608)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {                                          // fac = A_inner * inv(L*L')       (:609)
      f64x4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(bi[mt][s], ain[s], acc, 0, 0, 0);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (4 * mt + e < KS) x[4 * mt + e] = acc[e];
    }
    double s1 = 0, s2 = 0, s3 = 0, s4 = 0;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const double z = elem_prox(ep, x[s] + mu[s]);                            // Z = prox(fac + mu)              (:1425)
      const double mn = mu[s] + x[s] - z;                                      // mu = mu + fac - Z               (:1428)
      if (have && 4 * s + q < R) {
        const double d = x[s] - z;
        s1 += d * d;
        s2 += x[s] * x[s];
        s3 += mn * mn;
        const double e = z - zo[s];
        s4 += e * e;
        mu[s] = mn;
        zo[s] = z;
      }
    }
    if (!FINAL) {
      s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3); s4 = wave_sum(s4);   // fixed-order DPP tree
      if (lane == 0) {
        f64x4_t v = {s1, s2, s3, s4};
        *reinterpret_cast<f64x4_t*>(a.part + ((int64_t)it * nparts + blockIdx.x) * 4) = v;
      }
    }
  }
  if (!FINAL) return;
  if (have) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int c = 4 * s + q;
      if (c < R) {
        const int64_t o = i + a.rows * c;
        a.fac[o] = x[s];
        a.Z[o] = zo[s];
        a.mu[o] = mu[s];
      }
    }
  }
  if (ex.gram_ws == nullptr) return;
  // Row-major copy of the wave's 16 rows and their partial Gram matrix G_b = F_b'F_b (what atb_part_k would do
  // next).  The tile goes through the wave's LDS (a wave's LDS operations complete in order: no barrier); the Gram
  // partial is MFMA work as well: D[m][n] = sum_k F(k, m) F(k, n) with the 16 rows as the reduction index.
  const int RP = R | 1;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int c = 4 * s + q;
    if (c < R) rsh[r * RP + c] = have ? x[s] : 0.0;
  }
  const int64_t r0 = (int64_t)blockIdx.x * 16;
  const int nr = (int)((a.rows - r0 < 16) ? (a.rows - r0) : 16);
  if (ex.At)
    for (int e = lane; e < nr * R; e += kSpecThreads) {
      const int ii = e / R, k = e - ii * R;
      ex.At[(r0 + ii) * R + k] = rsh[ii * RP + k];
    }
  double fo[MT][4];                                   // F(row 4s + q, column 16mt + r), s = 0..3
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int c = 16 * mt + r;
      fo[mt][s] = c < R ? rsh[(4 * s + q) * RP + c] : 0.0;
    }
  double* G = ex.gram_ws + (int64_t)blockIdx.x * R * R;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < MT; ++nt) {
      f64x4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fo[mt][s], fo[nt][s], acc, 0, 0, 0);
      const int n = 16 * nt + r;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = 16 * mt + q + 4 * e;
        if (m < R && n < R) G[m + R * n] = acc[e];
      }
    }
}

// (A one-launch form of this loop -- the workgroups meeting at a grid barrier instead of a kernel boundary, the result of
// pass 1 kept in registers when the loop ran to its cap -- was built and measured in round 3: 735 GPU tests green, but no
// faster: 1.279 against 1.274 ms per iteration at one rank's share of 8 GPUs, 7.92 against 7.94 ms at one GPU.  The
// workgroups sit on eight dies with separate L2s; an agent-scope barrier (release fence = L2 write-back, atomics and
// acquire loads that go past the L2) costs about what the kernel boundary costs, ~10 us.  Not kept.)
// Thread-per-row variant for the triangular-solve path (no explicit inverse available: op-level entry)
template <int RMAX>
__global__ __launch_bounds__(kRowThreads) void admm_rowL_k(FusedArgs a) {
  extern __shared__ double Lsh[];
  const int R = a.R;
  for (int e = threadIdx.x; e < R * R; e += kRowThreads) Lsh[e] = a.L[e];
  const double rho = a.rho[0];
  const bool go = admm_continue(a.part + (int64_t)((a.it + 1) & 1) * kMaxParts * 4, a.nparts_prev, a.it, a.max_inner,
                                a.tol_pr, a.tol_du, a.ctl, blockIdx.x == 0 && threadIdx.x == 0);
  if (!go) return;
  __syncthreads();
  const double rh = rho / 2;
  const ElemProx ep = elem_prox_of(a.ptype, a.p0, a.p1, rho);
  double s1 = 0, s2 = 0, s3 = 0, s4 = 0;
  for (int64_t i = (int64_t)blockIdx.x * kRowThreads + threadIdx.x; i < a.rows; i += (int64_t)gridDim.x * kRowThreads) {
    double x[RMAX], mu[RMAX], zo[RMAX];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
      x[r] = 0; mu[r] = 0; zo[r] = 0;
      if (r < R) {
        const int64_t o = i + a.rows * r;
        zo[r] = a.Z[o];
        mu[r] = a.mu[o];
        x[r] = a.A[o] + rh * (zo[r] - mu[r]);       // A_inner = A + rho/2*(Z - mu)   (:608)
      }
    }
    row_solve_regs2<RMAX>(x, Lsh, R);               // fac = (A_inner/L')/L            (:609)
    if (a.fused) {
      double z[RMAX];
#pragma unroll
      for (int r = 0; r < RMAX; ++r) z[r] = x[r] + mu[r];
      if (a.ptype == AOADMM_C_SIMPLEX_ROW) {
        simplex_regs<RMAX>(z, R, a.p0);
      } else {
#pragma unroll
        for (int r = 0; r < RMAX; ++r) z[r] = elem_prox(ep, z[r]);
      }
#pragma unroll
      for (int r = 0; r < RMAX; ++r) {
        if (r < R) {
          const int64_t o = i + a.rows * r;
          const double mn = mu[r] + x[r] - z[r];     // mu = mu + fac - Z              (:1428)
          a.fac[o] = x[r];
          a.Z[o] = z[r];
          a.mu[o] = mn;
          const double d = x[r] - z[r];
          s1 += d * d;
          s2 += x[r] * x[r];
          s3 += mn * mn;
          const double e = z[r] - zo[r];
          s4 += e * e;
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < RMAX; ++r) {
        if (r < R) {
          const int64_t o = i + a.rows * r;
          a.fac[o] = x[r];
          a.V[o] = x[r] + mu[r];
        }
      }
    }
  }
  if (a.fused) {
    s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3); s4 = wave_sum(s4);   // fixed-order DPP tree
    if (threadIdx.x == 0) {
      double* pb = a.part + ((int64_t)(a.it & 1) * kMaxParts + blockIdx.x) * 4;
      pb[0] = s1; pb[1] = s2; pb[2] = s3; pb[3] = s4;
    }
  }
}

int admm_partials(int64_t rows) {
  (void)rows;
  return 2 * kMaxParts;                             // both parities
}

// records the residuals / iteration count of the last executed inner iteration when the loop ran to
// MaxInnerIters (an earlier exit was recorded by the iteration that detected it).  The solver folds this
// into the Gram kernel that follows every mode update (atb_small); the kernel is for callers without one.
__global__ void admm_loop_end_k(LoopEnd le) { loop_end_eval(le); }

// element-wise dual update after a column-wise prox: mu += fac - Znew ; Z <- Znew ; partial norms
__global__ void dual_update_k(const double* fac, double* Z, double* mu, const double* Znew, int64_t n,
                              double* part, const AdmmCtl* ctl) {   // part: this iteration's parity block
  CTL_GUARD(ctl);
  __shared__ double red[4][4];
  double s1 = 0, s2 = 0, s3 = 0, s4 = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double x = fac[i], z = Znew[i], zo = Z[i];
    const double mn = mu[i] + x - z;
    mu[i] = mn;
    Z[i] = z;
    const double d = x - z, e = z - zo;
    s1 += d * d; s2 += x * x; s3 += mn * mn; s4 += e * e;
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3); s4 = wave_sum(s4);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][w] = s1; red[1][w] = s2; red[2][w] = s3; red[3][w] = s4; }
  __syncthreads();
  if (threadIdx.x < 4) {
    double t = 0;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += red[threadIdx.x][k];
    part[(int64_t)blockIdx.x * 4 + threadIdx.x] = t;
  }
}

// ===========================================================================
// column-wise prox kernels: one block per column
// ===========================================================================
__device__ __forceinline__ double block_sum(double v, double* sh) {
  // fixed-order tree over blockDim.x (power of two <= 256)
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int st = blockDim.x >> 1; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
    __syncthreads();
  }
  const double r = sh[0];
  __syncthreads();
  return r;
}

struct ColArgs {
  const double* V;
  double* Z;
  int64_t ldv, ldz, rows;
  int R, type;
  double p0, p1;
  const double* rho;
  double rho_mul;
  double* ws;
};

// l2-ball family, l2 regularisation, non-negative sphere (:37,:40,:43,:56)
__global__ void prox_colnorm_k(ColArgs a, const AdmmCtl* ctl) {
  CTL_GUARD(ctl);
  __shared__ double sh[256];
  __shared__ double shv[256];
  __shared__ int shi[256];
  const int r = blockIdx.x;
  const double* v = a.V + a.ldv * r;
  double* z = a.Z + a.ldz * r;
  const bool clamp = a.type == AOADMM_C_NONNEG_L2_BALL || a.type == AOADMM_C_NONNEG_L2_SPHERE;
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < a.rows; i += blockDim.x) {
    double y = v[i];
    if (clamp) y = fmax(y, 0.0);
    acc += y * y;
  }
  const double nrm = sqrt(block_sum(acc, sh));
  if (a.type == AOADMM_C_NONNEG_L2_SPHERE) {
    if (nrm == 0.0) {
      // [~,maxcoord] = max(X(:,r)) : first maximum (prox_normalized_nonneg.m:6-7)
      double bv = -INFINITY;
      int64_t bi = 0;
      for (int64_t i = threadIdx.x; i < a.rows; i += blockDim.x)
        if (v[i] > bv) { bv = v[i]; bi = i; }
      shv[threadIdx.x] = bv;
      shi[threadIdx.x] = (int)bi;
      __syncthreads();
      if (threadIdx.x == 0) {
        double best = shv[0];
        int64_t besti = shi[0];
        for (int t = 1; t < (int)blockDim.x; ++t)
          if (shv[t] > best || (shv[t] == best && shi[t] < besti)) { best = shv[t]; besti = shi[t]; }
        shi[0] = (int)besti;
      }
      __syncthreads();
      const int64_t arg = shi[0];
      for (int64_t i = threadIdx.x; i < a.rows; i += blockDim.x) z[i] = (i == arg) ? 1.0 : 0.0;
    } else {
      for (int64_t i = threadIdx.x; i < a.rows; i += blockDim.x) z[i] = fmax(v[i], 0.0) / nrm;
    }
    return;
  }
  double scale;
  if (a.type == AOADMM_C_L2_REG) {
    const double g = a.p0 / (a.rho[0] * a.rho_mul);
    scale = nrm > g ? 1.0 - g / nrm : 0.0;
  } else {
    scale = nrm > a.p0 ? a.p0 / nrm : 1.0;
  }
  for (int64_t i = threadIdx.x; i < a.rows; i += blockDim.x) {
    double y = v[i];
    if (clamp) y = fmax(y, 0.0);
    z[i] = y * scale;
  }
}

// simplex column-wise / l1-ball (:21,:34): parallel fixed-point for the exact threshold
__global__ void prox_simplex_col_k(ColArgs a, const AdmmCtl* ctl) {
  CTL_GUARD(ctl);
  __shared__ double sh[256];
  const int r = blockIdx.x;
  const double* v = a.V + a.ldv * r;
  double* z = a.Z + a.ldz * r;
  const bool l1 = a.type == AOADMM_C_L1_BALL;
  const double eta = a.p0;
  if (l1) {
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < a.rows; i += blockDim.x) acc += fabs(v[i]);
    const double n1 = block_sum(acc, sh);
    if (!(n1 > eta)) {
      for (int64_t i = threadIdx.x; i < a.rows; i += blockDim.x) z[i] = v[i];
      return;
    }
  }
  double tau = -INFINITY;
  double cnt_prev = -1.0;
  for (int64_t it = 0; it <= a.rows; ++it) {
    double s = 0.0, c = 0.0;
    for (int64_t i = threadIdx.x; i < a.rows; i += blockDim.x) {
      const double y = l1 ? fabs(v[i]) : v[i];
      if (y > tau) { s += y; c += 1.0; }
    }
    const double S = block_sum(s, sh);
    const double Cn = block_sum(c, sh);
    if (Cn == cnt_prev || Cn == 0.0) break;
    cnt_prev = Cn;
    tau = (S - eta) / Cn;
  }
  for (int64_t i = threadIdx.x; i < a.rows; i += blockDim.x) {
    if (l1) {
      const double w = fmax(fabs(v[i]) - tau, 0.0);
      z[i] = v[i] > 0 ? w : (v[i] < 0 ? -w : 0.0);
    } else {
      z[i] = fmax(v[i] - tau, 0.0);
    }
  }
}

// ---- sequential column algorithms (one thread per task, data in LDS or L2-resident scratch)

// Exact 1-D TV prox: L. Condat, IEEE SPL 20(11) 2013 (restated; TV_Condat_v2 at prox_TV.m:7
// returns the same unique minimiser).
__device__ void tv1d_condat_dev(const double* y, double* x, int64_t n, double lam) {
  if (n <= 0) return;
  if (!(lam > 0.0)) {
    for (int64_t i = 0; i < n; ++i) x[i] = y[i];
    return;
  }
  int64_t k = 0, k0 = 0, km = 0, kp = 0;
  double vmin = y[0] - lam, vmax = y[0] + lam, umin = lam, umax = -lam;
  for (;;) {
    if (k == n - 1) {
      if (umin < 0.0) {
        do { x[k0++] = vmin; } while (k0 <= km);
        k = km = kp = k0;
        vmin = y[k];
        umin = lam;
        umax = vmin + umin - vmax;
      } else if (umax > 0.0) {
        do { x[k0++] = vmax; } while (k0 <= kp);
        k = km = kp = k0;
        vmax = y[k];
        umax = -lam;
        umin = vmax + umax - vmin;
      } else {
        vmin += umin / (double)(k - k0 + 1);
        do { x[k0++] = vmin; } while (k0 <= k);
        return;
      }
    } else {
      umin += y[k + 1] - vmin;
      if (umin < -lam) {
        do { x[k0++] = vmin; } while (k0 <= km);
        k = km = kp = k0;
        vmin = y[k];
        vmax = vmin + 2.0 * lam;
        umin = lam;
        umax = -lam;
      } else {
        umax += y[k + 1] - vmax;
        if (umax > lam) {
          do { x[k0++] = vmax; } while (k0 <= kp);
          k = km = kp = k0;
          vmax = y[k];
          vmin = vmax - 2.0 * lam;
          umin = lam;
          umax = -lam;
        } else {
          ++k;
          if (umin >= lam) {
            km = k;
            vmin += (umin - lam) / (double)(km - k0 + 1);
            umin = lam;
          }
          if (umax <= -lam) {
            kp = k;
            vmax += (umax + lam) / (double)(kp - k0 + 1);
            umax = -lam;
          }
        }
      }
    }
  }
}

__global__ void prox_tv_k(ColArgs a, int use_lds, const AdmmCtl* ctl) {
  CTL_GUARD(ctl);
  extern __shared__ double dyn[];
  const int r = blockIdx.x;
  const double* v = a.V + a.ldv * r;
  double* z = a.Z + a.ldz * r;
  const double lam = a.p0 / (a.rho[0] * a.rho_mul);     // prox_TV(x, eta/rho)  (:80)
  if (use_lds) {
    double* y = dyn;
    double* x = dyn + a.rows;
    for (int64_t i = threadIdx.x; i < a.rows; i += blockDim.x) y[i] = v[i];
    __syncthreads();
    if (threadIdx.x == 0) tv1d_condat_dev(y, x, a.rows, lam);
    __syncthreads();
    for (int64_t i = threadIdx.x; i < a.rows; i += blockDim.x) z[i] = x[i];
  } else {
    if (threadIdx.x == 0) tv1d_condat_dev(v, z, a.rows, lam);
  }
}

// Parallel exact TV prox for one column per 256-thread block (active-set / split-merge form of the
// same optimality system Condat's scan solves).  With u_i = sum_{t<=i}(y_t - x_t) the minimiser of
// 0.5||x-y||^2 + lam*sum|x_{i+1}-x_i| is characterised by |u_i| <= lam, u_{n-1} = 0 and
// u_i = -lam*sign(x_{i+1}-x_i) at every jump.  A guess of the jump set J (signs in {-1,0,+1}) fixes the
// segment values  v_S = (sum_S y + lam*(J[b] - J[a-1])) / |S| ; jumps whose sign disagrees with
// v_right - v_left are merged, segments whose interior |u| exceeds lam are split at the worst point.
// The fixed point satisfies the KKT system, i.e. is the unique minimiser; if the iteration cap is hit
// thread 0 falls back to the sequential scan, so the result is exact either way.  Inside the ADMM loop
// J is warm-started from the previous Z column, which usually converges in 1-3 rounds.
static constexpr int kTvParMax = 6400;       // rows of a column whose working arrays fit LDS (25 bytes per row, below)

// The iteration is built on prefix sums so that no step is sequential in a
// segment's length: with Pc[i] = sum_{t<i} (y_t - c) (c = mean(y), which keeps the prefix sums small) the value
// of segment [sa, sb] is  c + (Pc[sb+1] - Pc[sa] + lam*(J[sb] - J[sa-1])) / len  and the dual at an interior
// point is  u_i = -lam*J[sa-1] + (Pc[i+1] - Pc[sa]) - (v - c)*(i - sa + 1),  both O(1).  A round is: scan the
// segment starts, one thread per segment for the values, one per boundary for the merge test, every thread
// walks its own 8-16 entries for the split test and the per-segment worst violation is found with a 64-bit
// LDS max (magnitude bits above, index in the low 12 bits).  ~8 barriers per round, 1-3 rounds warm-started.
// With `fz` the kernel also does the dual update of the ADMM iteration for its column (mu = V - Z_new,
// Z <- Z_new) and the four residual sums, which removes the separate dual kernel.
struct TvFused {
  double* Z = nullptr;      // in: previous Z (warm start), out: new Z
  double* mu = nullptr;     // in/out
  double* part = nullptr;   // [R][4]: ||fac-Z||^2, ||fac||^2, ||mu||^2, ||Z-Zold||^2 of this column
  int64_t ld = 0;
};
// kTvThreads threads per column (256 for short columns, 1024 above 1024 rows: every phase walks a thread's chunk of
// ceil(n / kTvThreads) entries sequentially through LDS, so four times the threads is close to four times fewer
// dependent LDS round trips; measured at 2000 rows in DESIGN.md section 4.4).
// GLOBAL: columns beyond kTvParMax rows keep the same working arrays in a per-column slice of the prox workspace
// (L2-resident: 37 bytes per row) instead of LDS, so a long mode does not drop to the one-thread scan; the index field
// of the split key widens to 24 bits.
static __host__ __device__ inline size_t tv_ws_doubles(int64_t rows) {
  return ((size_t)(4 * rows + 2) * 8 + (size_t)(rows + 2) * 4 + (size_t)rows + 64 + 7) / 8;
}
template <int kTvThreads, int MEM = 0>
__global__ __launch_bounds__(kTvThreads) void prox_tv_fast_k(ColArgs a, const double* warm, int64_t ldw, TvFused fz,
                                                             const AdmmCtl* ctl) {
  CTL_GUARD(ctl);
  constexpr int NW = kTvThreads / 64;
  constexpr bool GLOBAL = MEM != 0;                                // columns beyond the LDS-resident 4096 rows
  // split key: magnitude bits of the worst |u| above, entry index below.  LDS-resident columns: 32 bits (float
  // magnitude, 13-bit index -- the key only ranks violations, any violating entry is a valid split point); workspace
  // forms: 64 bits with a 24-bit index
  typedef typename std::conditional<GLOBAL, unsigned long long, unsigned>::type KeyT;
  constexpr KeyT kIdxMask = GLOBAL ? (KeyT)0xffffffull : (KeyT)0x1fffu;
  extern __shared__ double lds_dyn[];
  __shared__ int wsum[NW];
  __shared__ double dsum[NW];
  __shared__ int flag_merge[2], flag_split[2];          // alternate by round: the reset of one never races with the read of the other
  const int n = (int)a.rows;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int r = blockIdx.x;
  const double* vin = a.V + a.ldv * r;
  double* z = a.Z + a.ldz * r;
  const double lam = a.p0 / (a.rho[0] * a.rho_mul);
  // working arrays: y (n), Pc (n + 1), val (n), best (n), start (n + 1 ints), J (n bytes) -- 37 bytes per row.
  //   MEM 0: all in LDS (<= 6400 rows) at 25 bytes per row: the input is staged in the slots of Pc (the prefix sums
  //          replace it in place, every thread its own chunk; the dual update takes the input from registers and the
  //          sequential fallback reads it from global memory again) and the split keys are 32 bits wide.  Round 3: the
  //          limit was 4096 rows at 37 bytes, and 6000 rows ran the hybrid form at 51 us per in-loop call.
  //   MEM 2: the arrays every phase of a round walks (Pc, start, J: 13 bytes per row) in LDS, y / val / best in the
  //          column's slice of the prox workspace (<= kTvHybridMax rows; round 2 had all six in the workspace there and
  //          every phase was a chain of L2 round trips: 90 us per in-loop call at 6000 rows against 22 us at 2000)
  //   MEM 1: all in the workspace (L2-resident)
  double* gws = GLOBAL ? a.ws + (size_t)blockIdx.x * tv_ws_doubles(a.rows) : nullptr;
  double *y, *Pc, *val;
  KeyT* best;
  int* start;
  signed char* J;
  if constexpr (MEM == 2) {
    y = gws; val = gws + n; best = reinterpret_cast<KeyT*>(gws + 2 * n);
    Pc = lds_dyn;                                                  // n + 1
    start = reinterpret_cast<int*>(lds_dyn + n + 1);               // n + 1
    J = reinterpret_cast<signed char*>(start + n + 1);             // n
  } else if constexpr (MEM == 0) {
    Pc = lds_dyn;                                                  // n + 1
    y = Pc;                                                        // the input, until the prefix sums take its place
    val = lds_dyn + n + 1;                                         // n
    best = reinterpret_cast<KeyT*>(val + n);                       // n (32-bit keys)
    start = reinterpret_cast<int*>(best + n);                      // n + 1
    J = reinterpret_cast<signed char*>(start + n + 1);             // n
  } else {
    double* dyn = gws;
    y = dyn;                                                       // n
    Pc = dyn + n;                                                  // n + 1
    val = dyn + 2 * n + 1;                                         // n
    best = reinterpret_cast<KeyT*>(dyn + 3 * n + 1);               // n
    start = reinterpret_cast<int*>(dyn + 4 * n + 1);               // n + 1
    J = reinterpret_cast<signed char*>(start + n + 1);             // n
  }
  const int chunk = (n + kTvThreads - 1) / kTvThreads;
  const int c0 = min(n, t * chunk), c1 = min(n, c0 + chunk);
  constexpr int kKeep = GLOBAL ? 1 : 8;                            // entries per thread of an LDS-resident column (launcher: rows <= 8 * threads)
  double keep_mu[kKeep], keep_z[kKeep];                            // old mu / old Z of the entries t + k*kTvThreads (fz only)
  double keep_y[kKeep];                                            // MEM 0: the input of the same entries (its LDS slots become Pc)
  // ---- load (coalesced, all loads of a thread independent: a chunk-ordered loop would serialise 8-16 memory
  // round trips), mean, centred prefix sums.  The warm-start column is staged in `val` the same way.
  {
    // every load of the thread is issued before the first LDS store, on clamped (always valid) addresses and without
    // an exec-mask branch in between: written as `for (i = t; i < n; i += 256) y[i] = vin[i]` the compiler emitted one
    // s_waitcnt vmcnt(0) per element, i.e. 8-16 dependent memory round trips before the kernel could start
    const double* wv = warm ? warm + ldw * r : vin;      // no warm start: a second read of the column, discarded
    // With `fz` the dual update at the end needs the old mu and the old Z of the same entries (t + k*kTvThreads): they
    // are fetched here, in the kernel's first round trip, and wait in registers (fz.Z is the warm-start column) -- read
    // at the end they were one more dependent memory round trip per call.
    const double* mv = fz.mu ? fz.mu + fz.ld * r : vin;
    const double* zv = fz.Z ? fz.Z + fz.ld * r : vin;    // (the same column as `wv` in the ADMM loop: the load hits the same lines)
    auto stage = [&](auto kper_tag) {
      constexpr int KP = decltype(kper_tag)::value;
      double ry[KP], rw[KP];
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        const int i = min(t + k * kTvThreads, n - 1);
        ry[k] = vin[i];
        rw[k] = wv[i];
        if (k < kKeep) { keep_mu[k] = mv[i]; keep_z[k] = zv[i]; keep_y[k] = ry[k]; }
      }
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        const int i = t + k * kTvThreads;
        if (i < n) { y[i] = ry[k]; val[i] = rw[k]; }
      }
    };
    if constexpr (GLOBAL) {
      for (int i0 = 0; i0 < n; i0 += 8 * kTvThreads) {  // eight independent loads of each column per round trip
        double ry[8], rw[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int i = min(i0 + t + k * kTvThreads, n - 1);
          ry[k] = vin[i];
          rw[k] = wv[i];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int i = i0 + t + k * kTvThreads;
          if (i < n) { y[i] = ry[k]; val[i] = rw[k]; }
        }
      }
    } else if (n <= kTvThreads) stage(std::integral_constant<int, 1>());
    else if (n <= 2 * kTvThreads) stage(std::integral_constant<int, 2>());
    else if (n <= 4 * kTvThreads) stage(std::integral_constant<int, 4>());
    else stage(std::integral_constant<int, 8>());
  }
  __syncthreads();
  double loc = 0.0;
  for (int i = c0; i < c1; ++i) loc += y[i];
  auto block_scan_d = [&](double v, double& total) {               // exclusive scan over threads, fixed order
    double inc = v;
    inc = wave_scan_incl(inc);
    if (lane == 63) dsum[w] = inc;
    __syncthreads();
    double base = 0.0, all = 0.0;
#pragma unroll
    for (int q = 0; q < NW; ++q) { if (q < w) base += dsum[q]; all += dsum[q]; }
    total = all;
    __syncthreads();
    return base + inc - v;
  };
  // ONE scan gives the mean and the centred prefix sums: the exclusive prefix of the plain sums at the thread's first
  // entry, less c times the number of entries before it, is the prefix of the centred values (the difference of the two
  // forms is a rounding of size |prefix| * 2^-53, far below the 1e-13 slack of `thr`; two scans were two more barriers)
  const double* sol = val;                                          // where the solution ends up (val, or Pc after the expansion)
  double tot;
  const double before = block_scan_d(loc, tot);
  const double c = n > 0 ? tot / n : 0.0;
  if (!(lam > 0.0)) {
    for (int i = c0; i < c1; ++i) val[i] = y[i];
  } else {
    double run = before - c * (double)c0;
    for (int i = c0; i < c1; ++i) { const double yi = y[i]; Pc[i] = run; run += yi - c; }   // (MEM 0: y[i] IS Pc[i])
    if (c1 == n && c0 < n) Pc[n] = run;                            // the thread holding the last entry
    // ---- warm start of the jump set
    if (warm) {
      for (int i = c0; i < c1; ++i) {
        if (i < n - 1) { const double d = val[i + 1] - val[i]; J[i] = d > 0 ? 1 : (d < 0 ? -1 : 0); }
        else J[i] = 0;
      }
    } else {
      for (int i = c0; i < c1; ++i) J[i] = 0;
    }
    __syncthreads();
    const double thr = lam * (1.0 + 1e-11) + 1e-13 * (fabs(c) + 1.0);
    const int max_rounds = 4 * n + 64;
    bool converged = false;
    for (int round = 0; round < max_rounds; ++round) {
      // 1. segment starts: i == 0 or a jump between i-1 and i
      int cnt = 0;
      for (int i = c0; i < c1; ++i) cnt += (i == 0 || J[i - 1] != 0) ? 1 : 0;
      int inc = cnt;
      inc = wave_scan_incl(inc);
      if (lane == 63) wsum[w] = inc;
      const int fp = round & 1;
      if (t == 0) { flag_merge[fp] = 0; flag_split[fp] = 0; }
      __syncthreads();
      int base = 0;
      for (int q = 0; q < w; ++q) base += wsum[q];
      const int first_seg = base + inc - cnt;                      // starts before this thread's chunk
      {
        int pos = first_seg;
        for (int i = c0; i < c1; ++i)
          if (i == 0 || J[i - 1] != 0) start[pos++] = i;
      }
      int nseg = 0;
#pragma unroll
      for (int q = 0; q < NW; ++q) nseg += wsum[q];
      if (t == 0) start[nseg] = n;
      __syncthreads();
      // 2. segment values
      for (int sgi = t; sgi < nseg; sgi += kTvThreads) {
        const int sa = start[sgi], sb = start[sgi + 1] - 1;
        const double sl = sa == 0 ? 0.0 : (double)J[sa - 1];
        const double sr = sb == n - 1 ? 0.0 : (double)J[sb];
        val[sgi] = c + (Pc[sb + 1] - Pc[sa] + lam * (sr - sl)) / (double)(sb - sa + 1);
        best[sgi] = (KeyT)0;
      }
      __syncthreads();
      // 3. merge jumps whose sign disagrees with the values on both sides
      for (int sgi = t; sgi + 1 < nseg; sgi += kTvThreads) {
        const int jp = start[sgi + 1] - 1;
        if ((double)J[jp] * (val[sgi + 1] - val[sgi]) <= 0.0) { J[jp] = 0; flag_merge[fp] = 1; }
      }
      __syncthreads();
      const int merged = flag_merge[fp];                           // (reset again two rounds on, behind several barriers)
      // 4. split segments whose interior dual leaves [-lam, lam]: per-segment worst violation.  In the same round as the
      // merges, for the segments no merge touched (a boundary whose J is zero now was merged: the segment's value is
      // stale): a round that only merged followed by a round that only split were two rounds (3.2 us each at 2000 rows)
      // for what is one change of the jump set between two ADMM iterations -- typically 3 rounds per call before, 2 now.
      {
        int sgi = first_seg - ((c0 < c1 && (c0 == 0 || J[c0 - 1] != 0)) ? 0 : 1);   // segment of entry c0
        for (int i = c0; i < c1; ++i) {
          if (i != c0 && J[i - 1] != 0) ++sgi;
          const int sa = start[sgi], sb = start[sgi + 1] - 1;
          if (i >= sb) continue;                                   // the segment's last entry carries the jump itself
          if (merged && ((sa > 0 && J[sa - 1] == 0) || (sb < n - 1 && J[sb] == 0))) continue;
          const double u0 = sa == 0 ? 0.0 : -lam * (double)J[sa - 1];
          const double u = u0 + (Pc[i + 1] - Pc[sa]) - (val[sgi] - c) * (double)(i - sa + 1);
          const double au = fabs(u);
          if (au > thr) {
            KeyT key;
            if constexpr (GLOBAL) key = ((KeyT)__double_as_longlong(au) & ~kIdxMask) | (KeyT)i;
            else key = ((KeyT)__float_as_uint((float)au) & ~kIdxMask) | (KeyT)i;       // au > thr >= 1e-13: never a zero key
            atomicMax(&best[sgi], key);
          }
        }
      }
      __syncthreads();
      for (int sgi = t; sgi < nseg; sgi += kTvThreads) {
        const KeyT b = best[sgi];
        if (b != (KeyT)0) {
          const int i = (int)(b & kIdxMask);
          const int sa = start[sgi];
          const double u0 = sa == 0 ? 0.0 : -lam * (double)J[sa - 1];
          const double u = u0 + (Pc[i + 1] - Pc[sa]) - (val[sgi] - c) * (double)(i - sa + 1);
          J[i] = u > 0 ? -1 : 1;
          flag_split[fp] = 1;
        }
      }
      __syncthreads();
      const int split = flag_split[fp];
      if (!split && !merged) { converged = true; break; }
    }
    if (converged) {
      // expand: entry i takes the value of its segment (val is indexed by segment; write through `best` as doubles
      // would alias, so expand into y, which is no longer needed)
      int sgi = 0;
      {
        int cnt = 0;
        for (int i = c0; i < c1; ++i) cnt += (i == 0 || J[i - 1] != 0) ? 1 : 0;
        int inc = cnt;
        inc = wave_scan_incl(inc);
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        int base = 0;
        for (int q = 0; q < w; ++q) base += wsum[q];
        sgi = base + inc - cnt - ((c0 < c1 && (c0 == 0 || J[c0 - 1] != 0)) ? 0 : 1);
      }
      for (int i = c0; i < c1; ++i) {
        if (i != c0 && J[i - 1] != 0) ++sgi;
        Pc[i] = val[sgi];                                          // Pc is free now: holds the solution
      }
      sol = Pc;
    } else {
      __syncthreads();
      if (t == 0) tv1d_condat_dev(MEM == 0 ? vin : y, val, n, lam);   // exact sequential fallback (MEM 0: the LDS copy of the input is gone)
    }
  }
  __syncthreads();
  // ---- write the column; optionally the ADMM dual update and residual sums for it
  if (fz.Z == nullptr) {
    for (int i = t; i < n; i += kTvThreads) z[i] = sol[i];
    return;
  }
  double* Zc = fz.Z + fz.ld * r;
  double* muc = fz.mu + fz.ld * r;
  double s1 = 0, s2 = 0, s3 = 0, s4 = 0;
  auto dual = [&](auto kper_tag) {
    constexpr int KP = decltype(kper_tag)::value;      // entries per thread, by the same size classes as above
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      const int i = t + k * kTvThreads;
      if (i < n) {
        const double zn = sol[i], vv = MEM == 0 ? keep_y[k < kKeep ? k : 0] : y[i];
        const double mo = keep_mu[k < kKeep ? k : 0], zo = keep_z[k < kKeep ? k : 0];   // fetched with the column
        const double x = vv - mo;                      // fac = V - mu_old
        const double mn = vv - zn;                     // mu + fac - Z   (:1428)
        Zc[i] = zn;
        muc[i] = mn;
        const double d = x - zn, e = zn - zo;
        s1 += d * d; s2 += x * x; s3 += mn * mn; s4 += e * e;
      }
    }
  };
  if constexpr (GLOBAL) {
    for (int i0 = 0; i0 < n; i0 += 4 * kTvThreads) {
      double mo[4], zo[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = min(i0 + t + k * kTvThreads, n - 1);
        mo[k] = muc[i];
        zo[k] = Zc[i];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + t + k * kTvThreads;
        if (i < n) {
          const double zn = sol[i], vv = y[i];
          const double x = vv - mo[k];
          const double mn = vv - zn;
          Zc[i] = zn;
          muc[i] = mn;
          const double d = x - zn, e = zn - zo[k];
          s1 += d * d; s2 += x * x; s3 += mn * mn; s4 += e * e;
        }
      }
    }
  } else if (n <= kTvThreads) dual(std::integral_constant<int, 1>());
  else if (n <= 2 * kTvThreads) dual(std::integral_constant<int, 2>());
  else if (n <= 4 * kTvThreads) dual(std::integral_constant<int, 4>());
  else dual(std::integral_constant<int, 8>());
  // the four residual sums of the column in one reduction (one barrier instead of eight)
  __shared__ double q4sum[NW][4];
  double q4[4] = {s1, s2, s3, s4};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    double v = q4[q];
    v = wave_sum(v);
    if (lane == 0) q4sum[w][q] = v;
  }
  __syncthreads();
  if (t < 4) {
    double tot4 = 0.0;
#pragma unroll
    for (int k = 0; k < NW; ++k) tot4 += q4sum[k][t];
    fz.part[(int64_t)r * 4 + t] = tot4;
  }
}

// isotonic / unimodal projections and the parallel GL solve live in iso.hip

// GL smoothness on columns longer than the cyclic-reduction kernel of iso.hip takes: one thread per column
// GL smoothness: (2*eta/rho*L + I) \ x with the path-graph Laplacian (:68-76): Thomas algorithm
__global__ void prox_gl_k(ColArgs a, const AdmmCtl* ctl) {
  CTL_GUARD(ctl);
  if (threadIdx.x != 0) return;
  const int r = blockIdx.x;
  const int64_t n = a.rows;
  const double* v = a.V + a.ldv * r;
  double* z = a.Z + a.ldz * r;
  double* cp = a.ws + (int64_t)r * n;
  const double s2 = 2.0 * (a.p0 / (a.rho[0] * a.rho_mul));
  if (n == 1) { z[0] = v[0] / (s2 * 1.0 + 1.0); return; }
  const double off = -s2;
  double diag = s2 * 1.0 + 1.0;
  cp[0] = off / diag;
  z[0] = v[0] / diag;
  for (int64_t i = 1; i < n; ++i) {
    const double d = ((i == n - 1) ? s2 * 1.0 : s2 * 2.0) + 1.0;
    const double m = d - off * cp[i - 1];
    cp[i] = off / m;
    z[i] = (v[i] - off * z[i - 1]) / m;
  }
  for (int64_t i = n - 2; i >= 0; --i) z[i] -= cp[i] * z[i + 1];
}

// orthonormal columns: U*V' of the thin SVD W = U S V' (project_ortho.m:3-4), i.e. the polar factor of W.
// One-sided Jacobi preconditioned through the Gram matrix: a one-sided Jacobi SVD looks for an orthogonal J with W*J =
// U*S (orthogonal columns); instead of rotating the rows x R matrix pair by pair (190 pairs x ~10 sweeps x a pass
// over 2000 rows each: 7.7 ms at 2000 x 20 in one workgroup), pass 1 takes J1 from the eigenvectors of G = W'W
// (R x R, one wave: sym_eig_small) and forms W1 = W*J1 with one small GEMM; pass 2 and 3 repeat that on W1, W2, whose
// Gram matrices are already diagonal up to rounding -- Jacobi on such a matrix is accurate in the relative sense, so the
// cond(W)^2 error of a single Gram-eigen step is removed.  Then S = column norms (the last eigenvalues), U = W3/S and
// Z = U*(J1*J2*J3)'.  Columns with S = 0 give zero columns, as before.
__global__ void ortho_m_k(double* M, const double* Jt, const double* G, int R, const AdmmCtl* ctl) {
  CTL_GUARD(ctl);                                      // M(r,c) = Jt(c,r) / S_r with S_r^2 = G(r,r)
  for (int e = threadIdx.x; e < R * R; e += blockDim.x) {
    const int r = e % R, c = e / R;
    const double g = G[r + R * r];
    const double sg = g > 0.0 ? sqrt(g) : 0.0;
    M[r + R * c] = sg > 0.0 ? Jt[c + R * r] / sg : 0.0;
  }
}
static size_t ortho_ws_doubles(int64_t rows, int R) {
  return (size_t)2 * rows * R + atb_ws_bytes(rows, R, R) / sizeof(double) + (size_t)5 * R * R + 64;
}
static void prox_ortho(const double* V, int64_t ldv, double* Z, int64_t ldz, int64_t rows, int R, double* ws, const AdmmCtl* ctl,
                       hipStream_t s) {
  double* Wa = ws;
  double* Wb = Wa + rows * R;
  double* aws = Wb + rows * R;
  double* G = aws + atb_ws_bytes(rows, R, R) / sizeof(double);
  double* Q = G + R * R;
  double* Ja = Q + R * R;
  double* Jb = Ja + R * R;
  double* M = Jb + R * R;
  double* wv = M;                                      // eigenvalues are not needed: M's storage takes them
  const double* W = V;
  int64_t ldw = ldv;
  const double* Jt = nullptr;                          // accumulated rotation J1*J2*...
  for (int pass = 0; pass < 3; ++pass) {
    atb_small(G, W, ldw, W, ldw, rows, R, R, aws, ctl, s);
    sym_eig_small(G, R, wv, Q, s, ctl);
    double* Wn = (W == Wa) ? Wb : Wa;
    gemm_small(Wn, rows, W, ldw, Q, R, rows, R, R, 0, coef(1.0), 0.0, ctl, s);          // W <- W*Q
    double* Jn = (Jt == Ja) ? Jb : Ja;
    if (pass == 0) AO_HIP(hipMemcpyAsync(Jn, Q, (size_t)R * R * sizeof(double), hipMemcpyDeviceToDevice, s));
    else gemm_small(Jn, R, Jt, R, Q, R, R, R, R, 0, coef(1.0), 0.0, ctl, s);            // J <- J*Q
    Jt = Jn;
    W = Wn; ldw = rows;
  }
  atb_small(G, W, ldw, W, ldw, rows, R, R, aws, ctl, s);                                 // squared column norms on its diagonal
  ortho_m_k<<<1, 256, 0, s>>>(M, Jt, G, R, ctl);
  AO_KERNEL_CHECK();
  gemm_small(Z, ldz, W, ldw, M, R, rows, R, R, 0, coef(1.0), 0.0, ctl, s);               // Z = (W/S)*J'
}

// stand-alone element-wise / row-wise prox (op-level entry and the generic loops)
template <int RMAX>
__global__ void prox_rows_k(ColArgs a, const AdmmCtl* ctl) {
  CTL_GUARD(ctl);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.rows) return;
  const double rho = a.rho[0] * a.rho_mul;
  double z[RMAX];
#pragma unroll
  for (int r = 0; r < RMAX; ++r) z[r] = r < a.R ? a.V[i + a.ldv * r] : 0.0;
  if (a.type == AOADMM_C_SIMPLEX_ROW) {
    simplex_regs<RMAX>(z, a.R, a.p0);
  } else {
#pragma unroll
    for (int r = 0; r < RMAX; ++r) z[r] = prox_elem(a.type, z[r], a.p0, a.p1, rho);
  }
#pragma unroll
  for (int r = 0; r < RMAX; ++r)
    if (r < a.R) a.Z[i + a.ldz * r] = z[r];
}

// ---- 'quadratic regularization' ------------------------------------------------------------------
void QuadPrep::build(const double* L_host, int64_t rows, hipStream_t s) {
  AO_REQUIRE(L_host != nullptr && rows > 0, "quadratic regularization needs its matrix L");
  if (rows > 4096) throw Error(AOADMM_ERR_UNSUPPORTED, "quadratic regularization: matrices beyond 4096 x 4096 are not prepared on the device");
  n = rows;
  std::vector<double> A(L_host, L_host + (size_t)rows * rows), wv, Uv;
  double asym = 0.0, nrm = 0.0;
  for (int64_t j = 0; j < rows; ++j)
    for (int64_t i = 0; i < rows; ++i) {
      const double d = A[(size_t)i + (size_t)rows * j] - A[(size_t)j + (size_t)rows * i];
      asym += d * d; nrm += A[(size_t)i + (size_t)rows * j] * A[(size_t)i + (size_t)rows * j];
    }
  const size_t nn = (size_t)rows * rows * sizeof(double);
  if (std::sqrt(asym) > 1e-12 * std::sqrt(nrm)) {      // no orthogonal eigenbasis: inverse per value of rho (refresh)
    nonsym = true;
    dirty = true;
    L.alloc(nn); Minv.alloc(nn); Mwork.alloc(nn);
    piv.alloc((size_t)rows * sizeof(int)); pval.alloc(sizeof(double));
    AO_HIP(hipMemcpyAsync(L.p, L_host, nn, hipMemcpyHostToDevice, s));
    AO_HIP(hipStreamSynchronize(s));
    return;
  }
  nonsym = false;
  const int sweeps = host_sym_eig(rows, A, wv, Uv);
  AO_REQUIRE(sweeps >= 0, "eigendecomposition of the quadratic-regularization matrix did not converge");
  std::vector<double> Utv((size_t)rows * rows);
  for (int64_t j = 0; j < rows; ++j)
    for (int64_t i = 0; i < rows; ++i) Utv[(size_t)j + (size_t)rows * i] = Uv[(size_t)i + (size_t)rows * j];
  L.alloc(nn); U.alloc(nn); Ut.alloc(nn); w.alloc((size_t)rows * sizeof(double));
  AO_HIP(hipMemcpyAsync(L.p, L_host, nn, hipMemcpyHostToDevice, s));
  AO_HIP(hipMemcpyAsync(U.p, Uv.data(), nn, hipMemcpyHostToDevice, s));
  AO_HIP(hipMemcpyAsync(Ut.p, Utv.data(), nn, hipMemcpyHostToDevice, s));
  AO_HIP(hipMemcpyAsync(w.p, wv.data(), (size_t)rows * sizeof(double), hipMemcpyHostToDevice, s));
  AO_HIP(hipStreamSynchronize(s));
}

// ---- non-symmetric L: (g*L + I)^-1 on the device, g = 2*eta/rho read from device memory (no host round trip) ----
// Gauss-Jordan inversion with row pivoting (what MATLAB's `\` pivots on), one elimination step per launch pair:
//   quad_pivot_k    (one workgroup)  p = argmax_{i >= k} |A(i,k)| (first maximum), pivot value
//   quad_gj_step_k  (whole chip)     B = step k applied to A, out of place (rows k and p change places on the way, so no
//                                    workgroup reads what another one writes); A and B swap roles every step
// and at the end the row exchanges are undone as column exchanges (quad_unswap_k, one thread per row).  2n launches
// and 2 n^2 doubles of traffic per step: n = 4096 costs ~0.25 s per refresh (once per outer iteration; the host version
// it replaces took tens of seconds and stalled the stream), n = 41 (the parity test) ~0.4 ms.  A zero pivot gives
// Inf/NaN like MATLAB's `\` on a singular matrix.
__global__ void quad_build_k(const double* __restrict__ L, double* __restrict__ A, int64_t n, double eta,
                             const double* rho, double rho_mul) {
  const double g = 2.0 * (eta / (rho[0] * rho_mul));
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n * n; e += (int64_t)gridDim.x * blockDim.x)
    A[e] = g * L[e] + ((e % n == e / n) ? 1.0 : 0.0);
}
__global__ __launch_bounds__(1024) void quad_pivot_k(const double* __restrict__ A, int64_t n, int64_t k, int* piv, double* pv) {
  __shared__ double bv[16];
  __shared__ int64_t bi[16];
  const double* col = A + n * k;
  double best = -1.0;
  int64_t at = k;
  for (int64_t i = k + threadIdx.x; i < n; i += 1024) {
    const double v = fabs(col[i]);
    if (v > best || !(v == v)) { best = v == v ? v : 1e308 * 10.0; at = i; if (!(v == v)) break; }   // NaN wins (it propagates)
  }
  // first maximum: larger value, then smaller index
  for (int o = 32; o > 0; o >>= 1) {
    const double ov = __shfl_xor(best, o);
    const int64_t oi = __shfl_xor(at, o);
    if (ov > best || (ov == best && oi < at)) { best = ov; at = oi; }
  }
  if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = at; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w)
      if (bv[w] > best || (bv[w] == best && bi[w] < at)) { best = bv[w]; at = bi[w]; }
    piv[k] = (int)at;
    pv[0] = col[at];
  }
}
static constexpr int kGjCols = 8;                     // columns per workgroup of a Gauss-Jordan step
__global__ __launch_bounds__(256) void quad_gj_step_k(const double* __restrict__ A, double* __restrict__ B, int64_t n, int64_t k,
                                                      const int* __restrict__ piv, const double* __restrict__ pv) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t j0 = (int64_t)blockIdx.y * kGjCols;
  const int64_t p = piv[k];
  const double inv = 1.0 / pv[0];
  if (i >= n) return;
  const int64_t s = (i == p) ? k : i;                  // row i after the exchange of rows k and p (i != k)
  const double f = A[s + n * k];
#pragma unroll
  for (int c = 0; c < kGjCols; ++c) {
    const int64_t j = j0 + c;
    if (j >= n) break;
    const double r = (j == k ? 1.0 : A[p + n * j]) * inv;       // new row k
    B[i + n * j] = (i == k) ? r : ((j == k ? 0.0 : A[s + n * j]) - f * r);
  }
}
__global__ void quad_unswap_k(double* A, int64_t n, const int* __restrict__ piv) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int64_t k = n - 1; k >= 0; --k) {
    const int64_t p = piv[k];
    if (p != k) { const double t = A[i + n * k]; A[i + n * k] = A[i + n * p]; A[i + n * p] = t; }
  }
}

const double* QuadPrep::refresh(double eta, const double* rho_dev, double rho_mul, hipStream_t s) {
  AO_REQUIRE(nonsym && n > 0 && Minv.p && Mwork.p, "quadratic regularization: non-symmetric matrix not prepared");
  // rho is fixed inside an ADMM loop and moves once per outer iteration: the engine marks the cache dirty at the head of
  // every outer iteration (Engine::solve), build() does for the op-level entries; a different rho pointer / multiplier /
  // eta is a different matrix as well
  if (!dirty && key_rho == rho_dev && key_mul == rho_mul && key_eta == eta) return result;
  const int64_t N = n;
  double* a = Minv.d();
  double* b = Mwork.d();
  int64_t nb = cdiv(N * N, 256);
  if (nb > 4096) nb = 4096;
  quad_build_k<<<(unsigned)nb, 256, 0, s>>>(L.d(), a, N, eta, rho_dev, rho_mul);
  AO_KERNEL_CHECK();
  const dim3 grid((unsigned)cdiv(N, 256), (unsigned)cdiv(N, kGjCols));
  for (int64_t k = 0; k < N; ++k) {
    quad_pivot_k<<<1, 1024, 0, s>>>(a, N, k, piv.as<int>(), pval.d());
    quad_gj_step_k<<<grid, 256, 0, s>>>(a, b, N, k, piv.as<int>(), pval.d());
    std::swap(a, b);
  }
  AO_KERNEL_CHECK();
  quad_unswap_k<<<(unsigned)cdiv(N, 256), 256, 0, s>>>(a, N, piv.as<int>());
  AO_KERNEL_CHECK();
  result = a;
  dirty = false; key_rho = rho_dev; key_mul = rho_mul; key_eta = eta;
  return result;
}

// W(i,:) *= 1 / (2*eta/rho*w_i + 1)
__global__ void quad_scale_k(double* W, int64_t rows, int R, const double* w, double eta, const double* rho, double rho_mul,
                             const AdmmCtl* ctl) {
  CTL_GUARD(ctl);
  const double g = 2.0 * (eta / (rho[0] * rho_mul));
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < rows * R; e += (int64_t)gridDim.x * blockDim.x)
    W[e] = W[e] / (g * w[e % rows] + 1.0);
}

static constexpr int64_t kTvLdsRows = 8192;

size_t prox_ws_bytes(int type, int64_t rows, int R) {
  switch (type) {
    case AOADMM_C_NONDECREASING:
    case AOADMM_C_NONINCREASING:
    case AOADMM_C_UNIMODAL: return iso_ws_bytes(rows, R);
    case AOADMM_C_GL_SMOOTH: return prox_gl_ws_doubles(rows, R) * sizeof(double);
    case AOADMM_C_ORTHONORMAL: return ortho_ws_doubles(rows, R) * sizeof(double);
    case AOADMM_C_QUADRATIC: return (size_t)R * rows * sizeof(double);
    case AOADMM_C_TV: return rows > kTvParMax ? (size_t)R * tv_ws_doubles(rows) * sizeof(double) : 16;
    default: return 16;
  }
}

// Pc, start and J of a column in LDS (13 bytes per row): up to 12 000 rows beside the kernel's static LDS
static constexpr int64_t kTvHybridMax = 12000;
static size_t tv_hybrid_lds(int64_t rows) { return (size_t)(rows + 1) * 8 + (size_t)(rows + 2) * 4 + (size_t)rows + 64; }
static size_t tv_fast_lds(int64_t rows) {            // Pc + val doubles, 32-bit keys, start, J
  return (size_t)(2 * rows + 1) * 8 + (size_t)rows * 4 + (size_t)(rows + 2) * 4 + (size_t)rows + 64;
}
static void tv_fast_launch(const ColArgs& a, const double* warm, int64_t ldw, const TvFused& fz, const AdmmCtl* ctl,
                           hipStream_t s) {
  if (a.rows > kTvParMax) {
    AO_REQUIRE(a.ws != nullptr && a.rows < (int64_t(1) << 24), "TV prox: %lld rows need the global workspace (below 2^24 rows)",
               (long long)a.rows);
    static const bool all_global = getenv("AOADMM_TV_ALL_GLOBAL") != nullptr;   // development switch (tools/time_tv_long*.py)
    if (a.rows <= kTvHybridMax && !all_global) {
      ensure_dynamic_lds(reinterpret_cast<const void*>(prox_tv_fast_k<1024, 2>), (int)tv_hybrid_lds(kTvHybridMax));
      prox_tv_fast_k<1024, 2><<<a.R, 1024, tv_hybrid_lds(a.rows), s>>>(a, warm, ldw, fz, ctl);
    } else {
      prox_tv_fast_k<1024, 1><<<a.R, 1024, 0, s>>>(a, warm, ldw, fz, ctl);
    }
  } else if (a.rows > 1024) {
    ensure_dynamic_lds(reinterpret_cast<const void*>(prox_tv_fast_k<1024>), (int)tv_fast_lds(kTvParMax));
    prox_tv_fast_k<1024><<<a.R, 1024, tv_fast_lds(a.rows), s>>>(a, warm, ldw, fz, ctl);
  } else {
    prox_tv_fast_k<256><<<a.R, 256, tv_fast_lds(a.rows), s>>>(a, warm, ldw, fz, ctl);
  }
  AO_KERNEL_CHECK();
}

void prox_apply(const ProxSpec& ps, const double* V, int64_t ldv, double* Zout, int64_t ldz,
                int64_t rows, int R, const double* rho_dev, double rho_mul, double* ws,
                const AdmmCtl* ctl, hipStream_t s, const double* warm, int64_t ldw) {
  ColArgs a;
  a.V = V; a.Z = Zout; a.ldv = ldv; a.ldz = ldz; a.rows = rows; a.R = R; a.type = ps.type;
  a.p0 = ps.p0; a.p1 = ps.p1; a.rho = rho_dev; a.rho_mul = rho_mul; a.ws = ws;
  if (rows <= 0 || R <= 0) return;
  switch (ps.type) {
    case AOADMM_C_NONNEG: case AOADMM_C_BOX: case AOADMM_C_L1_REG: case AOADMM_C_L0_REG:
    case AOADMM_C_RIDGE: case AOADMM_C_SIMPLEX_ROW: {
      const unsigned blocks = (unsigned)cdiv(rows, 128);
      if (R <= 8) prox_rows_k<8><<<blocks, 128, 0, s>>>(a, ctl);
      else if (R <= 16) prox_rows_k<16><<<blocks, 128, 0, s>>>(a, ctl);
      else if (R <= 32) prox_rows_k<32><<<blocks, 128, 0, s>>>(a, ctl);
      else prox_rows_k<64><<<blocks, 128, 0, s>>>(a, ctl);
      break;
    }
    case AOADMM_C_L2_BALL: case AOADMM_C_NONNEG_L2_BALL: case AOADMM_C_NONNEG_L2_SPHERE: case AOADMM_C_L2_REG:
      prox_colnorm_k<<<R, 256, 0, s>>>(a, ctl);
      break;
    case AOADMM_C_SIMPLEX_COL: case AOADMM_C_L1_BALL:
      prox_simplex_col_k<<<R, 256, 0, s>>>(a, ctl);
      break;
    case AOADMM_C_TV: {
      static const bool seq_long = getenv("AOADMM_TV_SEQ_LONG") != nullptr;     // development switch: one-thread scan beyond 4096 rows
      if (rows <= kTvParMax || (!seq_long && ws != nullptr && rows < (int64_t(1) << 24))) {   // LDS-resident, or the same in the workspace
        tv_fast_launch(a, warm, ldw, TvFused(), ctl, s);
        break;
      }
      const int use_lds = rows <= kTvLdsRows;
      const size_t sh = use_lds ? (size_t)2 * rows * sizeof(double) : 0;
      ensure_dynamic_lds(reinterpret_cast<const void*>(prox_tv_k), (int)(2 * kTvLdsRows * sizeof(double)));
      prox_tv_k<<<R, 64, sh, s>>>(a, use_lds, ctl);
      break;
    }
    case AOADMM_C_NONDECREASING: prox_iso(V, ldv, Zout, ldz, rows, R, 0, 0.0, ws, ctl, s); break;
    case AOADMM_C_NONINCREASING: prox_iso(V, ldv, Zout, ldz, rows, R, 1, 0.0, ws, ctl, s); break;
    case AOADMM_C_UNIMODAL: prox_iso(V, ldv, Zout, ldz, rows, R, 2, ps.p0, ws, ctl, s); break;
    case AOADMM_C_GL_SMOOTH:
      if (!prox_gl_pcr(V, ldv, Zout, ldz, rows, R, ps.p0, rho_dev, rho_mul, ctl, s, ws)) prox_gl_k<<<R, 64, 0, s>>>(a, ctl);
      break;
    case AOADMM_C_ORTHONORMAL: prox_ortho(V, ldv, Zout, ldz, rows, R, ws, ctl, s); break;
    case AOADMM_C_QUADRATIC: {                       // (2*eta/rho*L + I) \ x = U diag(1/(2 eta/rho w + 1)) U' x   (:66)
      if (ps.quad) {                                 // non-symmetric L: Z = (2*eta/rho*L + I)^-1 * V, inverse cached per rho
        const double* Mi = ps.quad->refresh(ps.p0, rho_dev, rho_mul, s);
        gemm_small(Zout, ldz, Mi, rows, V, ldv, rows, (int)rows, R, 0, coef(1.0), 0.0, ctl, s);
        break;
      }
      AO_REQUIRE(ps.LU && ps.LUt && ps.Lw, "quadratic regularization: matrix not prepared");
      gemm_small(ws, rows, ps.LUt, rows, V, ldv, rows, (int)rows, R, 0, coef(1.0), 0.0, ctl, s);
      int64_t nb = cdiv(rows * R, 256);
      if (nb > 1024) nb = 1024;
      quad_scale_k<<<(unsigned)nb, 256, 0, s>>>(ws, rows, R, ps.Lw, ps.p0, rho_dev, rho_mul, ctl);
      AO_KERNEL_CHECK();
      gemm_small(Zout, ldz, ps.LU, rows, ws, rows, rows, (int)rows, R, 0, coef(1.0), 0.0, ctl, s);
      break;
    }
    default:
      throw Error(AOADMM_ERR_UNSUPPORTED, fmt("constraint id %d has no device prox (route to the MATLAB path)", ps.type));
  }
  AO_KERNEL_CHECK();
}

// ===========================================================================
// host-side composition
// ===========================================================================
template <int CPW>
static void launch_rows(const FusedArgs& a, unsigned blocks, hipStream_t s) {
  const int RP = a.R | 1;
  const size_t sh = ((size_t)a.R * a.R + (size_t)64 * RP + 16) * sizeof(double);
  if (sh > 65536) {                                   // R = 61..64: just above the default dynamic-LDS limit
    ensure_dynamic_lds(reinterpret_cast<const void*>(admm_rows_k<CPW, true>), (int)sh);
    ensure_dynamic_lds(reinterpret_cast<const void*>(admm_rows_k<CPW, false>), (int)sh);
  }
  if (a.R == 4 * CPW) admm_rows_k<CPW, true><<<blocks, kRowBlock, sh, s>>>(a);
  else admm_rows_k<CPW, false><<<blocks, kRowBlock, sh, s>>>(a);
}
template <int KS>
static void launch_rows_mfma(const FusedArgs& a, const SpecExtra& ex, bool fin, unsigned blocks, hipStream_t s) {
  const size_t sh = fin ? (size_t)16 * (a.R | 1) * sizeof(double) : 0;
  if (fin) admm_rows_mfma_k<KS, true><<<blocks, kSpecThreads, sh, s>>>(a, ex);
  else admm_rows_mfma_k<KS, false><<<blocks, kSpecThreads, sh, s>>>(a, ex);
}
static void launch_rows_mfma_any(const FusedArgs& a, const SpecExtra& ex, bool fin, unsigned blocks, hipStream_t s) {
  const int need = (a.R + 3) / 4;
  if (need <= 1) launch_rows_mfma<1>(a, ex, fin, blocks, s);
  else if (need <= 2) launch_rows_mfma<2>(a, ex, fin, blocks, s);
  else if (need <= 3) launch_rows_mfma<3>(a, ex, fin, blocks, s);
  else if (need <= 4) launch_rows_mfma<4>(a, ex, fin, blocks, s);
  else if (need <= 5) launch_rows_mfma<5>(a, ex, fin, blocks, s);
  else if (need <= 6) launch_rows_mfma<6>(a, ex, fin, blocks, s);
  else launch_rows_mfma<8>(a, ex, fin, blocks, s);
}
static void launch_row_iteration(const FusedArgs& a, unsigned blocks, hipStream_t s) {
  if (a.Binv) {
    const int need = (a.R + 3) / 4;
    if (need <= 1) launch_rows<1>(a, blocks, s);
    else if (need <= 2) launch_rows<2>(a, blocks, s);
    else if (need <= 3) launch_rows<3>(a, blocks, s);
    else if (need <= 4) launch_rows<4>(a, blocks, s);
    else if (need <= 5) launch_rows<5>(a, blocks, s);
    else if (need <= 6) launch_rows<6>(a, blocks, s);
    else if (need <= 8) launch_rows<8>(a, blocks, s);
    else if (need <= 12) launch_rows<12>(a, blocks, s);
    else launch_rows<16>(a, blocks, s);
  } else {
    const size_t sh = (size_t)a.R * a.R * sizeof(double);
    if (a.R <= 8) admm_rowL_k<8><<<blocks, kRowThreads, sh, s>>>(a);
    else if (a.R <= 16) admm_rowL_k<16><<<blocks, kRowThreads, sh, s>>>(a);
    else if (a.R <= 32) admm_rowL_k<32><<<blocks, kRowThreads, sh, s>>>(a);
    else admm_rowL_k<64><<<blocks, kRowThreads, sh, s>>>(a);
  }
}

// ---------------------------------------------------------------------------
// Short modes: the whole ADMM_constrained_only loop (:596-622) in ONE launch of one workgroup, thread = row, the row's
// operands (A, Z, mu, fac) in registers for the length of the loop.  Per inner iteration: A_inner (:608), the solve
// (:609; inv(L*L') or L from LDS, or the row's own factor for the per-row systems of a PARAFAC2 C mode, :602-606),
// update_constraint (:1420-1429) for the element-/row-wise catalogue and the column-norm family (column sums through
// one workgroup reduction), the four sums of eval_res_ADMM_constr (:1079-1096) through a second one, and the while
// test (:600) -- every thread holds the same totals, so the decision needs no further exchange.  At the end the
// kernel also leaves the Gram matrix of the new factor and its row-major copy (what atb_small would compute, :148).
// Replaces 2 launches per loop (element-wise prox), 3 per inner iteration (column-norm prox, PARAFAC2 C mode) plus
// the two Gram launches.  Sums run over waves in a fixed order: results do not depend on timing.
// ---------------------------------------------------------------------------
static constexpr int kWgLoopRows = 256;             // one row per thread: longer modes keep the multi-workgroup loops
static bool prox_is_colnorm(int t) {
  return t == AOADMM_C_L2_BALL || t == AOADMM_C_NONNEG_L2_BALL || t == AOADMM_C_NONNEG_L2_SPHERE || t == AOADMM_C_L2_REG;
}

// totals of NV per-thread values over the workgroup, the same in every thread (butterfly inside the wave, waves in
// index order through LDS)
template <int NV, int NW>
__device__ __forceinline__ void wg_sum(double (&v)[NV], double* red /* [NW][NV] */) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = wave_sum(v[k]);
  if (NW == 1) return;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) red[w * NV + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    double x = 0.0;
#pragma unroll
    for (int q = 0; q < NW; ++q) x += red[q * NV + k];
    v[k] = x;
  }
  __syncthreads();
}

// SOLVE: 0 = inv(L*L') shared by all rows, 2 = one Cholesky factor per row.  For 0 the matrix sits in LDS as an
// RMAX x RMAX block padded with zeros and the row operands are padded with zeros, so the solve is straight-line code
// without rank tests; a padded column stays exactly zero.
template <int RMAX, int NT, int SOLVE>
__global__ __launch_bounds__(NT) void admm_loop_wg_k(WgLoopU a) {
  constexpr int NW = NT / 64;
  extern __shared__ double lds[];                    // M[RMAX*RMAX] | red[NW][max(RMAX,4)] | fsh[rows][RP] (Gram)
  constexpr int NRED = RMAX > 4 ? RMAX : 4;
  double* Msh = lds;
  double* red = lds + RMAX * RMAX;
  double* fsh = red + NW * NRED;
  AdmmCtl* ctl = a.ctl;
  if (!a.reset && ctl->active == 0) return;          // a failed factorisation (sys_build) leaves the state untouched
  const int t = threadIdx.x;
  const int R = a.R;
  const int64_t rows = a.rows;
  const bool have = t < rows;
  const int64_t i = have ? t : rows - 1;             // clamped: padding threads compute on the last row, store nothing
  double av[RMAX], z[RMAX], mu[RMAX], x[RMAX];
#pragma unroll
  for (int c = 0; c < RMAX; ++c) {
    const bool ok = c < R;
    const int64_t o = i + rows * (ok ? c : 0);
    av[c] = ok ? a.A[o] : 0.0; z[c] = ok ? a.Z[o] : 0.0; mu[c] = ok ? a.mu[o] : 0.0;
    x[c] = 0.0;
  }
  if (SOLVE != 2) {
    for (int e = t; e < RMAX * RMAX; e += NT) {
      const int r = e % RMAX, c = e / RMAX;
      Msh[e] = (r < R && c < R) ? a.Binv[r + R * c] : 0.0;
    }
  }
  const double rho = SOLVE == 2 ? a.rho[i] : a.rho[0];
  double rho_prox;                                   // max(rho) of a PARAFAC2 C mode (:1423-1424); rho otherwise
  if (SOLVE == 2 && a.rho_prox == nullptr) {         // the rows' own rho are this block's registers: max without a launch
    double mx = rho;                                 // padding threads hold the last row's value
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
    if ((t & 63) == 0) red[t >> 6] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int q = 1; q < NW; ++q) mx = fmax(mx, red[q]);
    rho_prox = mx;
    __syncthreads();                                 // red is reused by the loop
  } else {
    rho_prox = a.rho_prox[0];
  }
  const double rh = rho / 2;
  const ElemProx ep = elem_prox_of(a.ptype, a.p0, a.p1, rho_prox);
  const bool colnorm = a.ptype == AOADMM_C_L2_BALL || a.ptype == AOADMM_C_NONNEG_L2_BALL ||
                       a.ptype == AOADMM_C_NONNEG_L2_SPHERE || a.ptype == AOADMM_C_L2_REG;
  const bool clampv = a.ptype == AOADMM_C_NONNEG_L2_BALL || a.ptype == AOADMM_C_NONNEG_L2_SPHERE;
  __syncthreads();
  double pr = 0.0, du = 0.0;
  int it = 0;
  for (;;) {
    // the system matrix is re-read (LDS / L1) every iteration: hoisted out of the loop its entries alone would take
    // more registers than the row operands (compiler barrier)
    asm volatile("" ::: "memory");
    // A_inner = A + rho/2*(Z - mu) (:608) and the solve (:609)
#pragma unroll
    for (int c = 0; c < RMAX; ++c) x[c] = av[c] + rh * (z[c] - mu[c]);
    if (SOLVE == 2) {
      const double* Lk = a.L + i * (int64_t)R * R;
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
    } else {
      double y[RMAX];
#pragma unroll
      for (int c = 0; c < RMAX; ++c) {
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < RMAX; ++q) acc += x[q] * Msh[q + RMAX * c];
        y[c] = acc;
      }
#pragma unroll
      for (int c = 0; c < RMAX; ++c) x[c] = y[c];
    }
    // update_constraint (:1420-1429): Z = prox(fac + mu), mu += fac - Z
    double zn[RMAX];
#pragma unroll
    for (int c = 0; c < RMAX; ++c) zn[c] = x[c] + mu[c];
    if (colnorm) {
      double cs[RMAX];
#pragma unroll
      for (int c = 0; c < RMAX; ++c) {
        const double y = clampv ? fmax(zn[c], 0.0) : zn[c];
        cs[c] = have ? y * y : 0.0;
      }
      wg_sum<RMAX, NW>(cs, red);
#pragma unroll
      for (int c = 0; c < RMAX; ++c) {
        if (c >= R) { zn[c] = 0.0; continue; }       // uniform
        const double nrm = sqrt(cs[c]);
        const double y = clampv ? fmax(zn[c], 0.0) : zn[c];
        if (a.ptype == AOADMM_C_NONNEG_L2_SPHERE) {
          if (nrm == 0.0) {
            // all-zero column: unit vector at the first maximum of the input (prox_normalized_nonneg.m:5-7)
            double bv = have ? zn[c] : -INFINITY;
            int bi = have ? t : 0x7fffffff;
            for (int off = 32; off > 0; off >>= 1) {
              const double ov = __shfl_xor(bv, off);
              const int oi = __shfl_xor(bi, off);
              if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            if (NW > 1) {
              int* ired = reinterpret_cast<int*>(red + NW);
              if ((t & 63) == 0) { red[t >> 6] = bv; ired[t >> 6] = bi; }
              __syncthreads();
              bv = red[0]; bi = ired[0];
              for (int q = 1; q < NW; ++q)
                if (red[q] > bv || (red[q] == bv && ired[q] < bi)) { bv = red[q]; bi = ired[q]; }
              __syncthreads();
            }
            zn[c] = t == bi ? 1.0 : 0.0;
          } else {
            zn[c] = y / nrm;
          }
        } else {
          double scale;
          if (a.ptype == AOADMM_C_L2_REG) {
            const double g = a.p0 / rho_prox;
            scale = nrm > g ? 1.0 - g / nrm : 0.0;
          } else {
            scale = nrm > a.p0 ? a.p0 / nrm : 1.0;
          }
          zn[c] = y * scale;
        }
      }
    } else if (a.ptype == AOADMM_C_SIMPLEX_ROW) {
      simplex_regs<RMAX>(zn, R, a.p0);
#pragma unroll
      for (int c = 0; c < RMAX; ++c) zn[c] = c < R ? zn[c] : 0.0;
    } else {
#pragma unroll
      for (int c = 0; c < RMAX; ++c) zn[c] = c < R ? elem_prox(ep, zn[c]) : 0.0;
    }
    double sm[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int c = 0; c < RMAX; ++c) {
      const double mn = mu[c] + x[c] - zn[c];
      const double d = x[c] - zn[c], e = zn[c] - z[c];
      if (have) { sm[0] += d * d; sm[1] += x[c] * x[c]; sm[2] += mn * mn; sm[3] += e * e; }
      mu[c] = mn; z[c] = zn[c];
    }
    wg_sum<4, NW>(sm, red);
    pr = sqrt(sm[0]) / sqrt(sm[1]);                                          // :1085
    const double sc = sqrt(sm[2]);
    du = sc > 0 ? sqrt(sm[3]) / sc : sqrt(sm[3]);                            // :1087-1092
    ++it;
    if (!(it < a.max_inner && (pr > a.tol_pr || du > a.tol_du))) break;      // :600
  }
  if (have) {
#pragma unroll
    for (int c = 0; c < RMAX; ++c) {
      if (c < R) {
        const int64_t o = i + rows * c;
        a.fac[o] = x[c]; a.Z[o] = z[c]; a.mu[o] = mu[c];
      }
    }
  }
  if (t == 0) {
    ctl->res[0] = 0.0; ctl->res[1] = pr; ctl->res[2] = 0.0; ctl->res[3] = du;
    ctl->iters = it;
    ctl->active = 0;
  }
  if (a.gram == nullptr) return;
  // Gram matrix fac'*fac (:148) and the row-major copy of fac: rows through LDS, one (p, q) pair per wave at a time
  constexpr int RP = RMAX | 1;
  if (have) {
#pragma unroll
    for (int c = 0; c < RMAX; ++c) {
      fsh[t * RP + c] = x[c];
      if (c < R && a.facT) a.facT[(int64_t)t * R + c] = x[c];
    }
  }
  __syncthreads();
  const int lane = t & 63, w = t >> 6;
  const int npairs = R * (R + 1) / 2;
  for (int e = w; e < npairs; e += NW) {
    int p = 0, rem = e;                                // e -> (p, q), p <= q, row-wise over the upper triangle
    while (rem >= R - p) { rem -= R - p; ++p; }
    const int q = p + rem;
    double acc = 0.0;
    for (int64_t r0 = lane; r0 < rows; r0 += 64) acc += fsh[r0 * RP + p] * fsh[r0 * RP + q];
    acc = wave_sum(acc);
    if (lane == 0) { a.gram[p + R * q] = acc; a.gram[q + R * p] = acc; }
  }
}

size_t admm_loop_wg_lds(int rmax, int nt, int64_t rows, bool gram) {
  const int nred = rmax > 4 ? rmax : 4;
  return ((size_t)rmax * rmax + (size_t)(nt / 64) * nred + (gram ? (size_t)rows * (rmax | 1) : 0)) * sizeof(double);
}

bool admm_loop_wg_ok(int64_t rows, int R, int ptype, int max_inner) {
  static const bool off = getenv("AOADMM_NO_WG_LOOP") != nullptr;          // development switch
  if (off || max_inner < 1) return false;
  if (!(prox_is_fusable(ptype) || prox_is_colnorm(ptype))) return false;
  return rows <= kWgLoopRows && R <= 16;
}

void admm_loop_wg(const WgLoopU& a, hipStream_t s) {
  AO_REQUIRE(admm_loop_wg_ok(a.rows, a.R, a.ptype, a.max_inner), "admm_loop_wg: mode too large for the one-workgroup loop");
  const bool gram = a.gram != nullptr;
  AO_REQUIRE(a.per_row ? a.L != nullptr : a.Binv != nullptr, "admm_loop_wg: the shared system comes as inv(L*L'), per-row systems as factors");
  auto go = [&](auto rm) {
    constexpr int RM = decltype(rm)::value;
    const size_t lds = admm_loop_wg_lds(RM, kWgLoopRows, a.rows, gram);
    if (!a.per_row) admm_loop_wg_k<RM, kWgLoopRows, 0><<<1, kWgLoopRows, lds, s>>>(a);
    else admm_loop_wg_k<RM, kWgLoopRows, 2><<<1, kWgLoopRows, lds, s>>>(a);
  };
  if (a.R <= 4) go(std::integral_constant<int, 4>());
  else if (a.R <= 8) go(std::integral_constant<int, 8>());
  else go(std::integral_constant<int, 16>());
  AO_KERNEL_CHECK();
}

void admm_constrained_loop(const AdmmMode& m, double* part, double* V, double* Znew, double* prox_ws,
                           AdmmCtl* ctl, int max_inner, double tol_pr, double tol_du, hipStream_t s,
                           LoopEnd* deferred_end, GramFold* gf) {
  FusedArgs a;
  a.A = m.A; a.L = m.L; a.Binv = m.Binv; a.rho = m.rho; a.fac = m.fac; a.Z = m.Z; a.mu = m.mu; a.V = V; a.part = part;
  a.ctl = ctl;
  a.rows = m.rows; a.R = m.R; a.ptype = m.prox.type; a.p0 = m.prox.p0; a.p1 = m.prox.p1;
  a.fused = prox_is_fusable(m.prox.type) ? 1 : 0;
  a.max_inner = max_inner; a.tol_pr = tol_pr; a.tol_du = tol_du;
  int64_t nblk = cdiv(m.rows, kRowThreads);
  if (nblk > kMaxParts) nblk = kMaxParts;
  const unsigned blocks = (unsigned)nblk;
  const int64_t n = m.rows * m.R;
  int64_t nbd = cdiv(n, 1024);
  if (nbd > 64) nbd = 64;
  // element-wise prox: the whole loop in two launches (admm_rows_mfma_k)
  const int64_t tiles = cdiv(m.rows, 16);            // one wave per 16 rows
  if (a.fused && a.ptype != AOADMM_C_SIMPLEX_ROW && a.Binv && a.R <= 32 && tiles <= 64 * kSpecPartJ &&
      max_inner <= kSpecMaxInner && (int64_t)max_inner * tiles <= 2 * kMaxParts) {
    SpecExtra ex{nullptr, nullptr};
    launch_rows_mfma_any(a, ex, false, (unsigned)tiles, s);
    AO_KERNEL_CHECK();
    if (gf && gf->ws) { ex.gram_ws = gf->ws; ex.At = gf->At; gf->nb = (int)tiles; }
    launch_rows_mfma_any(a, ex, true, (unsigned)tiles, s);
    AO_KERNEL_CHECK();
    if (deferred_end) *deferred_end = LoopEnd();    // the loop is closed: pass 2 recorded iters / residuals
    return;
  }
  const bool tv_fused = !a.fused && m.prox.type == AOADMM_C_TV &&                        // prox + dual in one kernel
                        (m.rows <= kTvParMax || (prox_ws != nullptr && m.rows < (int64_t(1) << 24)));
  const int nparts = a.fused ? (int)blocks : (tv_fused ? m.R : (int)nbd);
  for (int it = 0; it < max_inner; ++it) {
    a.it = it;
    a.nparts_prev = nparts;
    launch_row_iteration(a, blocks, s);
    AO_KERNEL_CHECK();
    if (tv_fused) {
      ColArgs ca;
      ca.V = V; ca.Z = Znew; ca.ldv = m.rows; ca.ldz = m.rows; ca.rows = m.rows; ca.R = m.R; ca.type = m.prox.type;
      ca.p0 = m.prox.p0; ca.p1 = m.prox.p1; ca.rho = m.rho; ca.rho_mul = 1.0; ca.ws = prox_ws;
      TvFused fz;
      fz.Z = m.Z; fz.mu = m.mu; fz.part = part + (int64_t)(it & 1) * kMaxParts * 4; fz.ld = m.rows;
      tv_fast_launch(ca, m.Z, m.rows, fz, ctl, s);
    } else if (!a.fused) {
      prox_apply(m.prox, V, m.rows, Znew, m.rows, m.rows, m.R, m.rho, 1.0, prox_ws, ctl, s, m.Z, m.rows);
      dual_update_k<<<(unsigned)nbd, 256, 0, s>>>(m.fac, m.Z, m.mu, Znew, n, part + (int64_t)(it & 1) * kMaxParts * 4, ctl);
      AO_KERNEL_CHECK();
    }
  }
  LoopEnd le;
  le.part = part; le.nparts = nparts; le.max_inner = max_inner; le.tol_pr = tol_pr; le.tol_du = tol_du; le.ctl = ctl;
  if (deferred_end) {
    *deferred_end = le;
  } else {
    admm_loop_end_k<<<1, 64, 0, s>>>(le);
    AO_KERNEL_CHECK();
  }
}

__global__ void copy_add_k(double* V, double* Zold, const double* fac, const double* Z, const double* mu, int64_t n,
                           const AdmmCtl* ctl) {
  CTL_GUARD(ctl);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    V[i] = fac[i] + mu[i];
    Zold[i] = Z[i];
  }
}
// mu += fac - Z (:1063) and the four sums of eval_res_ADMM_constr in the same pass:
// ||fac-Z||^2, ||fac||^2, ||mu||^2, ||Z-Zold||^2.  One block writes the slots itself; several blocks leave
// per-block partial sums that dual_sums_fin_k adds in block order.
__global__ __launch_bounds__(256) void dual_sums_k(const double* fac, const double* Z, double* mu, const double* Zold,
                                                   int64_t n, double* slots, double* ws, const AdmmCtl* ctl) {
  CTL_GUARD(ctl);
  __shared__ double sh4[4];
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double f = fac[i], z = Z[i];
    const double m = mu[i] + f - z;
    mu[i] = m;
    const double dz = z - Zold[i];
    s0 += (f - z) * (f - z); s1 += f * f; s2 += m * m; s3 += dz * dz;
  }
  s0 = block256_sum(s0, sh4); s1 = block256_sum(s1, sh4); s2 = block256_sum(s2, sh4); s3 = block256_sum(s3, sh4);
  if (threadIdx.x == 0) {
    double* o = gridDim.x == 1 ? slots : ws + 4 * (int64_t)blockIdx.x;
    o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3;
  }
}
__global__ void dual_sums_fin_k(double* slots, const double* ws, int nb, const AdmmCtl* ctl) {
  CTL_GUARD(ctl);
  if (threadIdx.x >= 4) return;
  double t = 0.0;
  for (int b = 0; b < nb; ++b) t += ws[4 * b + threadIdx.x];
  slots[threadIdx.x] = t;
}

// update_constraint (:1420-1429) with an element-/row-wise prox in ONE launch of one workgroup (thread = row):
// Zold = Z, Z = prox(fac + mu, rho), mu += fac - Z and the four sums of eval_res_ADMM_constr.  The three-launch form
// below (fac + mu, prox, dual + sums) is ~12 us of launches per call; the coupled and PARAFAC2 loops of the example
// scripts call it 4-15 times per outer iteration.
template <int RMAX>
__global__ __launch_bounds__(256) void constraint_update_rows_k(ColArgs a, const double* fac, double* mu, double* Zold,
                                                               double* slots, const AdmmCtl* ctl) {
  CTL_GUARD(ctl);
  __shared__ double sh4[4];
  const double rho = a.rho[0] * a.rho_mul;
  const int R = a.R;
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  for (int64_t i = threadIdx.x; i < a.rows; i += 256) {
    double f[RMAX], m[RMAX], zo[RMAX], z[RMAX];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
      const bool ok = r < R;
      f[r] = ok ? fac[i + a.rows * r] : 0.0;
      m[r] = ok ? mu[i + a.rows * r] : 0.0;
      zo[r] = ok ? a.Z[i + a.ldz * r] : 0.0;
      z[r] = f[r] + m[r];
    }
    if (a.type == AOADMM_C_SIMPLEX_ROW) {
      simplex_regs<RMAX>(z, R, a.p0);
    } else {
#pragma unroll
      for (int r = 0; r < RMAX; ++r) z[r] = prox_elem(a.type, z[r], a.p0, a.p1, rho);
    }
#pragma unroll
    for (int r = 0; r < RMAX; ++r)
      if (r < R) {
        const double mn = m[r] + f[r] - z[r];
        Zold[i + a.rows * r] = zo[r];
        a.Z[i + a.ldz * r] = z[r];
        mu[i + a.rows * r] = mn;
        const double dz = z[r] - zo[r];
        s0 += (f[r] - z[r]) * (f[r] - z[r]); s1 += f[r] * f[r]; s2 += mn * mn; s3 += dz * dz;
      }
  }
  s0 = block256_sum(s0, sh4); s1 = block256_sum(s1, sh4); s2 = block256_sum(s2, sh4); s3 = block256_sum(s3, sh4);
  if (threadIdx.x == 0) { slots[0] = s0; slots[1] = s1; slots[2] = s2; slots[3] = s3; }
}

void constraint_update(const ProxSpec& ps, const double* fac, double* Z, double* mu, double* Zold,
                       double* V, int64_t rows, int R, const double* rho_dev, double rho_mul,
                       double* prox_ws, double* slots, double* red_ws, const AdmmCtl* ctl,
                       hipStream_t s) {
  if (prox_is_fusable(ps.type) && rows <= 8192 && R <= 32) {
    ColArgs a;
    a.V = nullptr; a.Z = Z; a.ldv = rows; a.ldz = rows; a.rows = rows; a.R = R; a.type = ps.type;
    a.p0 = ps.p0; a.p1 = ps.p1; a.rho = rho_dev; a.rho_mul = rho_mul; a.ws = nullptr;
    if (R <= 4) constraint_update_rows_k<4><<<1, 256, 0, s>>>(a, fac, mu, Zold, slots, ctl);
    else if (R <= 8) constraint_update_rows_k<8><<<1, 256, 0, s>>>(a, fac, mu, Zold, slots, ctl);
    else if (R <= 16) constraint_update_rows_k<16><<<1, 256, 0, s>>>(a, fac, mu, Zold, slots, ctl);
    else constraint_update_rows_k<32><<<1, 256, 0, s>>>(a, fac, mu, Zold, slots, ctl);
    AO_KERNEL_CHECK();
    return;
  }
  const int64_t n = rows * R;
  int64_t nb = cdiv(n, 256);
  if (nb > 1024) nb = 1024;
  copy_add_k<<<(unsigned)nb, 256, 0, s>>>(V, Zold, fac, Z, mu, n, ctl);
  AO_KERNEL_CHECK();
  prox_apply(ps, V, rows, Z, rows, rows, R, rho_dev, rho_mul, prox_ws, ctl, s, Zold, rows);   // Z = prox(fac+mu, rho)
  int64_t nr = cdiv(n, 2048);
  if (nr > 64) nr = 64;
  dual_sums_k<<<(unsigned)nr, 256, 0, s>>>(fac, Z, mu, Zold, n, slots, red_ws, ctl);
  AO_KERNEL_CHECK();
  if (nr > 1) {
    dual_sums_fin_k<<<1, 64, 0, s>>>(slots, red_ws, (int)nr, ctl);
    AO_KERNEL_CHECK();
  }
}

__global__ void admm_finalize_generic_k(FinalizeArgs fa, AdmmCtl* ctl) {
  if (ctl->active == 0) return;
  if (threadIdx.x != 0) return;
  double prc = 0, duc = 0, prz = 0, duz = 0;
  int nc = 0, nz = 0;
  for (int m = 0; m < fa.nmodes; ++m) {
    const double* s = fa.slots[m];
    if (fa.coupled[m]) {                        // eval_res_ADMM_coupl_case* (:1099-1210)
      prc += sqrt(s[4]) / sqrt(s[7]);             // [7] = ||C||^2, or ||H*C||^2 / ||C*H||^2 for types 1, 2
      const double sc = sqrt(s[5]);
      duc += sc > 0 ? sqrt(s[6]) / sc : sqrt(s[6]);
      ++nc;
    }
    if (fa.constrained[m]) {                    // eval_res_ADMM_constr (:1079-1096)
      prz += sqrt(s[0]) / sqrt(s[1]);
      const double sc = sqrt(s[2]);
      duz += sc > 0 ? sqrt(s[3]) / sc : sqrt(s[3]);
      ++nz;
    }
  }
  if (nc) { prc /= nc; duc /= nc; }
  if (nz) { prz /= nz; duz /= nz; }             // else 0 (:690-691)
  ctl->res[0] = prc; ctl->res[1] = prz; ctl->res[2] = duc; ctl->res[3] = duz;
  const int it = ctl->iters + 1;
  ctl->iters = it;
  ctl->active = (it < fa.max_inner && (prc > fa.tol_pr_coupl || prz > fa.tol_pr_constr ||
                                       duc > fa.tol_du_coupl || duz > fa.tol_du_constr)) ? 1 : 0;
}
void admm_finalize_generic(const FinalizeArgs& fa, AdmmCtl* ctl, hipStream_t s) {
  admm_finalize_generic_k<<<1, 64, 0, s>>>(fa, ctl);
  AO_KERNEL_CHECK();
}

}  // namespace aoadmm
