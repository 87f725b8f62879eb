// Workgroup-parallel shape-constraint projections: isotonic (non-decreasing / non-increasing), unimodal
// (functions/project_unimodal.m -> project_unimodal_vector.m; constraints_to_prox.m:25-31) and the graph-Laplacian
// smoothness prox on a path (constraints_to_prox.m:68-77).  One workgroup per column.
//
// Isotonic regression through the cumulative-sum diagram.  With P_k = sum_{t<k} s_t the isotonic (non-decreasing)
// least-squares fit of a prefix s_0..s_{i-1} is the left derivative of the greatest convex minorant of the points
// (k, P_k), k = 0..i.  The vertex of that minorant before i is
//        prev(i) = argmax_{a < i} (P_i - P_a) / (i - a)              (smallest a among equal slopes)
// -- every other point lies on or above the edge a -> i, i.e. sees i under a smaller slope -- and the slope itself is
// the level of the last block.  Hence, for EVERY prefix at once and with no sequential pool merging:
//   * prev(i) for all i is n independent maximisations over the shared array P (broadcast reads from LDS);
//   * the prefix fit errors of Stout's unimodal regression obey err(i) = sse(prev(i), i) + err(prev(i)), a sum along
//     the prev-tree, evaluated for all i by pointer doubling (log2 n rounds);
//   * the blocks of one particular prefix (the whole column for the monotone constraints, the best split for the
//     unimodal one) are the tree path from its node to the root; the path is marked by the same doubling and every
//     entry takes the level of the first marked node to its right (a suffix scan).
// Work is O(n^2) slope comparisons per direction -- 2 M for a 2000-row column -- spread over up to 64 workgroups per
// column (iso_prev_k); everything else is O(n log n) in one workgroup per column (prox_iso_k).  The reference (and
// this library's first version) walks each column sequentially.
// Quirks of project_unimodal_vector.m that change numbers are kept: blocks merge on equal levels (:63), the split
// criterion pairs the left prefix 1..i with the right prefix of length n-i+1 for i >= 2 and uses the right fit alone
// for i = 1 (:22-26), the first minimum wins (ties up to the rounding of the error sums: see prox_iso_k), the right part is rebuilt with length n-i (:16), and with
// non-negativity a prefix whose last level is negative gets the error sum_{t < i-1} s_t^2 (:70, one term short).
#include "admm.h"
#include "device_utils.h"

namespace aoadmm {

#define CTL_GUARD(ctl) \
  if ((ctl) != nullptr && (ctl)->active == 0) return;

// bytes of scratch per column: 6 double arrays, 4 int arrays, 3 byte arrays of n + 1 entries
__host__ __device__ inline size_t iso_col_bytes(int64_t n) {
  const size_t n1 = (size_t)n + 1;
  return (n1 * (6 * sizeof(double) + 4 * sizeof(int) + 3) + 63) / 64 * 64;
}

struct IsoCol {
  const double* V;
  double* Z;
  int64_t ldv, ldz, rows;
  int R;
  double nonneg;       // unimodal: 1 = with projection onto the non-negative orthant
  double* ws;          // global scratch when the column does not fit LDS
};

// exclusive prefix sum of one double per thread over the workgroup, fixed order; `total` = sum over all threads
template <int NTH>
__device__ __forceinline__ double wg_exscan(double v, double& total, double* sc) {
  constexpr int NW = NTH / 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double inc = v;
  for (int off = 1; off < 64; off <<= 1) { const double u = __shfl_up(inc, off); if (lane >= off) inc += u; }
  if (lane == 63) sc[w] = inc;
  __syncthreads();
  double base = 0.0, all = 0.0;
#pragma unroll
  for (int k = 0; k < NW; ++k) { if (k < w) base += sc[k]; all += sc[k]; }
  total = all;
  __syncthreads();
  return base + inc - v;
}

struct IsoBuf {          // per direction: survives until the column is written
  double* P;             // n + 1 prefix sums
  int* PV;               // n + 1: prev(i)
  unsigned char* FL;     // n + 1: node thresholded to zero (non-negativity)
  double c;              // offset the sums were taken about (0: plain sums)
};

// Step 1 for one direction: the sequence s_t = sign * src[flip ? n-1-t : t] (parked in `park`), its prefix sums P and
// the prefix sums of squares Q.
template <int NTH>
__device__ void iso_prefix(const double* __restrict__ src, int n, bool flip, double sign, IsoBuf& b, double* Q, double* park,
                           double* sc) {
  const int t = threadIdx.x;
  const int chunk = (n + NTH - 1) / NTH;
  const int c0 = min(n, t * chunk), c1 = min(n, c0 + chunk);
  double loc = 0.0, loc2 = 0.0;
  for (int i = c0; i < c1; ++i) {
    const double s = sign * src[flip ? n - 1 - i : i];
    park[i] = s;
    loc += s;
    loc2 += s * s;
  }
  // Plain (uncentred) sums, like the reference's sumwy / sumwy2 (:45-46, :83-85): on data with exact ties -- integers,
  // quantised measurements -- the sums, the slope comparisons of iso_prev and the block levels are then exact, so equal
  // levels merge exactly as the reference merges them (:63).  Sums about the mean were more accurate for columns with a
  // large offset but turned every exact tie into a coin flip.
  double tot, tot2;
  const double c = 0.0;
  double run = wg_exscan<NTH>(loc, tot, sc);
  double run2 = wg_exscan<NTH>(loc2, tot2, sc);
  for (int i = c0; i < c1; ++i) {
    const double s = park[i];
    b.P[i] = run; run += s - c;
    Q[i] = run2; run2 += s * s;
  }
  if (c1 == n && c0 < n) { b.P[n] = run; Q[n] = run2; }
  b.c = c;
  __syncthreads();
}

// Step 2: prev(i), the thresholding flag and (with_err) the block cost for the nodes of share `part` of `nparts`.
// Thread k pairs the nodes k+1 and n-k, so every thread compares n+1 candidates, and all lanes of a wave read the same
// P[a] in the same iteration (an LDS broadcast).  Node 0 is the caller's.
template <int NTH>
__device__ void iso_prev(int n, bool nonneg, bool with_err, const IsoBuf& b, const double* Q, int part, int nparts,
                         int* PV, unsigned char* FL, double* E) {
  const double* P = b.P;
  const double c = b.c;
  for (int k = part * NTH + (int)threadIdx.x; 2 * k < n; k += nparts * NTH) {
    const int i1 = k + 1, i2 = n - k;
    const double p1 = P[i1], p2 = P[i2];
    double nb1 = p1 - P[0], nb2 = p2 - P[0];
    double db1 = (double)i1, db2 = (double)i2;
    int a1 = 0, a2 = 0;
    for (int a = 1; a < i2; ++a) {
      const double pa = P[a];
      const double n2 = p2 - pa, d2 = (double)(i2 - a);
      if (n2 * db2 > nb2 * d2) { nb2 = n2; db2 = d2; a2 = a; }      // strictly steeper: the smallest a wins ties (:63 merges them)
      if (a < i1) {
        const double n1 = p1 - pa, d1 = (double)(i1 - a);
        if (n1 * db1 > nb1 * d1) { nb1 = n1; db1 = d1; a1 = a; }
      }
    }
    auto finish = [&](int i, int a, double num, double den) {
      const double level = c + num / den;
      const bool fl = nonneg && level < 0.0;
      PV[i] = a;
      FL[i] = fl ? 1 : 0;
      if (with_err) {
        const double S = num + c * den;                                // sum of the block
        const double sse = (Q[i] - Q[a]) - S * S / den;                // :67
        E[i] = fl ? Q[i - 1] : sse;                                    // :70 (one term short, as in the reference) / :72
      }
    };
    finish(i1, a1, nb1, db1);
    if (i2 != i1) finish(i2, a2, nb2, db2);
  }
}

// Long columns: the n^2/2 comparisons of step 2 are spread over `nparts` workgroups per column and direction
// (grid: columns x directions x shares); results go to global scratch for prox_iso_k.
struct IsoPrevOut { int* PV; unsigned char* FL; double* E; double* big; };      // [column][direction][n + 1]
template <int NTH, bool IN_LDS>
__global__ __launch_bounds__(NTH) void iso_prev_k(IsoCol a, int mode, IsoPrevOut o, const AdmmCtl* ctl) {
  CTL_GUARD(ctl);
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
  __shared__ double sc[NTH / 64];
  const int n = (int)a.rows, n1 = n + 1;
  const int r = blockIdx.x, dir = blockIdx.y;
  // prefix sums in LDS; beyond ~6800 rows in a global slice of this workgroup's own
  double* P = IN_LDS ? reinterpret_cast<double*>(dyn)
                     : o.big + (((size_t)r * gridDim.y + dir) * gridDim.z + blockIdx.z) * 3 * (size_t)n1;
  double* Q = P + n1;
  double* park = Q + n1;
  IsoBuf b{P, nullptr, nullptr, 0.0};
  const bool flip = mode == 2 && dir == 1;
  const double sign = mode == 1 ? -1.0 : 1.0;
  iso_prefix<NTH>(a.V + a.ldv * r, n, flip, sign, b, Q, park, sc);
  const size_t off = ((size_t)r * gridDim.y + dir) * n1;
  iso_prev<NTH>(n, mode == 2 && a.nonneg != 0.0, mode == 2, b, Q, blockIdx.z, gridDim.z, o.PV + off, o.FL + off, o.E + off);
}

// err(i) for every node: pointer doubling over the prev-tree.  Returns the buffer that holds the result.
template <int NTH>
__device__ double* iso_errors(int n, double* Ea, double* Eb, int* Ja, int* Jb) {
  for (int span = 1; span < n + 1; span <<= 1) {
    for (int v = threadIdx.x; v <= n; v += NTH) {
      const int j = Ja[v];
      Eb[v] = Ea[v] + Ea[j];
      Jb[v] = Ja[j];
    }
    __syncthreads();
    double* te = Ea; Ea = Eb; Eb = te;
    int* tj = Ja; Ja = Jb; Jb = tj;
  }
  return Ea;
}

// Writes the fit of the prefix of length B (B >= 1) of one direction: entry p < B takes the level of the first node of
// the path B -> prev(B) -> ... -> 0 that lies to its right.
template <int NTH>
__device__ void iso_write(int n, int B, const IsoBuf& b, int* Ja, int* Jb, unsigned char* M, bool flip, double sign,
                          double* z, int* sci) {
  constexpr int NW = NTH / 64;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  for (int v = t; v <= n; v += NTH) { Ja[v] = b.PV[v]; M[v] = (v == B) ? 1 : 0; }
  __syncthreads();
  for (int span = 1; span < B + 1; span <<= 1) {     // after round k the nodes up to 2^(k+1)-1 steps below B are marked
    for (int v = t; v <= B; v += NTH)
      if (M[v]) M[Ja[v]] = 1;
    __syncthreads();
    for (int v = t; v <= B; v += NTH) Jb[v] = Ja[Ja[v]];
    __syncthreads();
    int* tj = Ja; Ja = Jb; Jb = tj;
  }
  // first marked node > p for every position p < B: suffix minimum over chunks
  const int chunk = (B + NTH - 1) / NTH;
  const int c0 = min(B, t * chunk), c1 = min(B, c0 + chunk);
  int first = 0x7fffffff;                            // first marked node in (c0, c1]
  for (int v = c1; v > c0; --v)
    if (M[v]) first = v;
  int suf = first;                                   // inclusive suffix minimum over the threads
  for (int off = 1; off < 64; off <<= 1) { const int u = __shfl_down(suf, off); if (lane + off < 64) suf = min(suf, u); }
  if (lane == 0) sci[w] = suf;
  __syncthreads();
  int later = 0x7fffffff;                            // minimum over the later waves
#pragma unroll
  for (int k = 0; k < NW; ++k)
    if (k > w) later = min(later, sci[k]);
  int nxt = __shfl_down(suf, 1);                     // suffix minimum of the threads after this one
  if (lane == 63) nxt = 0x7fffffff;
  int cur = min(nxt, later);
  for (int p = c1 - 1; p >= c0; --p) {
    if (M[p + 1]) cur = p + 1;
    const int a = b.PV[cur];
    const double level = b.FL[cur] ? 0.0 : b.c + (b.P[cur] - b.P[a]) / (double)(cur - a);   // :76
    z[flip ? n - 1 - p : p] = sign * level;
  }
  __syncthreads();
}

// mode 0: non-decreasing, 1: non-increasing (-project_monotone(-x), :28), 2: unimodal
// `pre`: the nodes' prev / flag / cost were computed by iso_prev_k (null: this workgroup does it itself)
template <int NTH, bool IN_LDS>
__global__ __launch_bounds__(NTH) void prox_iso_k(IsoCol a, int mode, IsoPrevOut pre, const AdmmCtl* ctl) {
  CTL_GUARD(ctl);
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
  __shared__ double sc[NTH / 64];
  __shared__ int sci[NTH / 64];
  __shared__ double best_e[NTH / 64];
  __shared__ int best_i[NTH / 64];
  const int n = (int)a.rows, n1 = n + 1;
  const int r = blockIdx.x;
  const double* v = a.V + a.ldv * r;
  double* z = a.Z + a.ldz * r;
  // carve the arrays out of LDS or out of this column's slice of the global scratch
  unsigned char* base = IN_LDS ? dyn : reinterpret_cast<unsigned char*>(a.ws) + (size_t)r * iso_col_bytes(n);
  double* PL = reinterpret_cast<double*>(base);
  double* PR = PL + n1;
  double* Q = PR + n1;
  double* Ea = Q + n1;
  double* Eb = Ea + n1;
  double* ErrL = Eb + n1;
  int* PVL = reinterpret_cast<int*>(ErrL + n1);
  int* PVR = PVL + n1;
  int* Ja = PVR + n1;
  int* Jb = Ja + n1;
  unsigned char* FLL = reinterpret_cast<unsigned char*>(Jb + n1);
  unsigned char* FLR = FLL + n1;
  unsigned char* M = FLR + n1;
  const int ndir = mode == 2 ? 2 : 1;
  auto build = [&](IsoBuf& b, int dir, bool flip, double sign, bool nonneg, bool with_err) {
    iso_prefix<NTH>(v, n, flip, sign, b, Q, Ea, sc);
    if (pre.PV == nullptr) {
      iso_prev<NTH>(n, nonneg, with_err, b, Q, 0, 1, b.PV, b.FL, Ea);
    } else {
      const size_t off = ((size_t)r * ndir + dir) * n1;
      for (int i = 1 + threadIdx.x; i <= n; i += NTH) {
        b.PV[i] = pre.PV[off + i];
        b.FL[i] = pre.FL[off + i];
        if (with_err) Ea[i] = pre.E[off + i];
      }
    }
    if (threadIdx.x == 0) { b.PV[0] = 0; b.FL[0] = 0; Ea[0] = 0.0; }
    __syncthreads();
    if (with_err)                                                      // first jump of the error recurrence
      for (int i = threadIdx.x; i <= n; i += NTH) Ja[i] = b.FL[i] ? 0 : b.PV[i];
    __syncthreads();
  };
  if (mode != 2) {
    const double sign = mode == 1 ? -1.0 : 1.0;
    IsoBuf b{PL, PVL, FLL, 0.0};
    build(b, 0, false, sign, false, false);
    iso_write<NTH>(n, n, b, Ja, Jb, M, false, sign, z, sci);
    return;
  }
  const bool nonneg = a.nonneg != 0.0;
  IsoBuf bl{PL, PVL, FLL, 0.0}, br{PR, PVR, FLR, 0.0};
  build(bl, 0, false, 1.0, nonneg, true);
  {
    const double* e = iso_errors<NTH>(n, Ea, Eb, Ja, Jb);
    for (int i = threadIdx.x; i <= n; i += NTH) ErrL[i] = e[i];
    __syncthreads();
  }
  build(br, 1, true, 1.0, nonneg, true);
  const double* ErrR = iso_errors<NTH>(n, Ea, Eb, Ja, Jb);
  // split: i = 1 pairs nothing with the right fit of the whole column, i >= 2 the left prefix of length i with the
  // right prefix of length n-i+1 (:22-26); the first minimum wins
  // Splits whose criterion differs only by the rounding of the error sums are ties: the reference takes the first
  // minimum of ITS sums (accumulated pool by pool along each prefix), which on exactly tied data -- integers,
  // quantised measurements -- is decided by rounding noise no other summation order reproduces.  Here: the smallest i
  // among the splits within a few ulp of the minimum, i.e. the first minimum of the exact criterion.
  auto crit = [&](int i) { return i == 1 ? ErrR[n] : ErrL[i] + ErrR[n - i + 1]; };
  double be = INFINITY;
  for (int i = 1 + threadIdx.x; i <= n; i += NTH) be = fmin(be, crit(i));
  for (int off = 32; off > 0; off >>= 1) be = fmin(be, __shfl_xor(be, off));
  if ((threadIdx.x & 63) == 0) best_e[threadIdx.x >> 6] = be;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NTH / 64; ++k) be = fmin(be, best_e[k]);
  const double lim = be + 16.0 * 2.220446049250313e-16 * fabs(be);
  int bi = 0x7fffffff;
  for (int i = 1 + threadIdx.x; i <= n; i += NTH)
    if (crit(i) <= lim) { bi = i; break; }           // ascending i per thread
  for (int off = 32; off > 0; off >>= 1) bi = min(bi, __shfl_xor(bi, off));
  if ((threadIdx.x & 63) == 0) best_i[threadIdx.x >> 6] = bi;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NTH / 64; ++k) bi = min(bi, best_i[k]);
  const int split = bi;                              // same value in every thread
  __syncthreads();
  iso_write<NTH>(n, split, bl, Ja, Jb, M, false, 1.0, z, sci);                  // :15
  if (n - split >= 1) iso_write<NTH>(n, n - split, br, Ja, Jb, M, true, 1.0, z, sci);   // :16, written back flipped (:18)
}

static constexpr int kIsoLdsRows = 2048;             // 67 bytes per row: 137 KB of the 160 KB
static constexpr int kIsoOneKernelRows = 512;        // up to here one workgroup does the n^2/2 comparisons itself
static constexpr int kIsoPrevLdsRows = 6000;         // 3 double arrays: 144 KB
static int iso_shares(int R, int ndir) {
  int shares = 512 / (R * ndir);
  return shares < 1 ? 1 : (shares > 64 ? 64 : shares);
}
static size_t iso_pre_bytes(int64_t rows, int R) {   // iso_prev_k's output: 2 directions x (int + byte + double) per node
  if (rows <= kIsoOneKernelRows) return 0;
  size_t b = ((size_t)R * 2 * (rows + 1) * (sizeof(int) + 1 + sizeof(double)) + 255) / 256 * 256;
  if (rows > kIsoPrevLdsRows) b += (size_t)R * 2 * iso_shares(R, 1) * 3 * (rows + 1) * sizeof(double) + 256;   // its prefix sums
  return b;
}
size_t iso_ws_bytes(int64_t rows, int R) {
  return iso_pre_bytes(rows, R) + (rows > kIsoLdsRows ? (size_t)R * iso_col_bytes(rows) : 0) + 16;
}

void prox_iso(const double* V, int64_t ldv, double* Z, int64_t ldz, int64_t rows, int R, int mode, double nonneg,
              double* ws, const AdmmCtl* ctl, hipStream_t s) {
  AO_REQUIRE(rows < (int64_t)1 << 30, "isotonic projection: column too long");
  IsoCol a;
  a.V = V; a.Z = Z; a.ldv = ldv; a.ldz = ldz; a.rows = rows; a.R = R; a.nonneg = nonneg; a.ws = nullptr;
  IsoPrevOut pre{nullptr, nullptr, nullptr, nullptr};
  const int ndir = mode == 2 ? 2 : 1;
  if (rows > kIsoOneKernelRows) {
    AO_REQUIRE(ws != nullptr, "isotonic projection: no scratch");
    const size_t n1 = (size_t)rows + 1, cells = (size_t)R * 2 * n1;
    pre.E = ws;
    pre.PV = reinterpret_cast<int*>(ws + cells);
    pre.FL = reinterpret_cast<unsigned char*>(pre.PV + cells);
    a.ws = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(ws) + iso_pre_bytes(rows, R));
    const int shares = iso_shares(R, ndir);
    const dim3 grid((unsigned)R, (unsigned)ndir, (unsigned)shares);
    if (rows <= kIsoPrevLdsRows) {
      const size_t sh = 3 * n1 * sizeof(double);
      if (sh > 65536) ensure_dynamic_lds(reinterpret_cast<const void*>(iso_prev_k<256, true>), (int)(3 * (kIsoPrevLdsRows + 1) * sizeof(double)));
      iso_prev_k<256, true><<<grid, 256, sh, s>>>(a, mode, pre, ctl);
    } else {
      pre.big = reinterpret_cast<double*>(pre.FL + (cells + 7) / 8 * 8);
      iso_prev_k<256, false><<<grid, 256, 0, s>>>(a, mode, pre, ctl);
    }
    AO_KERNEL_CHECK();
  }
  const size_t bytes = iso_col_bytes(rows);
  if (rows <= 256) {
    prox_iso_k<256, true><<<R, 256, bytes, s>>>(a, mode, pre, ctl);
  } else if (rows <= kIsoLdsRows) {
    ensure_dynamic_lds(reinterpret_cast<const void*>(prox_iso_k<1024, true>), (int)iso_col_bytes(kIsoLdsRows));
    prox_iso_k<1024, true><<<R, 1024, bytes, s>>>(a, mode, pre, ctl);
  } else {
    prox_iso_k<1024, false><<<R, 1024, 0, s>>>(a, mode, pre, ctl);
  }
  AO_KERNEL_CHECK();
}

// ---------------------------------------------------------------------------
// GL smoothness: (2*eta/rho*L + I) x = v with L the Laplacian of a path (:68-76).  The matrix is tridiagonal and
// strictly diagonally dominant: parallel cyclic reduction, log2(n) rounds in which every equation eliminates its two
// neighbours at distance k, all four coefficient arrays in LDS, two barriers per round.
template <int NTH, int EPT>
__global__ __launch_bounds__(NTH) void prox_gl_pcr_k(IsoCol a, double eta, const double* rho, double rho_mul, const AdmmCtl* ctl) {
  CTL_GUARD(ctl);
  extern __shared__ __attribute__((aligned(16))) double gl[];
  const int n = (int)a.rows;
  double* A = gl;
  double* B = A + n;
  double* Cc = B + n;
  double* D = Cc + n;
  const int r = blockIdx.x, t = threadIdx.x;
  const double* v = a.V + a.ldv * r;
  double* z = a.Z + a.ldz * r;
  const double s2 = 2.0 * (eta / (rho[0] * rho_mul));
  double ra[EPT], rb[EPT], rc[EPT], rd[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int i = t + e * NTH;
    if (i < n) {
      const double deg = (n == 1) ? 1.0 : ((i == 0 || i == n - 1) ? 1.0 : 2.0);
      ra[e] = i > 0 ? -s2 : 0.0;
      rc[e] = i < n - 1 ? -s2 : 0.0;
      rb[e] = s2 * deg + 1.0;
      rd[e] = v[i];
      A[i] = ra[e]; B[i] = rb[e]; Cc[i] = rc[e]; D[i] = rd[e];
    }
  }
  __syncthreads();
  for (int k = 1; k < n; k <<= 1) {
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int i = t + e * NTH;
      if (i < n) {
        const int lo = i - k, hi = i + k;
        double nb = rb[e], nd = rd[e], na = 0.0, nc = 0.0;
        if (lo >= 0) {
          const double al = -ra[e] / B[lo];
          nb += al * Cc[lo]; nd += al * D[lo]; na = al * A[lo];
        }
        if (hi < n) {
          const double ga = -rc[e] / B[hi];
          nb += ga * A[hi]; nd += ga * D[hi]; nc = ga * Cc[hi];
        }
        ra[e] = na; rb[e] = nb; rc[e] = nc; rd[e] = nd;
      }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int i = t + e * NTH;
      if (i < n) { A[i] = ra[e]; B[i] = rb[e]; Cc[i] = rc[e]; D[i] = rd[e]; }
    }
    __syncthreads();
  }
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int i = t + e * NTH;
    if (i < n) z[i] = rd[e] / rb[e];
  }
}

// The same reduction for columns beyond the LDS-resident 4096 rows: the four coefficient arrays ping-pong between two
// copies in the prox workspace (8 doubles per row and column, L2-resident), one barrier per round; the arithmetic
// of an equation is that of prox_gl_pcr_k, statement for statement, so the result does not depend on the path.
template <int NTH>
__global__ __launch_bounds__(NTH) void prox_gl_pcr_long_k(IsoCol a, double eta, const double* rho, double rho_mul,
                                                          const AdmmCtl* ctl) {
  CTL_GUARD(ctl);
  const int n = (int)a.rows;
  const int r = blockIdx.x, t = threadIdx.x;
  double* w = a.ws + (size_t)r * 8 * n;
  const double* v = a.V + a.ldv * r;
  double* z = a.Z + a.ldz * r;
  const double s2 = 2.0 * (eta / (rho[0] * rho_mul));
  for (int i = t; i < n; i += NTH) {
    const double deg = (n == 1) ? 1.0 : ((i == 0 || i == n - 1) ? 1.0 : 2.0);
    w[i] = i > 0 ? -s2 : 0.0;
    w[n + i] = s2 * deg + 1.0;
    w[2 * n + i] = i < n - 1 ? -s2 : 0.0;
    w[3 * n + i] = v[i];
  }
  __syncthreads();
  int cur = 0;
  for (int k = 1; k < n; k <<= 1) {
    const double* A = w + (size_t)cur * 4 * n;
    const double* B = A + n;
    const double* Cc = B + n;
    const double* D = Cc + n;
    double* X = w + (size_t)(cur ^ 1) * 4 * n;
    for (int i = t; i < n; i += NTH) {
      const int lo = i - k, hi = i + k;
      const double ra = A[i], rc = Cc[i];
      double nb = B[i], nd = D[i], na = 0.0, nc = 0.0;
      if (lo >= 0) {
        const double al = -ra / B[lo];
        nb += al * Cc[lo]; nd += al * D[lo]; na = al * A[lo];
      }
      if (hi < n) {
        const double ga = -rc / B[hi];
        nb += ga * A[hi]; nd += ga * D[hi]; nc = ga * Cc[hi];
      }
      X[i] = na; X[n + i] = nb; X[2 * n + i] = nc; X[3 * n + i] = nd;
    }
    __syncthreads();
    cur ^= 1;
  }
  const double* B = w + (size_t)cur * 4 * n + n;
  const double* D = B + 2 * (size_t)n;
  for (int i = t; i < n; i += NTH) z[i] = D[i] / B[i];
}

size_t prox_gl_ws_doubles(int64_t rows, int R) { return rows > 4096 ? (size_t)8 * rows * R : (size_t)rows * R; }

bool prox_gl_pcr(const double* V, int64_t ldv, double* Z, int64_t ldz, int64_t rows, int R, double eta, const double* rho,
                 double rho_mul, const AdmmCtl* ctl, hipStream_t s, double* ws) {
  IsoCol a;
  a.V = V; a.Z = Z; a.ldv = ldv; a.ldz = ldz; a.rows = rows; a.R = R; a.nonneg = 0; a.ws = ws;
  if (rows > 4096) {
    if (ws == nullptr || rows >= (int64_t(1) << 28)) return false;   // no workspace: the caller solves sequentially
    prox_gl_pcr_long_k<1024><<<R, 1024, 0, s>>>(a, eta, rho, rho_mul, ctl);
    AO_KERNEL_CHECK();
    return true;
  }
  const size_t sh = (size_t)4 * rows * sizeof(double);
  if (rows <= 256) {
    prox_gl_pcr_k<256, 1><<<R, 256, sh, s>>>(a, eta, rho, rho_mul, ctl);
  } else if (rows <= 1024) {
    prox_gl_pcr_k<1024, 1><<<R, 1024, sh, s>>>(a, eta, rho, rho_mul, ctl);
  } else {
    ensure_dynamic_lds(reinterpret_cast<const void*>(prox_gl_pcr_k<1024, 4>), (int)(4 * 4096 * sizeof(double)));
    prox_gl_pcr_k<1024, 4><<<R, 1024, sh, s>>>(a, eta, rho, rho_mul, ctl);
  }
  AO_KERNEL_CHECK();
  return true;
}

}  // namespace aoadmm
