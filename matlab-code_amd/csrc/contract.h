// Tensor-pass kernels: partial contraction of a dense tensor with one factor
// matrix on the matrix cores, and the small reductions that finish an MTTKRP.
#pragma once
#include "common.h"

namespace aoadmm {

// A dense block resident in HBM, first dimension padded (zeros) to `pad0`
// elements so that every column of every unfolding starts 16-byte aligned.
struct DenseTensor {
  int prec = AOADMM_PREC_F64;
  int nd = 0;
  int64_t dims[8] = {0};   // logical sizes (local sizes when row-sharded)
  int64_t pad0 = 0;        // padded first dimension (multiple of 4 for f32, 2 for f64)
  DevBuf data;             // pad0 * dims[1] * ... elements of float/double
  int64_t elems_padded() const {
    int64_t n = pad0;
    for (int i = 1; i < nd; ++i) n *= dims[i];
    return n;
  }
  size_t elem_size() const { return prec == AOADMM_PREC_F32 ? 4 : 8; }
};

// Description of one batched "contiguous-M" contraction
//   T[b][m][r] = sum_{c<C} X[b*batch_stride + m + ld*c] * F[c][r]
// (m contiguous in memory).  T is row-major [nchunk][nbatch*M][R] in the tensor's own precision: an
// fp32 contraction accumulates in fp32 inside the MFMA, so storing its result as fp32 loses nothing and
// halves the bytes the reductions read back; chunk sums are added in fp64 by the reductions.
struct ContractPlan {
  int64_t nbatch, batch_stride, M, ld, C;
  int R;
  int tprec = AOADMM_PREC_F64;   // element type of T
  int nchunk;          // split of the reduction (bounds f32 accumulation length)
  bool lead = false;   // true: the contracted mode is the contiguous one (X[c + ld*m]), LDS-transposed kernel
  bool on_xp = false;  // the pass runs on the mode-permuted second copy of the tensor (solver.h CpBlock::Xp)
  bool on_xq = false;  // ... on the third copy Xq(k,i,j) (mode 2 trailing)
  bool on_xc = false;  // ... on the row-blocked copy of X itself (mode 3 trailing)
  int64_t trows() const { return nbatch * M; }
  size_t t_bytes() const { return (size_t)nchunk * trows() * R * (tprec == AOADMM_PREC_F32 ? 4 : 8); }
  size_t frag_bytes(int prec) const;
  double algorithmic_bytes(int prec) const {   // tensor read once + T written once
    return (double)nbatch * M * C * (prec == AOADMM_PREC_F32 ? 4.0 : 8.0) + (double)t_bytes();
  }
  double flops() const { return 2.0 * nbatch * M * C * R; }
};

extern int g_contract_variant;   // development switch (see contract.hip); -1 = read AOADMM_CONTRACT_VARIANT

ContractPlan make_plan(int64_t nbatch, int64_t batch_stride, int64_t M, int64_t ld, int64_t C, int R,
                       int prec);

// T(m,r) = sum_c X[c + ld*m] * F(c,r): contraction of the leading (contiguous) mode, fp32 tensors
ContractPlan make_lead_plan(int64_t M, int64_t ld, int64_t C, int R);

// F: device fp64, column-major (C x R) with leading dimension ldF.
void launch_contract(const void* X, int prec, const ContractPlan& pl, const double* F, int64_t ldF,
                     void* frag_ws, void* T, hipStream_t s, hipEvent_t ev0 = nullptr,
                     hipEvent_t ev1 = nullptr);   // events bracket the contraction kernel only

// out(b,r) = scale * sum_a sum_chunk T[chunk][a + Apad*b][r] * Fa(a,r)      (a < A)
// ft_scratch: reduce_factor_scratch_bytes(A, R) (row-major copy of the factor)
size_t reduce_factor_scratch_bytes(int64_t rows, int R);
// `sys` (both reductions, optional): the R x R system build of the mode whose MTTKRP this reduction finishes rides in the
// launch as one extra workgroup (small_dev.h: it needs Gram matrices only, so it runs beside the reduction instead of
// behind it).  Returns true when it did (R <= 20); otherwise the caller launches sys_build itself.
struct SysBuild;
bool launch_reduce_inner(const void* T, int tprec, int nchunk, int64_t trows, int64_t A, int64_t Apad,
                         int64_t B, int R, const double* Fa, int64_t ldFa, double scale,
                         double* out, int64_t ldOut, double* ft_scratch, hipStream_t s,
                         const double* FaT = nullptr,    // FaT: row-major copy of Fa if the caller keeps one
                         int rowmajor_out = 0,           // 1: out is [b][r] (T layout, for a further fold)
                         const SysBuild* sys = nullptr);
// out(a,r) = scale * sum_b sum_chunk T[chunk][a + Apad*b][r] * Fb(b,r)      (a < A)
// scratch must hold reduce_outer_scratch_bytes(); ft_scratch reduce_factor_scratch_bytes(B, R).
size_t reduce_outer_scratch_bytes(int64_t A, int64_t B, int R);
bool launch_reduce_outer(const void* T, int tprec, int nchunk, int64_t trows, int64_t A, int64_t Apad,
                         int64_t B, int R, const double* Fb, int64_t ldFb, double scale,
                         double* out, int64_t ldOut, double* scratch, double* ft_scratch, hipStream_t s,
                         const double* FbT = nullptr, int rowmajor_out = 0, const SysBuild* sys = nullptr);
// out(a,r) = scale * sum_chunk T[chunk][a][r]     (matrix blocks: nothing left to reduce)
void launch_t_to_colmajor(const void* T, int tprec, int nchunk, int64_t trows, int64_t A, int R, double scale,
                          double* out, int64_t ldOut, hipStream_t s);

// A whole MTTKRP of a tiny block in one launch (contract.hip small_mttkrp_k): out(n, r) = scale * sum_{a,b}
// X[n*sn + a*sa + b*sb] * Fa(a, r) * Fb(b, r).  Fb null with Nb = 1: matrices.
constexpr int64_t kSmallMttkrpElems = (int64_t)1 << 19;
struct SmallMttkrp {
  const void* X;
  int64_t sn, sa, sb;        // element strides of the target mode and of the two other modes
  int Na, Nb, R;
  const double *Fa, *Fb;     // column-major factors of the other modes
  int64_t lda, ldb;
  double scale;
  double* out;
  int64_t ldOut;
};
bool small_mttkrp_ok(int64_t elems, int nd, const int64_t* dims, int R);
// `sys`: as for the reductions over T -- the mode's system build as one extra workgroup; true when it rode
bool small_mttkrp(const SmallMttkrp& a, int prec, int64_t rows, hipStream_t s, const SysBuild* sys = nullptr);

}  // namespace aoadmm
