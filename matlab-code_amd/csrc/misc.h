// Data movement / generation helpers: padded upload, transposition, synthetic
// tensors generated in HBM, norms and regulariser values.
#pragma once
#include "common.h"
#include "contract.h"
#include "small.h"

namespace aoadmm {

// dst (padded, prec) <- src (device fp64, unpadded column-major rows x cols), for a column range
void pad_convert(void* dst, int prec, int64_t pad_rows, const double* src, int64_t rows, int64_t cols,
                 int64_t dst_col0, hipStream_t s);
// dst (padded pad_c x rows, prec) = transpose of src (device fp64 rows x cols)
void transpose_convert(void* dst, int prec, int64_t pad_c, const double* src, int64_t rows, int64_t cols,
                       hipStream_t s);
// sum of squares of a padded tensor (padding is zero) -> slot (fp64), deterministic
void tensor_sumsq(double* slot, const void* X, int prec, int64_t n, double* ws, hipStream_t s);

// synthetic CP data: element (i,j,k) of the local block with global row i+row0:
//   clean = sum_r A(i,r) B(j,r) C(k,r), A/B/C ~ U[0,1) from a counter-based generator;
//   noise ~ N(0,1) from the same generator family.
// pass 1 (acc != null): accumulate sum clean^2 and sum noise^2 into acc[0], acc[1] partial buffers;
// pass 2: write X = clean + sigma*noise.
struct SynthArgs {
  int64_t I_loc, I_pad, J, K, row0, I_full;
  int R;
  uint64_t seed;
  int64_t k0 = 0, K_loc = -1;    // third-mode slab [k0, k0 + K_loc) of the block (K_loc < 0: all of it)
};
void synth_factors(double* A, double* B, double* C, const SynthArgs& a, hipStream_t s);   // fp64 col-major, full sizes
void synth_norms(double* out3, const double* A, const double* B, const double* C, const SynthArgs& a,
                 double* ws, hipStream_t s);
void synth_write(void* X, int prec, const double* A, const double* B, const double* C, const SynthArgs& a,
                 double sigma, double inv_norm, hipStream_t s);
size_t synth_ws_bytes();

// regulariser value of constraints_to_prox.m (reg_func) on a factor matrix -> slot
void reg_value(double* slot, int type, double p0, const double* X, int64_t rows, int R, double* ws,
               hipStream_t s);

// Pass-specific resident copy of a 3-way tensor in the row-blocked layout of misc.hip (which = 0, 1, 2: the copy for
// the pass that contracts mode 3, 1, 2).  D holds round_up(M, kRowBlockElems) * C elements.  false: a mode is too long
// for one launch (the caller keeps the natural layout).
constexpr int64_t kRowBlockElems = 512;
bool block_layout_copy(const void* X, void* D, int which, int prec, int64_t I, int64_t Ip, int64_t J, int64_t K,
                       int64_t Tp, hipStream_t s);
// Y = X_(n) X_(n)' of a resident dense block (cmtf_nvecs.m:56): row a of the unfolding at X + a*sa, reduction
// over t1 < n1 (stride s1) x t2 < n2 (stride s2); Y is n x n column-major fp64.
struct UnfoldGramArgs {
  const void* X;
  int64_t n, sa, n1, s1, n2, s2;
};
size_t unfold_gram_ws_bytes(const UnfoldGramArgs& a);
void unfold_gram(const UnfoldGramArgs& a, int prec, double* ws, double* out, hipStream_t s);

// out(a + A*b, r) = Fa(a, r) * Fb(b, r)  (Khatri-Rao of two merged trailing modes, tensor storage order)
void kr_merge(double* out, const double* Fa, int64_t lda, int64_t A, const double* Fb, int64_t ldb, int64_t B, int R,
              hipStream_t s);

}  // namespace aoadmm
