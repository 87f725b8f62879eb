// EM missing-data imputation (functions/cmtf_fun_AOADMM.m:408-441) and the masked objective
// (:1224-1226, :1249-1252) as one pass over a data block:
//   for every entry   m = model value from the current factors
//   observed  (mask 1): obs_res += (x - m)^2 ; obs_x2 += x^2
//   missing   (mask 0): num += (m - x)^2 ; den += x^2 ; x <- m   (update mode only)
// out4 = {num, den, obs_res, obs_x2}, fixed summation order.
#pragma once
#include "common.h"

namespace aoadmm {

struct EmCpArgs {
  void* X;                 // float / double, first dimension padded to `Ipad`
  const uint8_t* mask;     // ONE BIT per entry of the same layout (em_mask_pack), 1 = observed
  const double *A, *B, *C; // factors, column-major; A points at the block's first local row; C null for matrices
  int64_t ldA, ldB, ldC;
  int64_t I, Ipad, J, K;   // local rows, padded rows, second mode, third mode (1 for matrices)
  int R;
  int update;              // 0: statistics only, 1: also overwrite the missing entries
  // 1 (default): a workgroup walks the second mode at a fixed third-mode index; 2: it walks the third mode at a
  // fixed second-mode index (3-way blocks only)
  int walk = 1;
  // Fused tensor pass (null: none).  T[chunk][i + Ipad*f][r] = sum over the chunk's part of the walked mode of
  // x_new(i, w, f) * W(w, r), in the tensor's precision: what launch_contract would leave for a plan that contracts the
  // walked mode (ContractPlan in contract.h; chunks = em_cp_fused_chunks, t_chunk_stride = rows of T times R).
  void* T = nullptr;
  int64_t t_chunk_stride = 0;
};
// bits[e >> 3] bit (e & 7) = bytes[e] != 0 for e < n; room for cdiv(n, 8) + 16 bytes (the strip kernel's clamped
// look-ahead loads stay inside the block, the slack is for the padding of the last byte)
void em_mask_pack(const uint8_t* bytes, uint8_t* bits, int64_t n, hipStream_t s);
inline size_t em_mask_bits_bytes(int64_t n) { return (size_t)((n + 7) / 8 + 16); }
constexpr int kEmFuseMaxRank = 24;
bool em_cp_can_fuse(const EmCpArgs& a, int prec);
int em_cp_fused_chunks(const EmCpArgs& a, int prec);
size_t em_cp_ws_bytes(int64_t Ipad, int64_t J, int64_t K);
// ws: em_cp_ws_bytes ; out4: device, 4 doubles
void em_cp_pass(const EmCpArgs& a, int prec, double* ws, double* out4, hipStream_t s);

struct EmPar2Args {
  double* X;               // slabs back to back (I x J_k each), fp64
  const uint8_t* mask;     // same layout
  const double *A, *B, *C; // A: I x R ; B: slabs J_k x R back to back ; C: K x R (ldC = K)
  const int64_t* off;      // device, K+1
  int K, I, R;
  int update;
};
// ws: 4*K doubles
void em_par2_pass(const EmPar2Args& a, double* ws, double* out4, hipStream_t s);

}  // namespace aoadmm
