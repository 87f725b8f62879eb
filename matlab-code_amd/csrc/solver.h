// Host-side engine: model (struct Z), state (struct G) and the AO-ADMM outer loop
// (functions/cmtf_fun_AOADMM.m:87-476) driving the HIP kernels.  Everything stays
// resident in HBM; the host synchronises once per outer iteration to read the
// objective values and the inner-iteration counters.
#pragma once
#include <atomic>
#include <memory>
#include <mutex>
#include <vector>

#include "admm.h"
#include "common.h"
#include "contract.h"
#include "misc.h"
#include "par2.h"
#include "small.h"

typedef struct ncclComm* ncclComm_t;

namespace aoadmm {

struct FactorRef {
  const double* p;    // device, column-major
  int64_t ld;
  uint64_t version;
  const double* pT = nullptr;   // optional row-major copy (rows x R) of the same version, written by the Gram kernel
};

// One dense CP block (tensor or matrix) and its partial-contraction cache.
struct CpBlock {
  DenseTensor X;       // natural layout, first dimension padded
  DenseTensor Xt;      // matrices only: transposed copy (second mode contiguous)
  // 3-way tensors: optional second copy Xp(j,k,i) = X(i,j,k) (leading dimension Jp), built on first use, so that
  // the pass that contracts mode 1 streams like the others (the tensor's size again in HBM; 288 GB per GPU)
  DevBuf Xp;
  DevBuf emkr, emkr2;  // order > 3 with Z.miss: Khatri-Rao factor of the merged trailing modes (ping-pong)
  int64_t Jp = 0;
  bool has_xp = false, xp_refused = false;
  // With a communicator the copy for the pass that contracts mode 1 is sharded along mode 3 instead of mode 1:
  // rank g holds X(:, :, K_g), contracts ALL of mode 1 and gets a complete T(j, k in K_g, r) of 1/N the size, instead
  // of a partial sum of full size J x K from its rows of mode 1.
  bool xp_ksharded = false;
  int64_t xp_k0 = 0, xp_kloc = 0;
  // third copy Xq(k,i,j) = X(i,j,k) (leading dimension Kp): the pass that contracts mode 2 then streams like the
  // other two instead of running K batches of an I x J matrix (measured 6.0 ms against 5.4-5.5 ms at 2000^3)
  DevBuf Xq;
  int64_t Kp = 0;
  bool has_xq = false, xq_refused = false;
  // and a copy for the pass that contracts mode 3: same row order as X, but row-blocked like the other two
  // (misc.hip block_layout_copy); all three copies are stored in that layout
  DevBuf Xc;
  bool has_xc = false, xc_refused = false;
  int nd = 0;
  int64_t dims[8] = {0};   // local sizes (dims[0] = local rows when sharded)
  int64_t full0 = 0;       // global size of the first mode
  int64_t row0 = 0;        // first local row of the first mode
  bool has_data = false;
  // The natural-layout array is only read to build the three pass copies, for ||X||^2, the EM pass and the fallbacks; once
  // all three copies exist (and no mask does) it can go: 4 -> 3 resident copies (Engine::maybe_release_natural)
  bool x_released = false;
  // dimension-tree cache: T = X x_c F_c, valid while factor c keeps `cached_version`
  int cached_mode = -1;
  uint64_t cached_version = 0;
  ContractPlan plan;
  DevBuf T, frag, scratch, ft, tmpA, tmpB;
  // sharded MTTKRP outputs that are this rank's ROWS of the result (mode 1; mode 3 under xp_ksharded): send buffers whose
  // other rows are zero for good (cleared once), all-reduced out of place into the caller's buffer
  DevBuf own[2];
  size_t own_bytes[2] = {0, 0};
  int64_t own_row0[2] = {-1, -1};
  // Z.miss{p}: one byte per entry in the layout of X (and of Xt for matrices), 1 = observed
  DevBuf mask, maskT;
  bool has_mask = false;
};

struct ModeInfo {
  bool defined = false;
  int64_t rows = 0;
  int R = 0;
  bool slabs = false;
  int K = 0;
  std::vector<int64_t> rows_k, off_k;
  int tensor = -1, pos = -1;
  int coupling = -1;
  bool constrained = false;
  ProxSpec prox;
  DevBuf H, H2, Ht, H2t;   // coupling transformation matrices and their transposes
  DevBuf HHt;              // coupling type 2: H*H' (R x R)
  DevBuf eU, eUt, eLam;    // coupling types 1/5: H'*H = eU diag(eLam) eU' (rows x rows, host Jacobi once per model)
  DevBuf eV, eMu;          // coupling types 1/5: eigendecomposition of the R x R system matrix (every outer iteration)
  std::vector<double> H_host;
  int64_t img_rows = 0, img_cols = 0;   // shape of the factor-side coupling image (= shape of coupling_dual_fac)
  QuadPrep quad;           // quadratic regularization: L and its eigendecomposition
  int64_t hr = 0, hc = 0, h2r = 0, h2c = 0;
  double ridge = 0.0;
  DevBuf fac, Z, mu, muD;
  DevBuf facT;             // row-major copy of fac (by-product of the Gram kernel), valid while facT_version == version
  uint64_t facT_version = 0;
  bool has_fac = false, has_Z = false, has_mu = false, has_muD = false;
  int64_t muD_rows = 0, muD_cols = 0;
  uint64_t version = 1;
  // work buffers
  DevBuf A, Ab, gram, C, Bsys, L, Binv, rho, Zold, V, Znew, part, proxws, RHS, TD, TF, tmp, W1, W2;
  const double* Aeff = nullptr;
};

// PARAFAC2 block: K ragged slabs X_k (I x J_k) and the block's internal coupling variables
struct Par2Block {
  int K = 0, I = 0, R = 0;
  std::vector<int64_t> off_h;     // K+1 prefix sums of J_k
  DevBuf off_d;
  int64_t Jtot = 0;
  int Jmax = 0;
  DevBuf X;                       // slabs back to back, fp64
  DevBuf mask;                    // Z.miss{p}{k}: one byte per entry, same layout, 1 = observed
  bool has_mask = false;
  std::vector<char> have_slab;
  DevBuf DeltaB, DeltaBold, P, Pold, muDB;        // state (G.DeltaB, G.P, G.mu_DeltaB)
  bool has_DeltaB = false;
  std::vector<char> have_P, have_mu;
  DevBuf W, T1, GB, Ak, Lk, rhok, part, norms, res, q, regv, Csys, ac, Lc, rhoc, rhomax;
  DevBuf Jrot;                                     // [K][R*R]: last Jacobi rotation per slab (warm start inside a B_k loop)
  // slab sharding over the ranks of a communicator (aoadmm_options.par2_slab_sharding): this rank runs the per-slab
  // kernels on [k0, k1) only; every sum over k is all-reduced, slab-valued state is gathered when the solve ends
  bool slab_sharded = false;
  int k0 = 0, k1 = 0;
  DevBuf psum;                    // R*R+1 partial sums of DeltaB, then 4 residual means
  // coupled C mode: sum(rho_k); H'H, the (K*R)^2 system and its inverse for coupling type 1 (:282-297)
  DevBuf rhosum, HtH, Mbig, Minv, Hs;   // Hs = diag(rho)*H for coupling type 3
  bool have_HtH = false, hth_diag = false;   // hth_diag: H'H is diagonal, HtH holds its K diagonal entries
  P2Dims dims() const {
    P2Dims d;
    d.K = K; d.I = I; d.R = R; d.off = off_d.as<int64_t>(); d.off_h = off_h.data(); d.Jtot = Jtot; d.Jmax = Jmax;
    d.k0 = slab_sharded ? k0 : 0; d.k1 = slab_sharded ? k1 : K;
    return d;
  }
  P2Dims dims_all() const {       // every slab, whatever the sharding (replicated C-mode loop)
    P2Dims d = dims();
    d.k0 = 0; d.k1 = K;
    return d;
  }
};

struct TensorInfo {
  bool defined = false;
  bool par2 = false;
  Par2Block p2;
  int nmodes = 0;
  int modes[8] = {0};
  double weight = 1.0;
  CpBlock blk;
  double normsq = 0.0;
  bool normsq_valid = false;
  int last_pos = -1;
  bool eval_shortcut = false;   // PARAFAC2: the enqueued objective evaluation took the last_mttkrp shortcut (:1254-1260)
};

struct CouplingInfo {
  int type = -1;
  std::vector<int> modes;
  DevBuf Delta, DeltaOld, BB, AA, LAA, dD, tmp, coef, rho_ptrs;
  std::vector<const double*> rho_ptrs_host;
  int64_t rows = 0, cols = 0;
  bool has_state = false;
};

struct KernelStats {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  double ms = 0.0, bytes = 0.0, flops = 0.0;
  int64_t launches = 0, timed = 0;      // timed <= launches: launches bracketed by an event pair
};

struct LocalGroup;

class Engine {
 public:
  explicit Engine(int device);
  ~Engine();

  // model
  void model_begin(int n_modes, int n_tensors, int n_couplings);
  void set_mode(int mode, int64_t rows, int rank);
  void set_mode_slabs(int mode, int K, const int64_t* rows_k, int rank);
  void add_cp(int p, int n, const int* modes, double weight);
  void add_par2(int p, const int* modes3, double weight);
  void set_constraint(int mode, int type, const double* params, int np, const double* Lmat);
  void set_coupling(int mode, int coupling, const double* H, int64_t hr, int64_t hc, const double* H2,
                    int64_t h2r, int64_t h2c);
  void set_coupling_type(int coupling, int type);
  void set_ridge(const double* ridge);
  void model_end();

  // data
  void tensor_upload(int p, const double* data, int prec, int64_t row0, int64_t local_rows);
  void tensor_synth(int p, int rank, uint64_t seed, double noise, int prec);
  void par2_slab_upload(int p, int k, const double* Xk);
  void tensor_mask_upload(int p, const uint8_t* mask);
  void par2_slab_mask_upload(int p, int k, const uint8_t* mask);
  double tensor_normsq(int p);

  // state
  void state_set(int field, int index, int slab, const double* host, int64_t rows, int64_t cols);
  void state_get(int field, int index, int slab, double* host, int64_t rows, int64_t cols);

  // solve
  void solve(const aoadmm_options& opt, aoadmm_result* out);
  void resident_mttkrp(int p, int pos, double* out_host, float* ms);
  void resident_unfold_gram(int p, int pos, int slab, double* out_host);
  void kernel_stats(int which, int reset, double* ms, int64_t* launches, double* bytes, double* flops);

  // communicator
  void comm_init(const char id[128], int rank, int world, bool share_only = false);
  void comm_init_local(int key, int rank, int world);
  // Unblocks this engine's collectives after a failure on ANOTHER rank of the same process (aoadmm_create_multi):
  // callable from a foreign thread while the engine's own thread waits inside a collective; the next collective
  // throws AOADMM_ERR_RCCL.  The communicator is gone afterwards.
  void comm_abort();
  void comm_info(int* nccl_version, int* comm_ranks, char* lib_path, int cap) const;
  void set_progress(aoadmm_progress_fn fn, void* user, int every) { progress_fn_ = fn; progress_user_ = user; progress_every_ = every; }
  void par2_gather_slabs(TensorInfo& t);
  // (world_ > 1 stays true after an abort took the communicator away: the engine must not fall back to unsharded work)
  bool sharded() const { return world_ > 1 || comm_ != nullptr || local_ != nullptr; }
  void require_usable() const;          // throws AOADMM_ERR_RCCL once comm_abort() has run (sticky)
  bool share_only() const { return share_only_; }
  // tiny unsharded block: MTTKRP by the one-launch kernel instead of contraction pass + reduction
  bool small_direct(const CpBlock& b, int R) const { return !sharded() && small_mttkrp_ok(b.X.elems_padded(), b.nd, b.dims, R); }
  int rank() const { return rank_; }
  int world() const { return world_; }

  hipStream_t stream() const { return stream_; }
  int device() const { return device_; }

  // MTTKRP of a dense block against factors (device), result scale*mttkrp into out (ld = ldOut)
  void ensure_contraction(CpBlock& b, int pos, const FactorRef* facs, int R, bool use_cache, const int* update_seq,
                          int nseq);
  bool ensure_permuted_copy(CpBlock& b);
  bool ensure_permuted_copy2(CpBlock& b);
  bool ensure_blocked_copy(CpBlock& b);
  // builds the mode-3-sharded Xp from a natural-layout slab X(:, :, [k0, k0 + kloc)) already on the device
  void adopt_ksharded_xp(CpBlock& b, const void* slab, int64_t k0, int64_t kloc);
  bool want_ksharded_xp(const CpBlock& b, int64_t K, int64_t* k0, int64_t* kloc) const;
  void drop_permuted_copies(CpBlock& b);
  void maybe_release_natural(TensorInfo& t);
  bool prefetch_next_contraction(const aoadmm_options& opt);   // true: a tensor pass was enqueued
  // `collective` = false: the block holds the whole tensor and the result is complete on this engine (op-level
  // entry on an engine that happens to belong to a communicator)
  // `tensor_pass` = true: always the tensor-pass kernels, also for blocks small enough for the one-launch kernel
  // (the op-level entries, so that their parity tests exercise the pass kernels at every size)
  void block_mttkrp(CpBlock& b, int pos, const FactorRef* facs, int R, double scale, double* out,
                    int64_t ldOut, bool use_cache, const int* update_seq, int nseq, bool collective = true,
                    bool tensor_pass = false, const SysBuild* sys = nullptr, bool* sys_done = nullptr);
  // `full_array`: the caller's whole tensor when it holds one (lets a sharded engine take its mode-3 slab as well)
  void block_upload(CpBlock& b, int nd, const int64_t* dims, const double* host, int prec, int64_t row0,
                    int64_t local_rows, const double* full_array = nullptr);
  void allreduce(double* buf, int64_t n);
  void allreduce_from(const double* send, double* recv, int64_t n);   // out of place (send == recv: in place)
  double* scratch_slots() { return slots_.d(); }
  double* red_ws() { return redws_.d(); }

 private:
  void check_mode(int m) const;
  void compute_gram(ModeInfo& mi, const LoopEnd* close = nullptr);
  FactorRef factor_ref(const ModeInfo& o) const {
    return FactorRef{o.fac.d(), o.rows, o.version, o.facT_version == o.version ? o.facT.d() : nullptr};
  }
  void update_uncoupled_cp_mode(int m, const aoadmm_options& opt);
  void prepare_mode_system(int m, int nrho, const aoadmm_options& opt);
  void coupled_admm(int c, const aoadmm_options& opt);
  void eval_objective_enqueue(bool first);
  bool has_missing() const;
  void em_pass_enqueue(int p, int update, bool fuse_next_pass = false);
  void prepare_next_first_mode(const aoadmm_options& opt);         // statistics of tensor p into its EM slots (+ imputation)
  double* em_slot(int p) const;
  void ensure_mode_work(ModeInfo& mi);
  // PARAFAC2 (solver_par2.hip)
  void par2_ensure_work(TensorInfo& t);
  void par2_prepare_modeA(int m, int nrho, const aoadmm_options& opt);
  void par2_update_B(int m, const aoadmm_options& opt, int iter);
  void par2_update_C(int m, const aoadmm_options& opt);
  void par2_prepare_C_coupled(int m, int ctype, const aoadmm_options& opt);
  void par2_objective_enqueue(TensorInfo& t);
  double* resid_slots(int m);
  std::vector<int> update_sequence(int p) const;

  int device_ = 0;
  hipStream_t stream_ = nullptr;
  hipStream_t side_ = nullptr;        // objective evaluation beside the prefetched tensor pass (Engine::solve)
  hipEvent_t side_ev_ = nullptr;
  int n_modes_ = 0, n_tensors_ = 0, n_couplings_ = 0;
  bool model_done_ = false;
  bool has_ridge_ = false;
  std::vector<ModeInfo> modes_;
  std::vector<TensorInfo> tensors_;
  std::vector<CouplingInfo> couplings_;
  DevBuf readback_;      // everything the host reads once per outer iteration, in one piece (one copy): slots_ | ctls_ |
                         // per PARAFAC2 block res (K + 1), q (4 K), regv (K); the three below are views into it
  DevBuf ctls_;          // AdmmCtl[n_modes + n_couplings]
  DevBuf slots_;         // objective scalars
  DevBuf redws_;         // reduction workspace
  DevBuf ones_;          // a device 1.0 (unit weight where a kernel expects a rho pointer)
  DevBuf emws_;          // EM pass partial sums
  bool allow_xp_ = true;  // options.hip.no_permuted_copy
  DevBuf atbws_;
  DevBuf staging_;
  std::vector<hipEvent_t> event_pool_;   // timing events are recycled: creating two per tensor pass cost host time in the loop
  hipEvent_t take_event();
  void fold_finished(KernelStats& ks);
  KernelStats kstats_[3];   // [0] streaming contraction, [1] leading-mode contraction, [2] reductions over T
  int prepared_mode_ = -1;  // mode whose MTTKRP + system build were enqueued ahead (prepare_next_first_mode)
  bool profile_ = true;
  bool profile_reductions_ = false;   // switched on by the first kernel_stats(2, ...) call: two more events per reduction
  ncclComm_t comm_ = nullptr;
  mutable std::mutex comm_mu_;          // comm_ / aborted_ against comm_abort() from another worker thread
  std::atomic<bool> aborted_{false};
  std::shared_ptr<LocalGroup> local_;   // process-local group (threads of one process), see solver.hip
  int rank_ = 0, world_ = 1;
  bool share_only_ = false;             // aoadmm_comm_init_rank_share: rank_/world_ of an N-rank job on a one-rank communicator
  aoadmm_progress_fn progress_fn_ = nullptr;   // options.Display = 'iter'
  void* progress_user_ = nullptr;
  int progress_every_ = 0;

  AdmmCtl* ctl_of_mode(int m) { return ctls_.as<AdmmCtl>() + m; }
  AdmmCtl* ctl_of_coupling(int c) { return ctls_.as<AdmmCtl>() + n_modes_ + c; }
  void timed_contract(const void* X, int prec, const ContractPlan& pl, const double* F, int64_t ldF,
                      void* frag, void* T);
};

}  // namespace aoadmm
