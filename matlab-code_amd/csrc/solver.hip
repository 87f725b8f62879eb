// Engine implementation: see solver.h.  Reference control flow:
// functions/cmtf_fun_AOADMM.m:87-476 (outer loop), :625-695 / :904-983 (coupled ADMM
// cases 0 and 4), :1213-1363 (objective), functions/evaluate_stopping_conditions.m.
#include "solver.h"
#include "device_utils.h"
#include "em.h"
#include "hosteig.h"
#include "prox_dev.h"

#include <rccl/rccl.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <set>

namespace aoadmm {

#define AO_NCCL(expr)                                                                          \
  do {                                                                                         \
    ncclResult_t r__ = (expr);                                                                 \
    if (r__ != ncclSuccess)                                                                    \
      throw Error(AOADMM_ERR_RCCL, fmt("%s failed: %s", #expr, ncclGetErrorString(r__)));      \
  } while (0)

static constexpr int kSlotsPerMode = 8;      // objective slots
static constexpr int kResidPerMode = 8;      // ADMM residual slots

Engine::Engine(int device) : device_(device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    throw Error(AOADMM_ERR_HIP, "no HIP device available: this library has no CPU fallback");
  if (device < 0 || device >= n) throw Error(AOADMM_ERR_INVALID, fmt("device %d out of range [0,%d)", device, n));
  AO_HIP(hipSetDevice(device));
  AO_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  AO_HIP(hipStreamCreateWithFlags(&side_, hipStreamNonBlocking));
  AO_HIP(hipEventCreateWithFlags(&side_ev_, hipEventDisableTiming));
  redws_.alloc(4096 * sizeof(double));
  ones_.alloc(sizeof(double));
  const double one = 1.0;
  AO_HIP(hipMemcpy(ones_.p, &one, sizeof one, hipMemcpyHostToDevice));
}

Engine::~Engine() {
  (void)hipSetDevice(device_);
  for (auto& ks : kstats_)
    for (auto& pr : ks.pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  for (hipEvent_t e : event_pool_) (void)hipEventDestroy(e);
  if (comm_) (void)ncclCommDestroy(comm_);
  if (side_ev_) (void)hipEventDestroy(side_ev_);
  if (side_) (void)hipStreamDestroy(side_);
  if (stream_) (void)hipStreamDestroy(stream_);
}

// ---------------------------------------------------------------------------
// communicator
// ---------------------------------------------------------------------------
void Engine::comm_init(const char id[128], int rank, int world, bool share_only) {
  AO_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad rank/world %d/%d", rank, world);
  AO_REQUIRE(id != nullptr || world == 1, "a communicator of %d ranks needs the id from aoadmm_comm_unique_id", world);
  AO_HIP(hipSetDevice(device_));
  if (comm_) { (void)ncclCommDestroy(comm_); comm_ = nullptr; }
  local_.reset();
  if (id != nullptr) {               // world == 1 with an id: one-rank communicator (exercises the RCCL path on one GPU)
    ncclUniqueId uid;
    static_assert(sizeof(uid) <= 128, "unique id larger than the ABI buffer");
    std::memcpy(&uid, id, sizeof(uid));
    // share_only (aoadmm_comm_init_rank_share): this engine takes rank `rank` of `world` in every sharding decision
    // but its communicator has ONE rank, so the collectives run (ncclAllReduce on the library's stream) without
    // peers: one rank's share of an N-GPU job, timed on a one-GPU box.  The sums are this rank's partial sums only.
    if (share_only) AO_NCCL(ncclCommInitRank(&comm_, 1, uid, 0));
    else AO_NCCL(ncclCommInitRank(&comm_, world, uid, rank));
  }
  rank_ = rank;
  world_ = world;
  share_only_ = share_only;
  aborted_ = false;
}

// Process-local group: engines driven by threads of ONE process (on one device or several) meet at a
// mutex/condvar barrier and add their buffers through host staging, in rank order, so every rank gets the same
// bits.  It exists so the sharded data path (row blocks, own-rows buffers, objective partial sums) can be run with
// world > 1 on a one-GPU box, where RCCL refuses two ranks on one device.  Not a transport for production: the
// data crosses PCIe twice per collective.
struct LocalGroup {
  std::mutex m;
  std::condition_variable cv;
  int world = 0, arrived = 0, joined = 0;
  uint64_t gen = 0;
  bool aborted = false;                       // a rank failed outside the collectives: nobody waits for it any more
  std::vector<std::vector<double>> stage;     // one host buffer per rank
  void barrier() {
    std::unique_lock<std::mutex> lk(m);
    if (aborted) throw Error(AOADMM_ERR_RCCL, "local group: aborted after a failure on another rank");
    const uint64_t g = gen;
    if (++arrived == world) {
      arrived = 0;
      ++gen;
      cv.notify_all();
      return;
    }
    if (!cv.wait_for(lk, std::chrono::seconds(120), [&] { return gen != g || aborted; }))
      throw Error(AOADMM_ERR_RCCL, "local group: a rank did not reach the collective within 120 s");
    if (gen == g) throw Error(AOADMM_ERR_RCCL, "local group: aborted after a failure on another rank");
  }
  void abort() {
    std::lock_guard<std::mutex> lk(m);
    aborted = true;
    cv.notify_all();
  }
};
static std::mutex g_groups_mutex;
static std::map<int, std::shared_ptr<LocalGroup>> g_groups;

void Engine::comm_init_local(int key, int rank, int world) {
  AO_REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad rank/world %d/%d", rank, world);
  if (comm_) { (void)ncclCommDestroy(comm_); comm_ = nullptr; }
  std::lock_guard<std::mutex> lk(g_groups_mutex);
  std::shared_ptr<LocalGroup>& g = g_groups[key];
  if (!g || g->joined == g->world) {           // first rank of a new (or re-used) key
    g = std::make_shared<LocalGroup>();
    g->world = world;
    g->stage.resize(world);
  }
  AO_REQUIRE(g->world == world, "local group %d was created for %d ranks, not %d", key, g->world, world);
  g->joined++;
  local_ = g;
  rank_ = rank;
  world_ = world;
  aborted_ = false;
}

void Engine::comm_abort() {
  aborted_ = true;                              // sticky: every later collective, solve or upload of this engine throws
  if (local_) local_->abort();
  ncclComm_t c = nullptr;
  {
    std::lock_guard<std::mutex> lk(comm_mu_);
    c = comm_;
    comm_ = nullptr;
  }
  // outside the lock: the owner thread may sit inside ncclAllReduce's enqueue with a copy of the handle
  if (c) (void)ncclCommAbort(c);                // the collective kernels of this rank see the flag and exit
}

void Engine::require_usable() const {
  if (aborted_) throw Error(AOADMM_ERR_RCCL, "context unusable: its communicator was aborted after a failure on another rank");
}

void Engine::comm_info(int* nccl_version, int* comm_ranks, char* lib_path, int cap) const {
  if (nccl_version) {
    int v = 0;
    AO_NCCL(ncclGetVersion(&v));
    *nccl_version = v;
  }
  if (comm_ranks) {
    int n = local_ ? world_ : 0;
    std::lock_guard<std::mutex> lk(comm_mu_);
    if (comm_) AO_NCCL(ncclCommCount(comm_, &n));
    *comm_ranks = n;
  }
  if (lib_path && cap > 0) {
    lib_path[0] = 0;
    Dl_info di;
    if (dladdr(reinterpret_cast<const void*>(&ncclGetVersion), &di) && di.dli_fname) {
      std::strncpy(lib_path, di.dli_fname, (size_t)cap - 1);
      lib_path[cap - 1] = 0;
    }
  }
}

void Engine::allreduce(double* buf, int64_t n) { allreduce_from(buf, buf, n); }

// recv = sum over ranks of send (send == recv: in place)
void Engine::allreduce_from(const double* send, double* buf, int64_t n) {
  if (n <= 0) return;
  if (aborted_) throw Error(AOADMM_ERR_RCCL, "communicator aborted after a failure on another rank");
  if (local_) {
    LocalGroup& g = *local_;
    std::vector<double>& mine = g.stage[rank_];
    mine.resize((size_t)n);
    AO_HIP(hipMemcpyAsync(mine.data(), send, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, stream_));
    AO_HIP(hipStreamSynchronize(stream_));
    g.barrier();                                // every rank has staged its contribution
    std::vector<double> tot((size_t)n, 0.0);
    for (int r = 0; r < g.world; ++r) {
      AO_REQUIRE((int64_t)g.stage[r].size() == n, "local group: rank %d brought %lld values, rank %d brought %lld", r,
                 (long long)g.stage[r].size(), rank_, (long long)n);
      for (int64_t i = 0; i < n; ++i) tot[i] += g.stage[r][i];
    }
    g.barrier();                                // every rank has read all contributions
    AO_HIP(hipMemcpyAsync(buf, tot.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, stream_));
    AO_HIP(hipStreamSynchronize(stream_));
    return;
  }
  ncclComm_t c = nullptr;
  {
    std::lock_guard<std::mutex> lk(comm_mu_);
    if (aborted_) throw Error(AOADMM_ERR_RCCL, "communicator aborted after a failure on another rank");
    c = comm_;
  }
  if (!c) {
    // a sharded engine without a transport would go on with its partial sums: never silently
    if (world_ > 1) throw Error(AOADMM_ERR_RCCL, fmt("rank %d of %d has no communicator", rank_, world_));
    if (send != buf) AO_HIP(hipMemcpyAsync(buf, send, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, stream_));
    return;
  }
  // enqueued outside the lock so that comm_abort() from the caller's thread never waits behind a stuck enqueue
  AO_NCCL(ncclAllReduce(send, buf, (size_t)n, ncclDouble, ncclSum, c, stream_));
}

// ---------------------------------------------------------------------------
// model
// ---------------------------------------------------------------------------
void Engine::check_mode(int m) const {
  AO_REQUIRE(m >= 0 && m < n_modes_, "mode %d out of range [0,%d)", m, n_modes_);
}

void Engine::model_begin(int n_modes, int n_tensors, int n_couplings) {
  AO_REQUIRE(n_modes > 0 && n_tensors > 0 && n_couplings >= 0, "model_begin: bad counts");
  AO_HIP(hipSetDevice(device_));
  n_modes_ = n_modes; n_tensors_ = n_tensors; n_couplings_ = n_couplings;
  modes_.clear(); tensors_.clear(); couplings_.clear();
  modes_.resize(n_modes); tensors_.resize(n_tensors); couplings_.resize(n_couplings);
  model_done_ = false;
  has_ridge_ = false;
}

void Engine::set_mode(int mode, int64_t rows, int rank) {
  check_mode(mode);
  AO_REQUIRE(rows > 0 && rank > 0 && rank <= kMaxRank, "mode %d: rows=%lld rank=%d invalid (rank <= %d)", mode,
             (long long)rows, rank, kMaxRank);
  ModeInfo& mi = modes_[mode];
  mi.defined = true; mi.rows = rows; mi.R = rank; mi.slabs = false;
}

void Engine::set_mode_slabs(int mode, int K, const int64_t* rows_k, int rank) {
  check_mode(mode);
  AO_REQUIRE(K > 0 && rank > 0 && rank <= kMaxRank, "slab mode %d: bad K/rank", mode);
  ModeInfo& mi = modes_[mode];
  mi.defined = true; mi.slabs = true; mi.K = K; mi.R = rank;
  mi.rows_k.assign(rows_k, rows_k + K);
  mi.off_k.assign(K + 1, 0);
  for (int k = 0; k < K; ++k) {
    AO_REQUIRE(rows_k[k] > 0, "slab %d has no rows", k);
    mi.off_k[k + 1] = mi.off_k[k] + rows_k[k];
  }
  mi.rows = mi.off_k[K];
}

void Engine::add_cp(int p, int n, const int* modes, double weight) {
  AO_REQUIRE(p >= 0 && p < n_tensors_, "tensor %d out of range", p);
  AO_REQUIRE(n >= 2 && n <= 8, "CP block needs 2..8 modes");
  TensorInfo& t = tensors_[p];
  t.defined = true; t.par2 = false; t.nmodes = n; t.weight = weight;
  for (int i = 0; i < n; ++i) {
    check_mode(modes[i]);
    AO_REQUIRE(modes_[modes[i]].defined && !modes_[modes[i]].slabs, "mode %d undefined or slab-valued", modes[i]);
    AO_REQUIRE(modes_[modes[i]].tensor < 0, "mode %d already belongs to tensor %d", modes[i], modes_[modes[i]].tensor);
    t.modes[i] = modes[i];
    modes_[modes[i]].tensor = p;
    modes_[modes[i]].pos = i;
    AO_REQUIRE(modes_[modes[i]].R == modes_[modes[0]].R, "modes of tensor %d disagree on the rank", p);
  }
}

void Engine::set_constraint(int mode, int type, const double* params, int np, const double* Lmat) {
  check_mode(mode);
  ModeInfo& mi = modes_[mode];
  AO_REQUIRE(type >= AOADMM_C_NONE && type <= AOADMM_C_TPARAFAC2, "unknown constraint id %d", type);
  mi.constrained = type != AOADMM_C_NONE;
  mi.prox = ProxSpec();
  mi.prox.type = type;
  if (np > 0) mi.prox.p0 = params[0];
  if (np > 1) mi.prox.p1 = params[1];
  auto need = [&](int n) { AO_REQUIRE(np >= n, "constraint %d on mode %d needs %d parameter(s)", type, mode, n); };
  switch (type) {
    case AOADMM_C_BOX: need(2); break;
    case AOADMM_C_SIMPLEX_COL: case AOADMM_C_SIMPLEX_ROW: case AOADMM_C_UNIMODAL: case AOADMM_C_L1_BALL:
    case AOADMM_C_L2_BALL: case AOADMM_C_NONNEG_L2_BALL: case AOADMM_C_L1_REG: case AOADMM_C_L0_REG:
    case AOADMM_C_L2_REG: case AOADMM_C_RIDGE: case AOADMM_C_GL_SMOOTH: case AOADMM_C_TV: need(1); break;
    case AOADMM_C_TPARAFAC2: need(1); break;
    case AOADMM_C_QUADRATIC:
      need(1);
      AO_REQUIRE(mi.defined, "quadratic regularization: define the mode before its constraint");
      if (mi.slabs) throw Error(AOADMM_ERR_UNSUPPORTED, "quadratic regularization on the PARAFAC2 B_k mode is not in the device path");
      AO_REQUIRE(Lmat != nullptr, "quadratic regularization on mode %d needs its matrix L (constraints{m}{3})", mode + 1);
      AO_HIP(hipSetDevice(device_));
      mi.quad.build(Lmat, mi.rows, stream_);
      mi.quad.attach(mi.prox);
      break;
    default: break;
  }
}

static void upload_small(DevBuf& b, const double* host, int64_t n, hipStream_t s) {
  b.ensure((size_t)n * sizeof(double));
  AO_HIP(hipMemcpyAsync(b.p, host, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
  AO_HIP(hipStreamSynchronize(s));
}

static void upload_with_transpose(DevBuf& b, DevBuf& bt, const double* host, int64_t r, int64_t c, hipStream_t s) {
  std::vector<double> t((size_t)r * c);
  for (int64_t j = 0; j < c; ++j)
    for (int64_t i = 0; i < r; ++i) t[(size_t)j + (size_t)c * i] = host[(size_t)i + (size_t)r * j];
  upload_small(b, host, r * c, s);
  upload_small(bt, t.data(), r * c, s);
}

void Engine::set_coupling(int mode, int coupling, const double* H, int64_t hr, int64_t hc, const double* H2,
                          int64_t h2r, int64_t h2c) {
  check_mode(mode);
  AO_REQUIRE(coupling >= -1 && coupling < n_couplings_, "coupling id %d out of range", coupling);
  ModeInfo& mi = modes_[mode];
  mi.coupling = coupling;
  mi.hr = mi.hc = mi.h2r = mi.h2c = 0;
  mi.H_host.clear();
  if (H && hr > 0 && hc > 0) {
    upload_with_transpose(mi.H, mi.Ht, H, hr, hc, stream_);
    mi.hr = hr; mi.hc = hc;
    mi.H_host.assign(H, H + (size_t)hr * hc);
  }
  if (H2 && h2r > 0 && h2c > 0) { upload_with_transpose(mi.H2, mi.H2t, H2, h2r, h2c, stream_); mi.h2r = h2r; mi.h2c = h2c; }
}

void Engine::set_coupling_type(int coupling, int type) {
  AO_REQUIRE(coupling >= 0 && coupling < n_couplings_, "coupling id %d out of range", coupling);
  AO_REQUIRE(type >= 0 && type <= 5, "coupling type %d invalid", type);
  couplings_[coupling].type = type;
}

void Engine::set_ridge(const double* ridge) {
  has_ridge_ = ridge != nullptr;
  for (int m = 0; m < n_modes_; ++m) modes_[m].ridge = ridge ? ridge[m] : 0.0;
}

void Engine::model_end() {
  for (int m = 0; m < n_modes_; ++m) {
    AO_REQUIRE(modes_[m].defined, "mode %d has no size", m);
    AO_REQUIRE(modes_[m].tensor >= 0, "mode %d belongs to no tensor (Mismatch between size and modes inputs)", m);
  }
  for (int p = 0; p < n_tensors_; ++p) AO_REQUIRE(tensors_[p].defined, "tensor %d undefined", p);
  for (int m = 0; m < n_modes_; ++m) {
    const ModeInfo& mi = modes_[m];
    if (mi.constrained && mi.prox.type == AOADMM_C_TPARAFAC2)       // cmtf_AOADMM.m:33-41
      AO_REQUIRE(tensors_[mi.tensor].par2 && mi.pos == 1, "The tPARAFAC2 constraint can only be impsed on the second mode of a PARAFAC2 model");
    if (!tensors_[mi.tensor].par2) continue;
    if (mi.pos == 1 && mi.constrained && mi.prox.type == AOADMM_C_TPARAFAC2)
      for (int k = 1; k < mi.K; ++k)
        AO_REQUIRE(mi.rows_k[k] == mi.rows_k[0], "tPARAFAC2 needs slabs of equal size (t_smoothness_prox.m adds B_k matrices)");
    if (mi.pos == 1) {
      // check_data_input.m:33-35
      AO_REQUIRE(mi.coupling < 0, "Coupling in 2. mode (the varying mode) of Parafac2 decomposition not supported.");
    }
  }
  for (int c = 0; c < n_couplings_; ++c) {
    CouplingInfo& ci = couplings_[c];
    AO_REQUIRE(ci.type >= 0, "coupling %d has no type (Mismatch between number of couplings and coupling types)", c);
    ci.modes.clear();
    for (int m = 0; m < n_modes_; ++m)
      if (modes_[m].coupling == c) ci.modes.push_back(m);
    AO_REQUIRE(!ci.modes.empty(), "coupling %d couples no mode", c);
    AO_REQUIRE(ci.modes.size() <= 8, "more than 8 modes in one coupling");
    for (int m : ci.modes)
      if (tensors_[modes_[m].tensor].par2 && modes_[m].pos == 2 && ci.type == 5)
        AO_REQUIRE(modes_[m].hr <= modes_[m].rows, "coupling type 5 of a PARAFAC2 C mode: Delta has more rows than the mode (cmtf_fun_AOADMM.m:1049-1051 indexes rho by Delta's row)");
    if (ci.type == 4 || ci.type == 5) {       // :945-961, :1034-1052 keep one PARAFAC2 term apart (AAA); two would overwrite each other
      int npc = 0;
      for (int m : ci.modes) npc += (tensors_[modes_[m].tensor].par2 && modes_[m].pos == 2) ? 1 : 0;
      if (npc > 1) throw Error(AOADMM_ERR_UNSUPPORTED, fmt("coupling type %d with more than one PARAFAC2 C mode is not supported", ci.type));
    }
    const ModeInfo& m0 = modes_[ci.modes[0]];
    auto need_H = [&](int m) { AO_REQUIRE(modes_[m].hr > 0, "Coupling matrix for mode %d is missing.", m + 1); };
    switch (ci.type) {
      case 0:                                   // C = Delta  (check_data_input.m:48-61)
        ci.rows = m0.rows; ci.cols = m0.R;
        for (int m : ci.modes) {
          AO_REQUIRE(modes_[m].rows == m0.rows, "Coupled factor matrices of mode %d and mode %d need to have same number of rows.", ci.modes[0] + 1, m + 1);
          AO_REQUIRE(modes_[m].R == m0.R, "Coupled factor matrices of mode %d and mode %d need to have same number of components/columns.", ci.modes[0] + 1, m + 1);
          modes_[m].img_rows = modes_[m].rows; modes_[m].img_cols = modes_[m].R;
        }
        break;
      case 1:                                   // H*C = Delta : H is (rows_Delta x rows_m)  (:62-80)
        need_H(ci.modes[0]);
        ci.rows = m0.hr; ci.cols = m0.R;
        for (int m : ci.modes) {
          need_H(m);
          AO_REQUIRE(modes_[m].hc == modes_[m].rows, "Mismatch between sz and number of columns of coupling matrix for mode %d.", m + 1);
          AO_REQUIRE(modes_[m].hr == ci.rows, "Coupling transformation matrices need to have same number of rows for mode %d and mode %d.", ci.modes[0] + 1, m + 1);
          AO_REQUIRE(modes_[m].R == m0.R, "Coupled factor matrices of mode %d and mode %d need to have same number of components/columns.", ci.modes[0] + 1, m + 1);
          modes_[m].img_rows = ci.rows; modes_[m].img_cols = modes_[m].R;
        }
        break;
      case 2:                                   // C*H = Delta : H is (R_m x cols_Delta)  (:81-98)
        need_H(ci.modes[0]);
        ci.rows = m0.rows; ci.cols = m0.hc;
        for (int m : ci.modes) {
          need_H(m);
          AO_REQUIRE(modes_[m].hr == modes_[m].R, "Mismatch between number of components and number of rows of coupling matrix for mode %d.", m + 1);
          AO_REQUIRE(modes_[m].hc == ci.cols, "Coupling transformation matrices need to have same number of columns for mode %d and mode %d.", ci.modes[0] + 1, m + 1);
          AO_REQUIRE(modes_[m].rows == ci.rows, "Coupled factor matrices of mode %d and mode %d need to have same number of rows.", ci.modes[0] + 1, m + 1);
          modes_[m].img_rows = ci.rows; modes_[m].img_cols = ci.cols;
        }
        AO_REQUIRE(ci.cols <= kMaxRank, "coupling type 2: Delta has more than %d columns", kMaxRank);
        break;
      case 3:                                   // C = H*Delta : H is (rows_m x rows_Delta)  (:99-114)
        need_H(ci.modes[0]);
        ci.rows = m0.hc; ci.cols = m0.R;
        for (int m : ci.modes) {
          need_H(m);
          AO_REQUIRE(modes_[m].hr == modes_[m].rows, "Mismatch between sz and number of rows of coupling matrix for mode %d.", m + 1);
          AO_REQUIRE(modes_[m].hc == ci.rows, "Coupling transformation matrices need to have same number of columns for mode %d and mode %d.", ci.modes[0] + 1, m + 1);
          AO_REQUIRE(modes_[m].R == m0.R, "Coupled factor matrices of mode %d and mode %d need to have same number of components/columns.", ci.modes[0] + 1, m + 1);
          modes_[m].img_rows = modes_[m].rows; modes_[m].img_cols = modes_[m].R;
        }
        break;
      case 4:                                   // C = Delta*H : H is (cols_Delta x R_m)
        need_H(ci.modes[0]);
        ci.rows = m0.rows; ci.cols = m0.hr;
        for (int m : ci.modes) {
          need_H(m);
          AO_REQUIRE(modes_[m].rows == ci.rows && modes_[m].hr == ci.cols && modes_[m].hc == modes_[m].R,
                     "coupling type 4: transformation matrix of mode %d has the wrong shape", m + 1);
          modes_[m].img_rows = modes_[m].rows; modes_[m].img_cols = modes_[m].R;
        }
        AO_REQUIRE(ci.cols <= kMaxRank, "coupling type 4: Delta has more than %d columns", kMaxRank);
        break;
      default:                                  // 5: H*C = Delta*H2 : H (rows_Delta x rows_m), H2 (cols_Delta x R_m)  (:125-140)
        need_H(ci.modes[0]);
        AO_REQUIRE(m0.h2r > 0, "Coupling matrix H2 for mode %d is missing.", ci.modes[0] + 1);
        ci.rows = m0.hr; ci.cols = m0.h2r;
        for (int m : ci.modes) {
          need_H(m);
          AO_REQUIRE(modes_[m].h2r > 0, "Coupling matrix H2 for mode %d is missing.", m + 1);
          AO_REQUIRE(modes_[m].hc == modes_[m].rows && modes_[m].hr == ci.rows && modes_[m].h2r == ci.cols &&
                     modes_[m].h2c == modes_[m].R, "coupling type 5: transformation matrices of mode %d have the wrong shape", m + 1);
          modes_[m].img_rows = ci.rows; modes_[m].img_cols = modes_[m].R;
        }
        AO_REQUIRE(ci.cols <= kMaxRank, "coupling type 5: Delta has more than %d columns", kMaxRank);
        break;
    }
    for (int m : ci.modes) {
      ModeInfo& mi = modes_[m];
      if (ci.type == 2) {                       // H*H' (R x R) for the system matrix (:314)
        std::vector<double> hh((size_t)mi.R * mi.R, 0.0);
        for (int a = 0; a < mi.R; ++a)
          for (int b2 = 0; b2 < mi.R; ++b2) {
            double acc = 0.0;
            for (int64_t c2 = 0; c2 < mi.hc; ++c2) acc += mi.H_host[(size_t)a + (size_t)mi.hr * c2] * mi.H_host[(size_t)b2 + (size_t)mi.hr * c2];
            hh[(size_t)a + (size_t)mi.R * b2] = acc;
          }
        upload_small(mi.HHt, hh.data(), (int64_t)mi.R * mi.R, stream_);
      }
      if (ci.type == 1 || ci.type == 5) {       // H'*H = U diag(lam) U' once: the Sylvester solves reuse it (:288, :377)
        const int64_t n = mi.rows;
        if (n > 4096) throw Error(AOADMM_ERR_UNSUPPORTED, "coupling types 1/5: modes beyond 4096 rows are not diagonalised on the host");
        std::vector<double> hth((size_t)n * n, 0.0), lam, U;
        for (int64_t a = 0; a < n; ++a)
          for (int64_t b2 = a; b2 < n; ++b2) {
            double acc = 0.0;
            for (int64_t q = 0; q < mi.hr; ++q) acc += mi.H_host[(size_t)q + (size_t)mi.hr * a] * mi.H_host[(size_t)q + (size_t)mi.hr * b2];
            hth[(size_t)a + (size_t)n * b2] = acc; hth[(size_t)b2 + (size_t)n * a] = acc;
          }
        AO_REQUIRE(host_sym_eig(n, hth, lam, U) >= 0, "eigendecomposition of H'*H (mode %d) did not converge", m + 1);
        std::vector<double> Ut((size_t)n * n);
        for (int64_t j = 0; j < n; ++j)
          for (int64_t i = 0; i < n; ++i) Ut[(size_t)j + (size_t)n * i] = U[(size_t)i + (size_t)n * j];
        upload_small(mi.eU, U.data(), n * n, stream_);
        upload_small(mi.eUt, Ut.data(), n * n, stream_);
        upload_small(mi.eLam, lam.data(), n, stream_);
      }
    }
  }
  {
    // one arena for what the host reads back per outer iteration, so that one copy fetches it
    const size_t nsl = (size_t)(n_modes_ * (kSlotsPerMode + kResidPerMode) + 2 * n_tensors_ + 16 + 4 * n_tensors_) * sizeof(double);
    const size_t nct = (size_t)(n_modes_ + n_couplings_ + 1) * sizeof(AdmmCtl);
    const size_t off_ctl = (size_t)round_up((int64_t)nsl, 64);
    size_t tot = (size_t)round_up((int64_t)(off_ctl + nct), 64);
    std::vector<size_t> off_p2(n_tensors_, 0);
    for (int p = 0; p < n_tensors_; ++p)
      if (tensors_[p].par2) { off_p2[p] = tot; tot += ((size_t)6 * tensors_[p].p2.K + 1) * sizeof(double); }
    for (int p = 0; p < n_tensors_; ++p)            // views of the previous arena go before it does
      if (tensors_[p].par2) { tensors_[p].p2.res.release(); tensors_[p].p2.q.release(); tensors_[p].p2.regv.release(); }
    ctls_.release(); slots_.release();
    readback_.alloc(tot);
    AO_HIP(hipMemsetAsync(readback_.p, 0, readback_.bytes, stream_));
    char* base = readback_.as<char>();
    slots_.view(base, nsl);
    ctls_.view(base + off_ctl, nct);
    for (int p = 0; p < n_tensors_; ++p) {
      if (!tensors_[p].par2) continue;
      Par2Block& b = tensors_[p].p2;
      char* q = base + off_p2[p];
      b.res.view(q, (size_t)(b.K + 1) * 8);
      b.q.view(q + (size_t)(b.K + 1) * 8, (size_t)b.K * 4 * 8);
      b.regv.view(q + (size_t)(5 * b.K + 1) * 8, (size_t)b.K * 8);
    }
  }
  AO_HIP(hipStreamSynchronize(stream_));
  model_done_ = true;
}

// ---------------------------------------------------------------------------
// data
// ---------------------------------------------------------------------------
static int64_t pad_of(int prec, int64_t n) { return round_up(n, prec == AOADMM_PREC_F32 ? 4 : 2); }
static size_t blocked_bytes(int64_t M, int64_t C, size_t es) { return (size_t)round_up(M, kRowBlockElems) * C * es; }

void Engine::block_upload(CpBlock& b, int nd, const int64_t* dims, const double* host, int prec, int64_t row0,
                          int64_t local_rows, const double* full_array) {
  AO_REQUIRE(nd >= 2 && nd <= 8, "tensor order %d unsupported", nd);
  AO_REQUIRE(prec == AOADMM_PREC_F64 || prec == AOADMM_PREC_F32, "bad precision id %d", prec);
  AO_REQUIRE(row0 >= 0 && local_rows > 0 && row0 + local_rows <= dims[0], "bad row block [%lld,+%lld) of %lld",
             (long long)row0, (long long)local_rows, (long long)dims[0]);
  AO_HIP(hipSetDevice(device_));
  b.nd = nd;
  b.full0 = dims[0];
  b.row0 = row0;
  b.dims[0] = local_rows;
  int64_t ncols = 1;
  for (int i = 1; i < nd; ++i) { b.dims[i] = dims[i]; ncols *= dims[i]; }
  b.X.prec = prec; b.X.nd = nd;
  for (int i = 0; i < nd; ++i) b.X.dims[i] = b.dims[i];
  b.X.pad0 = pad_of(prec, local_rows);
  b.X.data.alloc((size_t)b.X.elems_padded() * b.X.elem_size());
  // host block layout: local_rows x ncols column-major (the caller extracted its rows)
  const int64_t chunk_cols = std::max<int64_t>(1, (int64_t)(64ll << 20) / local_rows);   // ~512 MB of doubles
  staging_.ensure((size_t)std::min(chunk_cols, ncols) * local_rows * sizeof(double));
  for (int64_t c0 = 0; c0 < ncols; c0 += chunk_cols) {
    const int64_t nc = std::min(chunk_cols, ncols - c0);
    AO_HIP(hipMemcpyAsync(staging_.p, host + c0 * local_rows, (size_t)nc * local_rows * sizeof(double),
                          hipMemcpyHostToDevice, stream_));
    pad_convert(b.X.data.p, prec, b.X.pad0, staging_.d(), local_rows, nc, c0, stream_);
    AO_HIP(hipStreamSynchronize(stream_));
  }
  if (nd == 2) {
    // transposed copy for the second mode (matrices are small next to tensors)
    b.Xt.prec = prec; b.Xt.nd = 2;
    b.Xt.dims[0] = dims[1]; b.Xt.dims[1] = local_rows;
    b.Xt.pad0 = pad_of(prec, dims[1]);
    b.Xt.data.alloc((size_t)b.Xt.pad0 * local_rows * b.Xt.elem_size());
    AO_REQUIRE(ncols * local_rows <= (int64_t)(1ll << 28), "matrix block too large for the transposed copy");
    staging_.ensure((size_t)ncols * local_rows * sizeof(double));
    AO_HIP(hipMemcpyAsync(staging_.p, host, (size_t)ncols * local_rows * sizeof(double), hipMemcpyHostToDevice, stream_));
    transpose_convert(b.Xt.data.p, prec, b.Xt.pad0, staging_.d(), local_rows, ncols, stream_);
    AO_HIP(hipStreamSynchronize(stream_));
  }
  b.has_data = true;
  b.x_released = false;
  b.cached_mode = -1;
  b.has_xp = false; b.xp_refused = false; b.has_xq = false; b.xq_refused = false; b.has_xc = false; b.xc_refused = false;
  b.xp_ksharded = false;
  if (nd == 3 && full_array != nullptr) {             // the caller holds the whole tensor: mode-3 slab for the mode-1 pass
    int64_t k0 = 0, kloc = 0;
    if (want_ksharded_xp(b, dims[2], &k0, &kloc)) {
      const int64_t I = dims[0], J = dims[1], Ipf = pad_of(prec, I);
      DevBuf slab;
      slab.alloc((size_t)Ipf * J * kloc * b.X.elem_size());
      const double* src = full_array + (size_t)I * J * k0;      // X(:, :, k0 : k0+kloc) is contiguous
      const int64_t ncs = J * kloc;
      const int64_t cc = std::max<int64_t>(1, (int64_t)(64ll << 20) / I);
      staging_.ensure((size_t)std::min(cc, ncs) * I * sizeof(double));
      for (int64_t c0 = 0; c0 < ncs; c0 += cc) {
        const int64_t nc = std::min(cc, ncs - c0);
        AO_HIP(hipMemcpyAsync(staging_.p, src + c0 * I, (size_t)nc * I * sizeof(double), hipMemcpyHostToDevice, stream_));
        pad_convert(slab.p, prec, Ipf, staging_.d(), I, nc, c0, stream_);
        AO_HIP(hipStreamSynchronize(stream_));
      }
      adopt_ksharded_xp(b, slab.p, k0, kloc);
      AO_HIP(hipStreamSynchronize(stream_));          // slab is a local
    }
  }
  if (nd == 3) { (void)ensure_permuted_copy2(b); (void)ensure_blocked_copy(b); }   // one-off set-up cost belongs to the upload
}

void Engine::tensor_upload(int p, const double* data, int prec, int64_t row0, int64_t local_rows) {
  require_usable();
  AO_REQUIRE(model_done_, "call aoadmm_model_end first");
  AO_REQUIRE(p >= 0 && p < n_tensors_, "tensor %d out of range", p);
  TensorInfo& t = tensors_[p];
  AO_REQUIRE(!t.par2, "tensor %d is PARAFAC2: use aoadmm_par2_slab_upload", p);
  int64_t dims[8];
  for (int i = 0; i < t.nmodes; ++i) dims[i] = modes_[t.modes[i]].rows;
  if (local_rows < 0) {            // full array given: every rank keeps its block of rows
    int64_t I = dims[0];
    int64_t per = cdiv(I, world_);
    row0 = std::min<int64_t>(I, per * rank_);
    local_rows = std::min<int64_t>(I, row0 + per) - row0;
    // the same verdict on every rank (a rank that throws alone leaves the others waiting in the next collective)
    AO_REQUIRE(per * (world_ - 1) < I, "tensor %d: first mode of %lld rows cannot be split over %d ranks (the last rank would own no rows)",
               p, (long long)I, world_);
    if (world_ == 1) {
      block_upload(t.blk, t.nmodes, dims, data, prec, 0, I);
    } else {
      int64_t ncols = 1;
      for (int i = 1; i < t.nmodes; ++i) ncols *= dims[i];
      std::vector<double> blk((size_t)local_rows * ncols);
      for (int64_t c = 0; c < ncols; ++c)
        std::memcpy(&blk[(size_t)c * local_rows], data + c * I + row0, (size_t)local_rows * sizeof(double));
      block_upload(t.blk, t.nmodes, dims, blk.data(), prec, row0, local_rows, data);
    }
  } else {
    block_upload(t.blk, t.nmodes, dims, data, prec, row0, local_rows);
  }
  t.normsq_valid = false;
}

double Engine::tensor_normsq(int p) {
  require_usable();
  AO_REQUIRE(p >= 0 && p < n_tensors_, "tensor %d out of range", p);
  // test hook (tests/test_gpu_sharded.py): AOADMM_FAULT_INJECT=normsq:<rank> makes that rank fail ALONE in front of
  // a collective, the situation MultiCtx::run's abort path exists for (a device error or OOM on one GPU)
  if (const char* fi = getenv("AOADMM_FAULT_INJECT"))
    if (std::strncmp(fi, "normsq:", 7) == 0 && std::atoi(fi + 7) == rank_ && world_ > 1)
      throw Error(AOADMM_ERR_HIP, "injected fault (AOADMM_FAULT_INJECT)");
  TensorInfo& t = tensors_[p];
  AO_REQUIRE(t.blk.has_data, "tensor %d has no data", p);
  if (!t.normsq_valid) {
    AO_HIP(hipSetDevice(device_));
    DevBuf ws;
    ws.alloc(1024 * sizeof(double) + 64);
    double* slot = slots_.d() + n_modes_ * (kSlotsPerMode + kResidPerMode) + 2 * n_tensors_;
    if (t.par2 ? t.p2.has_mask : t.blk.has_mask) {
      // ||miss .* X||^2 (cmtf_AOADMM.m:133-148): the observed-entry sum of squares of a statistics-only EM pass
      // (needs factors on the device: solve() calls this after the state checks)
      em_pass_enqueue(p, 0);
      double h4[4];
      AO_HIP(hipMemcpyAsync(h4, em_slot(p), sizeof h4, hipMemcpyDeviceToHost, stream_));
      AO_HIP(hipStreamSynchronize(stream_));
      t.normsq = h4[3];
      t.normsq_valid = true;
      return t.normsq;
    }
    if (t.par2) {   // sum_k ||X_k||_F^2  (cmtf_AOADMM.m:145-155)
      tensor_sumsq(slot, t.p2.X.p, AOADMM_PREC_F64, (int64_t)t.p2.I * t.p2.Jtot, ws.d(), stream_);
    } else {
      AO_REQUIRE(!t.blk.x_released, "internal: ||X||^2 of tensor %d asked for after its natural-layout array was released", p);
      tensor_sumsq(slot, t.blk.X.data.p, t.blk.X.prec, t.blk.X.elems_padded(), ws.d(), stream_);
      allreduce(slot, 1);
    }
    double v = 0;
    AO_HIP(hipMemcpyAsync(&v, slot, sizeof(double), hipMemcpyDeviceToHost, stream_));
    AO_HIP(hipStreamSynchronize(stream_));
    t.normsq = v;
    t.normsq_valid = true;
  }
  return t.normsq;
}

void Engine::tensor_synth(int p, int rank, uint64_t seed, double noise, int prec) {
  require_usable();
  AO_REQUIRE(model_done_, "call aoadmm_model_end first");
  AO_REQUIRE(p >= 0 && p < n_tensors_, "tensor %d out of range", p);
  TensorInfo& t = tensors_[p];
  AO_REQUIRE(!t.par2 && t.nmodes == 3, "synthetic generator handles 3-way CP blocks");
  AO_REQUIRE(rank > 0 && rank <= kMaxRank, "bad rank");
  AO_HIP(hipSetDevice(device_));
  const int64_t I = modes_[t.modes[0]].rows, J = modes_[t.modes[1]].rows, K = modes_[t.modes[2]].rows;
  const int64_t per = cdiv(I, world_);
  const int64_t row0 = std::min<int64_t>(I, per * rank_);
  const int64_t loc = std::min<int64_t>(I, row0 + per) - row0;
  AO_REQUIRE(per * (world_ - 1) < I, "tensor %d: first mode of %lld rows cannot be split over %d ranks (the last rank would own no rows)",
             p, (long long)I, world_);           // the same verdict on every rank
  CpBlock& b = t.blk;
  b.nd = 3; b.full0 = I; b.row0 = row0;
  b.dims[0] = loc; b.dims[1] = J; b.dims[2] = K;
  b.X.prec = prec; b.X.nd = 3;
  b.X.dims[0] = loc; b.X.dims[1] = J; b.X.dims[2] = K;
  b.X.pad0 = pad_of(prec, loc);
  b.X.data.alloc((size_t)b.X.elems_padded() * b.X.elem_size());
  DevBuf A, B, C, ws;
  A.alloc((size_t)I * rank * 8); B.alloc((size_t)J * rank * 8); C.alloc((size_t)K * rank * 8);
  ws.alloc(synth_ws_bytes());
  SynthArgs a;
  a.I_loc = loc; a.I_pad = b.X.pad0; a.J = J; a.K = K; a.row0 = row0; a.I_full = I; a.R = rank; a.seed = seed;
  synth_factors(A.d(), B.d(), C.d(), a, stream_);
  double* slot = slots_.d() + n_modes_ * (kSlotsPerMode + kResidPerMode) + 2 * n_tensors_;
  synth_norms(slot, A.d(), B.d(), C.d(), a, ws.d(), stream_);
  allreduce(slot, 3);
  double h[3];
  AO_HIP(hipMemcpyAsync(h, slot, 3 * sizeof(double), hipMemcpyDeviceToHost, stream_));
  AO_HIP(hipStreamSynchronize(stream_));
  // sigma = noise*||X0||/||N|| (create_coupled_data.m:158-162), then X <- X/||X|| (example_script1:91-92)
  const double sigma = h[1] > 0 ? noise * std::sqrt(h[0]) / std::sqrt(h[1]) : 0.0;
  const double nsq = h[0] + 2.0 * sigma * h[2] + sigma * sigma * h[1];
  synth_write(b.X.data.p, prec, A.d(), B.d(), C.d(), a, sigma, 1.0 / std::sqrt(nsq), stream_);
  AO_HIP(hipStreamSynchronize(stream_));
  b.has_data = true;
  b.x_released = false;
  b.cached_mode = -1;
  b.has_xp = false; b.xp_refused = false; b.has_xq = false; b.xq_refused = false; b.has_xc = false; b.xc_refused = false;
  b.xp_ksharded = false;
  {
    int64_t k0 = 0, kloc = 0;
    if (want_ksharded_xp(b, K, &k0, &kloc)) {        // this rank's third-mode slab of the SAME tensor, all rows
      SynthArgs ak = a;
      ak.I_loc = I; ak.I_pad = pad_of(prec, I); ak.row0 = 0; ak.k0 = k0; ak.K_loc = kloc;
      DevBuf slab;
      slab.alloc((size_t)ak.I_pad * J * kloc * b.X.elem_size());
      synth_write(slab.p, prec, A.d(), B.d(), C.d(), ak, sigma, 1.0 / std::sqrt(nsq), stream_);
      adopt_ksharded_xp(b, slab.p, k0, kloc);
      AO_HIP(hipStreamSynchronize(stream_));          // slab is a local
    }
  }
  (void)ensure_permuted_copy2(b);                    // set-up cost of the data, like the generation itself
  (void)ensure_blocked_copy(b);
  AO_HIP(hipStreamSynchronize(stream_));
  t.normsq_valid = false;
}

// ---------------------------------------------------------------------------
// missing data (Z.miss, cmtf_AOADMM.m:68-121)
// ---------------------------------------------------------------------------
void Engine::tensor_mask_upload(int p, const uint8_t* mask) {
  require_usable();
  AO_REQUIRE(p >= 0 && p < n_tensors_ && !tensors_[p].par2, "tensor %d is not a CP block", p);
  AO_REQUIRE(mask != nullptr, "null mask");
  TensorInfo& t = tensors_[p];
  CpBlock& b = t.blk;
  AO_REQUIRE(b.has_data, "upload Z.object{%d} before Z.miss{%d}", p + 1, p + 1);
  AO_REQUIRE(!b.x_released, "Z.object{%d} was released after its pass copies were built: upload it again before Z.miss{%d}", p + 1, p + 1);
  AO_HIP(hipSetDevice(device_));
  const int64_t Iloc = b.dims[0], Ip = b.X.pad0, Ifull = b.full0;
  int64_t ncols = 1;
  for (int i = 1; i < b.nd; ++i) ncols *= b.dims[i];
  // the caller's bytes land in a staging buffer in the padded layout and are packed to one bit per entry on the device
  // (the EM pass reads the mask once per outer iteration: 1/8 of the bytes, and 7/8 of a byte per entry of HBM back)
  DevBuf bytes;
  bytes.alloc((size_t)Ip * ncols);
  AO_HIP(hipMemsetAsync(bytes.p, 1, (size_t)Ip * ncols, stream_));
  // rows [row0, row0 + Iloc) of every column of the full column-major mask; the padding rows stay 1
  AO_HIP(hipMemcpy2DAsync(bytes.p, (size_t)Ip, mask + b.row0, (size_t)Ifull, (size_t)Iloc, (size_t)ncols,
                          hipMemcpyHostToDevice, stream_));
  b.mask.alloc(em_mask_bits_bytes(Ip * ncols));
  em_mask_pack(bytes.as<uint8_t>(), b.mask.as<uint8_t>(), Ip * ncols, stream_);
  if (b.nd == 2) {                                   // matrices keep a transposed copy of the data: mask too
    const int64_t J = b.dims[1], Jp = b.Xt.pad0;
    std::vector<uint8_t> mt((size_t)Jp * Iloc, 1);
    for (int64_t i = 0; i < Iloc; ++i)
      for (int64_t j = 0; j < J; ++j) mt[(size_t)j + (size_t)Jp * i] = mask[b.row0 + i + Ifull * j];
    DevBuf bytesT;
    bytesT.alloc(mt.size());
    AO_HIP(hipMemcpyAsync(bytesT.p, mt.data(), mt.size(), hipMemcpyHostToDevice, stream_));
    b.maskT.alloc(em_mask_bits_bytes((int64_t)mt.size()));
    em_mask_pack(bytesT.as<uint8_t>(), b.maskT.as<uint8_t>(), (int64_t)mt.size(), stream_);
    AO_HIP(hipStreamSynchronize(stream_));             // bytesT and mt are locals
  }
  AO_HIP(hipStreamSynchronize(stream_));
  b.has_mask = true;
  drop_permuted_copies(b);                           // the imputation would have to update them too
  t.normsq_valid = false;
}

bool Engine::has_missing() const {
  for (int p = 0; p < n_tensors_; ++p)
    if (tensors_[p].par2 ? tensors_[p].p2.has_mask : tensors_[p].blk.has_mask) return true;
  return false;
}

double* Engine::em_slot(int p) const {
  return slots_.d() + n_modes_ * (kSlotsPerMode + kResidPerMode) + 2 * n_tensors_ + 16 + 4 * p;
}

static int next_update_distance(int pos, int c, const int* seq, int n);

// One EM pass over tensor p with the current factors: {num, den, obs_res, obs_x2} -> em_slot(p), all-reduced
// over the row shards; update = 1 also overwrites the missing entries with the model (:416-435).
void Engine::em_pass_enqueue(int p, int update, bool fuse_next_pass) {
  TensorInfo& t = tensors_[p];
  if (t.par2) {
    Par2Block& b = t.p2;
    EmPar2Args a;
    a.X = b.X.d(); a.mask = b.mask.as<uint8_t>();
    a.A = modes_[t.modes[0]].fac.d(); a.B = modes_[t.modes[1]].fac.d(); a.C = modes_[t.modes[2]].fac.d();
    a.off = b.off_d.as<int64_t>(); a.K = b.K; a.I = b.I; a.R = b.R; a.update = update;
    emws_.ensure((size_t)b.K * 4 * sizeof(double));
    em_par2_pass(a, emws_.d(), em_slot(p), stream_);
    return;
  }
  CpBlock& b = t.blk;
  const ModeInfo& m0 = modes_[t.modes[0]];
  const ModeInfo& m1 = modes_[t.modes[1]];
  EmCpArgs a;
  a.X = b.X.data.p; a.mask = b.mask.as<uint8_t>();
  a.A = m0.fac.d() + (sharded() ? b.row0 : 0); a.ldA = m0.rows;
  a.B = m1.fac.d(); a.ldB = m1.rows;
  a.C = nullptr; a.ldC = 0;
  a.I = b.dims[0]; a.Ipad = b.X.pad0; a.J = b.dims[1]; a.K = 1; a.R = m0.R; a.update = update;
  if (b.nd == 3) {
    const ModeInfo& m2 = modes_[t.modes[2]];
    a.C = m2.fac.d(); a.ldC = m2.rows; a.K = b.dims[2];
  } else if (b.nd > 3) {
    // order > 3: modes 3..N are merged into one, its factor is their Khatri-Rao product in storage order
    int64_t Km = 1;
    for (int i = 2; i < b.nd; ++i) Km *= b.dims[i];
    b.emkr.ensure((size_t)Km * a.R * 8); b.emkr2.ensure((size_t)Km * a.R * 8);
    const ModeInfo& m2 = modes_[t.modes[2]];
    const double* cur = m2.fac.d();
    int64_t curK = b.dims[2], ld = m2.rows;
    double* bufs[2] = {b.emkr.d(), b.emkr2.d()};
    for (int i = 3; i < b.nd; ++i) {
      const ModeInfo& mn = modes_[t.modes[i]];
      double* dst = bufs[(i - 3) & 1];
      kr_merge(dst, cur, ld, curK, mn.fac.d(), mn.rows, b.dims[i], a.R, stream_);
      cur = dst; curK *= b.dims[i]; ld = curK;
    }
    a.C = cur; a.ldC = curK; a.K = curK;
  }
  emws_.ensure(em_cp_ws_bytes(a.Ipad, a.J, a.K));
  // The imputation pass reads and rewrites the whole block: it can leave the partial contraction the next outer
  // iteration starts with (the pass that serves the first mode it updates), taken from the values it writes back.
  // Contracted mode: 2 or 3 (the strip kernel vectorises mode 1), the one whose factor stays unchanged longest.
  int fused_c = -1;
  ContractPlan fpl;
  FactorRef facs[8];
  static const bool no_fuse = getenv("AOADMM_NO_EM_FUSE") != nullptr;       // development switch (tools/time_em.py)
  if (fuse_next_pass && update && b.nd == 3 && !no_fuse && em_cp_can_fuse(a, b.X.prec) && b.dims[1] <= 65535 &&
      b.dims[2] <= 65535 && !small_direct(b, a.R)) {
    for (int i = 0; i < t.nmodes; ++i) facs[i] = factor_ref(modes_[t.modes[i]]);
    const std::vector<int> seq = update_sequence(p);
    const int pos0 = seq.empty() ? 0 : seq[0];
    int best = -1;
    for (int cand = 2; cand >= 1; --cand) {
      if (cand == pos0) continue;
      const int dist = next_update_distance(pos0, cand, seq.data(), (int)seq.size());
      if (dist > best) { best = dist; fused_c = cand; }
    }
    // test hook (read per call): contract the second mode whenever that is allowed, so that the strip's walk along
    // mode 2 with the fused contraction is exercised by models whose update order would never pick it
    if (getenv("AOADMM_EM_FUSE_SECOND_MODE") != nullptr && pos0 != 1) fused_c = 1;
    const int64_t J = b.dims[1], K = b.dims[2];
    a.walk = fused_c == 1 ? 1 : 2;
    fpl = fused_c == 2 ? make_plan(1, 0, a.Ipad * J, a.Ipad * J, K, a.R, b.X.prec)
                       : make_plan(K, a.Ipad * J, a.Ipad, a.Ipad, J, a.R, b.X.prec);
    fpl.nchunk = em_cp_fused_chunks(a, b.X.prec);
    b.T.ensure(fpl.t_bytes());
    a.T = b.T.p;
    a.t_chunk_stride = fpl.trows() * a.R;
  }
  em_cp_pass(a, b.X.prec, emws_.d(), em_slot(p), stream_);
  if (b.nd == 2 && update) {
    // same imputation on the transposed copy (roles of the two factors swapped); its statistics are discarded
    EmCpArgs at = a;
    at.X = b.Xt.data.p; at.mask = b.maskT.as<uint8_t>();
    at.A = m1.fac.d(); at.ldA = m1.rows; at.B = m0.fac.d() + (sharded() ? b.row0 : 0); at.ldB = m0.rows;
    at.I = b.dims[1]; at.Ipad = b.Xt.pad0; at.J = b.dims[0];
    const size_t wsb = em_cp_ws_bytes(at.Ipad, at.J, 1);
    emws_.ensure(wsb + 64);
    em_cp_pass(at, b.Xt.prec, emws_.d(), emws_.d() + wsb / sizeof(double), stream_);   // statistics to a scratch tail
  }
  allreduce(em_slot(p), 4);
  if (update) b.cached_mode = -1;                    // the data changed: cached partial contractions are stale
  if (fused_c >= 0) { b.cached_mode = fused_c; b.cached_version = facs[fused_c].version; b.plan = fpl; }
}

// ---------------------------------------------------------------------------
// state
// ---------------------------------------------------------------------------
// Resolve a field of the struct G to its device location.  Slab-valued fields (PARAFAC2 B mode and the
// block's P / mu_DeltaB) live back to back; `slab` selects J_k x R block k.
struct StateLoc {
  double* p;
  int64_t rows, cols;
};
static StateLoc slab_loc(DevBuf& buf, const ModeInfo& mB, int slab) {
  buf.ensure((size_t)mB.rows * mB.R * sizeof(double));
  if (slab == AOADMM_ALL_SLABS) return StateLoc{buf.d(), mB.rows, (int64_t)mB.R};   // all K slabs back to back
  AO_REQUIRE(slab >= 0 && slab < mB.K, "slab %d out of range [0,%d)", slab, mB.K);
  return StateLoc{buf.d() + mB.off_k[slab] * mB.R, mB.rows_k[slab], (int64_t)mB.R};
}

void Engine::state_set(int field, int index, int slab, const double* host, int64_t rows, int64_t cols) {
  AO_REQUIRE(model_done_, "call aoadmm_model_end first");
  AO_REQUIRE(host != nullptr && rows > 0 && cols > 0, "state_set: empty array");
  AO_HIP(hipSetDevice(device_));
  StateLoc loc{nullptr, 0, 0};
  if (field == AOADMM_F_COUPLING_FAC) {
    AO_REQUIRE(index >= 0 && index < n_couplings_, "coupling %d out of range", index);
    CouplingInfo& ci = couplings_[index];
    ci.Delta.ensure((size_t)ci.rows * ci.cols * sizeof(double));
    loc = StateLoc{ci.Delta.d(), ci.rows, ci.cols};
    ci.has_state = true;
  } else if (field == AOADMM_F_DELTAB || field == AOADMM_F_P || field == AOADMM_F_MU_DELTAB) {
    AO_REQUIRE(index >= 0 && index < n_tensors_ && tensors_[index].par2, "tensor %d is not a PARAFAC2 block", index);
    TensorInfo& t = tensors_[index];
    Par2Block& b = t.p2;
    const ModeInfo& mB = modes_[t.modes[1]];
    if (field == AOADMM_F_DELTAB) {
      b.DeltaB.ensure((size_t)b.R * b.R * sizeof(double));
      loc = StateLoc{b.DeltaB.d(), (int64_t)b.R, (int64_t)b.R};
      b.has_DeltaB = true;
    } else if (field == AOADMM_F_P) {
      loc = slab_loc(b.P, mB, slab);
      if (slab == AOADMM_ALL_SLABS) b.have_P.assign(b.K, 1); else b.have_P[slab] = 1;
    } else {
      loc = slab_loc(b.muDB, mB, slab);
      if (slab == AOADMM_ALL_SLABS) b.have_mu.assign(b.K, 1); else b.have_mu[slab] = 1;
    }
  } else {
    check_mode(index);
    ModeInfo& mi = modes_[index];
    auto whole = [&](DevBuf& buf, int64_t r, int64_t c) {
      buf.ensure((size_t)r * c * sizeof(double));
      return StateLoc{buf.d(), r, c};
    };
    switch (field) {
      case AOADMM_F_FAC:
        loc = mi.slabs ? slab_loc(mi.fac, mi, slab) : whole(mi.fac, mi.rows, mi.R);
        mi.has_fac = true; mi.version++;
        break;
      case AOADMM_F_CONSTRAINT_FAC:
        loc = mi.slabs ? slab_loc(mi.Z, mi, slab) : whole(mi.Z, mi.rows, mi.R);
        mi.has_Z = true;
        break;
      case AOADMM_F_CONSTRAINT_DUAL:
        loc = mi.slabs ? slab_loc(mi.mu, mi, slab) : whole(mi.mu, mi.rows, mi.R);
        mi.has_mu = true;
        break;
      case AOADMM_F_COUPLING_DUAL:
        AO_REQUIRE(!mi.slabs, "the PARAFAC2 B_k mode cannot be coupled");
        loc = whole(mi.muD, rows, cols);
        mi.has_muD = true; mi.muD_rows = rows; mi.muD_cols = cols;
        break;
      default: throw Error(AOADMM_ERR_INVALID, fmt("unknown state field %d", field));
    }
  }
  AO_REQUIRE(rows == loc.rows && cols == loc.cols, "state field %d index %d slab %d must be %lld x %lld, got %lld x %lld", field,
             index + 1, slab + 1, (long long)loc.rows, (long long)loc.cols, (long long)rows, (long long)cols);
  AO_HIP(hipMemcpyAsync(loc.p, host, (size_t)rows * cols * sizeof(double), hipMemcpyHostToDevice, stream_));
  AO_HIP(hipStreamSynchronize(stream_));
}

void Engine::state_get(int field, int index, int slab, double* host, int64_t rows, int64_t cols) {
  AO_REQUIRE(host != nullptr, "state_get: null destination");
  AO_HIP(hipSetDevice(device_));
  StateLoc loc{nullptr, 0, 0};
  if (field == AOADMM_F_COUPLING_FAC) {
    AO_REQUIRE(index >= 0 && index < n_couplings_, "coupling %d out of range", index);
    AO_REQUIRE(couplings_[index].has_state, "coupling_fac{%d} was never set", index + 1);
    loc = StateLoc{couplings_[index].Delta.d(), couplings_[index].rows, couplings_[index].cols};
  } else if (field == AOADMM_F_DELTAB || field == AOADMM_F_P || field == AOADMM_F_MU_DELTAB) {
    AO_REQUIRE(index >= 0 && index < n_tensors_ && tensors_[index].par2, "tensor %d is not a PARAFAC2 block", index);
    TensorInfo& t = tensors_[index];
    Par2Block& b = t.p2;
    const ModeInfo& mB = modes_[t.modes[1]];
    AO_REQUIRE(b.has_DeltaB, "DeltaB{%d} was never set", index + 1);
    if (field == AOADMM_F_DELTAB) loc = StateLoc{b.DeltaB.d(), (int64_t)b.R, (int64_t)b.R};
    else if (field == AOADMM_F_P) loc = slab_loc(b.P, mB, slab);
    else loc = slab_loc(b.muDB, mB, slab);
  } else {
    check_mode(index);
    ModeInfo& mi = modes_[index];
    switch (field) {
      case AOADMM_F_FAC:
        AO_REQUIRE(mi.has_fac, "fac{%d} was never set", index + 1);
        loc = mi.slabs ? slab_loc(mi.fac, mi, slab) : StateLoc{mi.fac.d(), mi.rows, (int64_t)mi.R};
        break;
      case AOADMM_F_CONSTRAINT_FAC:
        AO_REQUIRE(mi.has_Z, "constraint_fac{%d} was never set", index + 1);
        loc = mi.slabs ? slab_loc(mi.Z, mi, slab) : StateLoc{mi.Z.d(), mi.rows, (int64_t)mi.R};
        break;
      case AOADMM_F_CONSTRAINT_DUAL:
        AO_REQUIRE(mi.has_mu, "constraint_dual_fac{%d} was never set", index + 1);
        loc = mi.slabs ? slab_loc(mi.mu, mi, slab) : StateLoc{mi.mu.d(), mi.rows, (int64_t)mi.R};
        break;
      case AOADMM_F_COUPLING_DUAL:
        AO_REQUIRE(mi.has_muD, "coupling_dual_fac{%d} was never set", index + 1);
        loc = StateLoc{mi.muD.d(), mi.muD_rows, mi.muD_cols};
        break;
      default: throw Error(AOADMM_ERR_UNSUPPORTED, fmt("state field %d not available", field));
    }
  }
  AO_REQUIRE(rows == loc.rows && cols == loc.cols, "state_get: destination is %lld x %lld, field is %lld x %lld", (long long)rows,
             (long long)cols, (long long)loc.rows, (long long)loc.cols);
  AO_HIP(hipMemcpyAsync(host, loc.p, (size_t)rows * cols * sizeof(double), hipMemcpyDeviceToHost, stream_));
  AO_HIP(hipStreamSynchronize(stream_));
}

// ---------------------------------------------------------------------------
// MTTKRP engine
// ---------------------------------------------------------------------------
hipEvent_t Engine::take_event() {
  if (event_pool_.empty()) {
    for (int i = 0; i < 64; ++i) {
      hipEvent_t e = nullptr;
      AO_HIP(hipEventCreate(&e));
      event_pool_.push_back(e);
    }
  }
  hipEvent_t e = event_pool_.back();
  event_pool_.pop_back();
  return e;
}

void Engine::timed_contract(const void* X, int prec, const ContractPlan& pl, const double* F, int64_t ldF,
                            void* frag, void* T) {
  KernelStats& ks = kstats_[pl.lead ? 1 : 0];
  hipEvent_t e0 = nullptr, e1 = nullptr;
  static const bool no_events = getenv("AOADMM_NO_PASS_EVENTS") != nullptr;   // development switch (tools/gap_analysis.py)
  // Every 4th pass is bracketed by events (the three kinds of pass alternate with period 3, so the sample cycles through
  // them): the records cost ~4 us of launch gap on each side of a pass -- nothing at 2000^3, 1 % of an iteration at one
  // rank's share of 8 GPUs.  kernel_stats() returns the mean of the timed launches times the launch count.
  // AOADMM_PASS_EVENT_EVERY=1 times every pass (the profile tools).
  static const int every = [] { const char* e = getenv("AOADMM_PASS_EVENT_EVERY"); const int v = e ? atoi(e) : 4; return v < 1 ? 1 : v; }();
  if (profile_ && !no_events && ks.launches % every == 0) {
    if (ks.pending.size() >= 512) fold_finished(ks);     // a long solve never holds more than a few hundred events
    if (ks.pending.size() < 4096) {
      e0 = take_event();
      e1 = take_event();
    }
  }
  launch_contract(X, prec, pl, F, ldF, frag, T, stream_, e0, e1);
  if (e0) { ks.pending.emplace_back(e0, e1); ks.timed++; }
  ks.launches++;
  ks.bytes += pl.algorithmic_bytes(prec);
  ks.flops += pl.flops();
}

// pairs whose second event has completed are added to ks.ms and their events go back to the pool (no synchronisation)
void Engine::fold_finished(KernelStats& ks) {
  size_t keep = 0;
  for (size_t i = 0; i < ks.pending.size(); ++i) {
    auto& pr = ks.pending[i];
    float t = 0.f;
    if (hipEventQuery(pr.second) == hipSuccess && hipEventElapsedTime(&t, pr.first, pr.second) == hipSuccess) {
      ks.ms += t;
      event_pool_.push_back(pr.first);
      event_pool_.push_back(pr.second);
    } else {
      ks.pending[keep++] = pr;
    }
  }
  ks.pending.resize(keep);
}

void Engine::kernel_stats(int which, int reset, double* ms, int64_t* launches, double* bytes, double* flops) {
  AO_REQUIRE(which >= 0 && which <= 2, "kernel_stats: which must be 0, 1 or 2");
  if (which == 2) profile_reductions_ = true;
  AO_HIP(hipSetDevice(device_));
  AO_HIP(hipStreamSynchronize(stream_));
  KernelStats& ks = kstats_[which];
  for (auto& pr : ks.pending) {
    float t = 0.f;
    AO_HIP(hipEventElapsedTime(&t, pr.first, pr.second));
    ks.ms += t;
    event_pool_.push_back(pr.first);
    event_pool_.push_back(pr.second);
  }
  ks.pending.clear();
  // launches that went untimed (event budget exhausted) count at the mean of the timed ones, so ms / launches stays
  // the mean launch duration
  if (ms) *ms = (ks.timed > 0 && ks.timed < ks.launches) ? ks.ms * (double)ks.launches / (double)ks.timed : ks.ms;
  if (launches) *launches = ks.launches;
  if (bytes) *bytes = ks.bytes;
  if (flops) *flops = ks.flops;
  if (reset) { ks.ms = 0; ks.launches = 0; ks.timed = 0; ks.bytes = 0; ks.flops = 0; }
}

// distance (in updates) until tensor position `c` is updated again after position `pos`
static int next_update_distance(int pos, int c, const int* seq, int n) {
  if (!seq || n <= 0) return c;            // no information: prefer the last mode
  int at = -1;
  for (int i = 0; i < n; ++i)
    if (seq[i] == pos) at = i;
  if (at < 0) return c;
  for (int d = 1; d <= n; ++d)
    if (seq[(at + d) % n] == c) return d;
  return n + 1;                              // never updated
}

// Tensor pass for a 3-way block: makes b.T hold the partial contraction that serves an MTTKRP for tensor
// position `pos` (a cached one is reused while its factor is unchanged).
// Second resident copy with the first mode last (see CpBlock::Xp).  Built lazily; refused when the mask of an EM
// problem would have to be kept in sync, when the caller opted out, or when HBM cannot hold it.

static bool room_for(size_t bytes) {
  size_t free_b = 0, total_b = 0;
  return hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b >= bytes + (size_t)(4ull << 30);
}

// Mode-3 sharding of the mode-1 pass's copy: every rank must reach the same verdict (the collectives that follow the
// pass differ: own rows of mode 3 instead of partial sums).
bool Engine::want_ksharded_xp(const CpBlock& b, int64_t K, int64_t* k0, int64_t* kloc) const {
  static const bool off = getenv("AOADMM_NO_KSHARD") != nullptr;            // development switch
  if (!sharded() || world_ <= 1 || !allow_xp_ || b.has_mask || b.nd != 3 || off) return false;
  const int64_t per = cdiv(K, world_);
  if (per * (world_ - 1) >= K) return false;          // some rank would own no slab
  *k0 = per * rank_;
  *kloc = std::min<int64_t>(K, *k0 + per) - *k0;
  return true;
}
void Engine::adopt_ksharded_xp(CpBlock& b, const void* slab, int64_t k0, int64_t kloc) {
  const int64_t Ifull = b.full0, J = b.dims[1];
  const int prec = b.X.prec;
  const int64_t Ipf = round_up(Ifull, prec == AOADMM_PREC_F32 ? 4 : 2);
  const int64_t Jp = round_up(J, prec == AOADMM_PREC_F32 ? 4 : 2);
  b.Xp.alloc(blocked_bytes(Jp * kloc, Ifull, b.X.elem_size()));
  b.Jp = Jp;
  AO_REQUIRE(block_layout_copy(slab, b.Xp.p, 1, prec, Ifull, Ipf, J, kloc, Jp, stream_), "tensor mode too long for the copy kernels");
  b.has_xp = true; b.xp_ksharded = true; b.xp_k0 = k0; b.xp_kloc = kloc;
}

bool Engine::ensure_permuted_copy(CpBlock& b) {
  if (b.has_xp) return true;
  if (b.xp_refused || !allow_xp_ || b.has_mask || b.nd != 3) return false;
  const int64_t I = b.dims[0], J = b.dims[1], K = b.dims[2];
  const int64_t Jp = round_up(J, b.X.prec == AOADMM_PREC_F32 ? 4 : 2);
  const size_t bytes = blocked_bytes(Jp * K, I, b.X.elem_size());
  if (!room_for(bytes)) { b.xp_refused = true; return false; }
  b.Xp.alloc(bytes);
  b.Jp = Jp;
  if (!block_layout_copy(b.X.data.p, b.Xp.p, 1, b.X.prec, I, b.X.pad0, J, K, Jp, stream_)) {
    b.Xp.release(); b.xp_refused = true; return false;
  }
  b.has_xp = true;
  return true;
}

// Xq: rows (k, i), columns j -- built from X directly
bool Engine::ensure_permuted_copy2(CpBlock& b) {
  if (b.has_xq) return true;
  if (b.xq_refused || !ensure_permuted_copy(b)) return false;
  const int64_t I = b.dims[0], J = b.dims[1], K = b.dims[2];
  const int64_t Kp = round_up(K, b.X.prec == AOADMM_PREC_F32 ? 4 : 2);
  const size_t bytes = blocked_bytes(Kp * I, J, b.X.elem_size());
  if (!room_for(bytes)) { b.xq_refused = true; return false; }
  b.Xq.alloc(bytes);
  b.Kp = Kp;
  if (!block_layout_copy(b.X.data.p, b.Xq.p, 2, b.X.prec, I, b.X.pad0, J, K, Kp, stream_)) {
    b.Xq.release(); b.xq_refused = true; return false;
  }
  b.has_xq = true;
  return true;
}

// Xc: the rows of X itself, row-blocked (the pass that contracts mode 3)
bool Engine::ensure_blocked_copy(CpBlock& b) {
  if (b.has_xc) return true;
  if (b.xc_refused || !allow_xp_ || b.has_mask || b.nd != 3) return false;
  const int64_t I = b.dims[0], J = b.dims[1], K = b.dims[2];
  const size_t bytes = blocked_bytes(b.X.pad0 * J, K, b.X.elem_size());
  if (!room_for(bytes)) { b.xc_refused = true; return false; }
  b.Xc.alloc(bytes);
  if (!block_layout_copy(b.X.data.p, b.Xc.p, 0, b.X.prec, I, b.X.pad0, J, K, 0, stream_)) {
    b.Xc.release(); b.xc_refused = true; return false;
  }
  b.has_xc = true;
  return true;
}
// Releases Z.object{p} in its natural layout once every tensor pass has its own resident copy.  A 2000^3 double array is
// 64 GB: with the natural array and three copies a MATLAB caller sat at 256 of 288 GB before any workspace.  Policy:
// AOADMM_RELEASE_NATURAL=1 always, =0 never, default: when less HBM is free than the array itself occupies.  Afterwards
// Z.miss cannot be attached without uploading the data again, and aoadmm_resident_unfold_gram answers
// AOADMM_ERR_UNSUPPORTED (the caller falls back to the host-array form).
void Engine::maybe_release_natural(TensorInfo& t) {
  CpBlock& b = t.blk;
  if (b.x_released || b.nd != 3 || !(b.has_xc && b.has_xp && b.has_xq) || b.has_mask || !t.normsq_valid || !b.X.data.p) return;
  const char* pe = getenv("AOADMM_RELEASE_NATURAL");   // read per solve (the test suite switches it inside one process)
  const int policy = pe ? (atoi(pe) != 0 ? 1 : -1) : 0;
  if (policy < 0) return;
  if (policy == 0) {
    size_t free_b = 0, total_b = 0;
    const size_t mine = (size_t)b.X.elems_padded() * b.X.elem_size();
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b >= mine) return;
  }
  AO_HIP(hipStreamSynchronize(stream_));             // the copies were built from it on this stream
  b.X.data.release();
  b.x_released = true;
}

void Engine::drop_permuted_copies(CpBlock& b) {
  if (b.has_xp) { b.Xp.release(); b.has_xp = false; b.xp_ksharded = false; b.cached_mode = -1; }
  if (b.has_xq) { b.Xq.release(); b.has_xq = false; b.cached_mode = -1; }
  if (b.has_xc) { b.Xc.release(); b.has_xc = false; b.cached_mode = -1; }
}

void Engine::ensure_contraction(CpBlock& b, int pos, const FactorRef* facs, int R, bool use_cache,
                                const int* update_seq, int nseq) {
  const int prec = b.X.prec;
  const int64_t I = b.dims[0], Ip = b.X.pad0, J = b.dims[1], K = b.dims[2];
  if (b.x_released) use_cache = true;                  // only the pass copies are resident (maybe_release_natural)
  const bool hit = use_cache && b.cached_mode >= 0 && b.cached_mode != pos &&
                   facs[b.cached_mode].version == b.cached_version;
  if (hit) return;
  // which mode to contract: any mode but `pos`; prefer the one whose factor stays unchanged longest so
  // that the partial contraction also serves the next update (cycle 3->{1,2}, 2->{3,1}, 1->{2,3}:
  // 1.5 tensor reads per outer iteration).  The leading mode needs the LDS-transposed kernel (fp32 only).
  int c = -1, best = -1;
  for (int cand = 2; cand >= 0; --cand) {
    if (cand == pos) continue;
    static const bool force_lead = getenv("AOADMM_FORCE_LEAD") != nullptr;   // development switch (tools/perf_mttkrp.py)
    // contracting mode 1 needs the permuted copy (any precision) or the LDS-transposed kernel (fp32 only)
    if (cand == 0 && !((use_cache || force_lead) && (prec == AOADMM_PREC_F32 || ensure_permuted_copy(b)))) continue;
    if (cand != 0 && force_lead && prec == AOADMM_PREC_F32 && pos != 0) continue;
    const int dist = next_update_distance(pos, cand, update_seq, nseq);
    if (dist > best) { best = dist; c = cand; }
  }
  ContractPlan pl;
  const double* Fc = facs[c].p;
  // a pass on a row-blocked copy: one "batch" per row block, each a contiguous MB x C matrix
  auto blocked_plan = [&](int64_t M, int64_t C) {
    const int64_t MB = kRowBlockElems;
    return make_plan(round_up(M, MB) / MB, MB * C, MB, MB, C, R, prec);
  };
  if (c == 2) {
    if (use_cache && ensure_blocked_copy(b)) {
      pl = blocked_plan(Ip * J, K);
      pl.on_xc = true;
    } else {
      pl = make_plan(1, 0, Ip * J, Ip * J, K, R, prec);
    }
  } else if (c == 1) {
    if (use_cache && ensure_permuted_copy2(b)) {
      // Xq: rows (k, i), columns j -> one streaming pass instead of K batches of an I x J matrix
      pl = blocked_plan(b.Kp * I, J);
      pl.on_xq = true;
    } else {
      pl = make_plan(K, Ip * J, Ip, Ip, J, R, prec);
    }
  } else {
    Fc = facs[0].p + (sharded() ? b.row0 : 0);
    static const bool force_ldskernel = getenv("AOADMM_LEAD_KERNEL") != nullptr;   // development switch
    if (!force_ldskernel && ensure_permuted_copy(b)) {
      // Xp: rows (j, k), columns i -> the register-streaming contraction.  With a communicator the copy holds this
      // rank's slab of mode 3 and ALL of mode 1 (CpBlock::xp_ksharded): a complete T of 1/N the size
      if (b.xp_ksharded) { pl = blocked_plan(b.Jp * b.xp_kloc, b.full0); Fc = facs[0].p; }
      else pl = blocked_plan(b.Jp * K, I);
      pl.on_xp = true;
    } else {
      pl = make_lead_plan(J * K, Ip, I, R);
    }
  }
  b.T.ensure(pl.t_bytes()); b.frag.ensure(pl.frag_bytes(prec));
  timed_contract(pl.on_xp ? b.Xp.p : (pl.on_xq ? b.Xq.p : (pl.on_xc ? b.Xc.p : b.X.data.p)), prec, pl, Fc, facs[c].ld, b.frag.p,
                 b.T.p);
  b.cached_mode = c; b.cached_version = facs[c].version; b.plan = pl;
}

// The first tensor pass of the next outer iteration does not depend on the host's stopping decision, so
// it is enqueued before the host waits for the objective values: the round trip hides behind it.
bool Engine::prefetch_next_contraction(const aoadmm_options& opt) {
  if (!opt.use_dimtree) return false;
  for (int cid = -1; cid < n_couplings_; ++cid) {
    for (int p = 0; p < n_tensors_; ++p)
      for (int m = 0; m < n_modes_; ++m) {
        const ModeInfo& mi = modes_[m];
        if (mi.coupling != cid || mi.tensor != p) continue;
        TensorInfo& t = tensors_[p];                      // first mode the next iteration updates
        if (t.par2 || t.blk.nd != 3 || small_direct(t.blk, mi.R)) return false;
        FactorRef facs[8];
        for (int i = 0; i < t.nmodes; ++i) {
          const ModeInfo& o = modes_[t.modes[i]];
          facs[i] = factor_ref(o);
        }
        std::vector<int> seq = update_sequence(p);
        const int64_t before = kstats_[0].launches + kstats_[1].launches;
        ensure_contraction(t.blk, mi.pos, facs, mi.R, true, seq.data(), (int)seq.size());
        return kstats_[0].launches + kstats_[1].launches > before;     // false: the cached pass still serves
      }
  }
  return false;
}

void Engine::block_mttkrp(CpBlock& b, int pos, const FactorRef* facs, int R, double scale, double* out,
                          int64_t ldOut, bool use_cache, const int* update_seq, int nseq, bool collective,
                          bool tensor_pass, const SysBuild* sys, bool* sys_done) {
  if (sys_done) *sys_done = false;
  AO_REQUIRE(b.has_data, "tensor has no data");
  AO_REQUIRE(pos >= 0 && pos < b.nd, "mttkrp: mode %d out of range", pos);
  const int prec = b.X.prec;
  const int64_t I = b.dims[0], Ip = b.X.pad0;
  const bool sharded = collective && this->sharded();
  double* out_local = out;
  const int64_t out_rows_full = (pos == 0) ? b.full0 : b.dims[pos];
  // "Every rank fills its own rows of a zeroed buffer; the all-reduce is the all-gather": the zeroed buffer is a send
  // buffer of the block that is cleared ONCE -- the rows of other ranks are never written, the own rows are overwritten
  // by every MTTKRP -- and the all-reduce goes from it into `out` (a fill in front of every such MTTKRP was 5 us on the
  // critical path, two per outer iteration).
  const double* send = nullptr;
  int64_t ld_local = ldOut;
  auto own_rows_buffer = [&](int which, int64_t rows_full, int64_t row_first) {
    DevBuf& ob = b.own[which];
    const size_t need = (size_t)rows_full * R * sizeof(double);
    if (b.own_bytes[which] != need || b.own_row0[which] != row_first) {    // (another rank's rows would stay behind)
      if (b.own_bytes[which] != need) ob.alloc(need);
      AO_HIP(hipMemsetAsync(ob.p, 0, need, stream_));
      b.own_bytes[which] = need;
      b.own_row0[which] = row_first;
    }
    send = ob.d();
    ld_local = rows_full;
    out_local = ob.d() + row_first;
  };
  if (sharded && pos == 0) own_rows_buffer(0, out_rows_full, b.row0);
  const double* F0 = facs[0].p + (sharded ? b.row0 : 0);     // local rows of the first factor

  if (!tensor_pass && small_direct(b, R)) {
    // tiny block: the whole MTTKRP in one launch (contract.hip small_mttkrp_k), no partial-contraction cache
    const int64_t J = b.dims[1], K = b.nd == 3 ? b.dims[2] : 1;
    const int64_t st[3] = {1, Ip, Ip * J};
    const int64_t ext[3] = {I, J, K};
    int ia = pos == 0 ? 1 : 0, ib = pos == 2 ? 1 : 2;
    SmallMttkrp sm;
    sm.X = b.X.data.p;
    sm.sn = st[pos]; sm.sa = st[ia]; sm.Na = (int)ext[ia];
    sm.Fa = facs[ia].p; sm.lda = facs[ia].ld;
    if (b.nd == 3) { sm.sb = st[ib]; sm.Nb = (int)ext[ib]; sm.Fb = facs[ib].p; sm.ldb = facs[ib].ld; }
    else { sm.sb = 0; sm.Nb = 1; sm.Fb = nullptr; sm.ldb = 0; }
    sm.R = R; sm.scale = scale; sm.out = out; sm.ldOut = ldOut;
    const bool rode = small_mttkrp(sm, prec, ext[pos], stream_, sys);
    if (sys_done) *sys_done = rode;
    return;
  }
  if (b.nd == 2) {
    const int64_t J = b.dims[1];
    if (pos == 0) {
      ContractPlan pl = make_plan(1, 0, Ip, Ip, J, R, prec);
      b.T.ensure(pl.t_bytes()); b.frag.ensure(pl.frag_bytes(prec));
      timed_contract(b.X.data.p, prec, pl, facs[1].p, facs[1].ld, b.frag.p, b.T.p);
      launch_t_to_colmajor(b.T.p, pl.tprec, pl.nchunk, pl.trows(), I, R, scale, out_local, ld_local, stream_);
    } else {
      const int64_t Jp = b.Xt.pad0;
      ContractPlan pl = make_plan(1, 0, Jp, Jp, I, R, prec);
      b.T.ensure(pl.t_bytes()); b.frag.ensure(pl.frag_bytes(prec));
      timed_contract(b.Xt.data.p, prec, pl, F0, facs[0].ld, b.frag.p, b.T.p);
      launch_t_to_colmajor(b.T.p, pl.tprec, pl.nchunk, pl.trows(), J, R, scale, out_local, ld_local, stream_);
    }
    b.cached_mode = -1;
  } else if (b.nd == 3) {
    const int64_t J = b.dims[1], K = b.dims[2];
    ensure_contraction(b, pos, facs, R, use_cache, update_seq, nseq);
    const int c = b.cached_mode;
    const ContractPlan& pl = b.plan;
    // the mode-1 pass on a copy sharded along mode 3: T(j, k in K_g, r) is complete; mode 3's output is this rank's rows
    const bool ksh = sharded && pl.on_xp && b.xp_ksharded;
    // T rows are (a + Apad*bb) with (a, bb) the two uncontracted modes in the order the pass's copy stores them:
    // tensor order on X and Xp, (k, i) on Xq
    int ia = c == 0 ? 1 : 0, ib = c == 2 ? 1 : 2;
    if (pl.on_xq) { ia = 2; ib = 0; }
    const int64_t ext[3] = {I, J, ksh ? b.xp_kloc : K};
    const int64_t An = ext[ia], Bn = ext[ib];
    const int64_t Apad = pl.on_xq ? b.Kp : (ia == 0 ? Ip : (pl.on_xp ? b.Jp : J));
    // factor of a mode: the first mode's factor is addressed at this rank's rows (the third mode's too under `ksh`)
    auto fac_p = [&](int m) { return m == 0 ? F0 : (m == 2 && ksh ? facs[2].p + b.xp_k0 : facs[m].p); };
    auto fac_pT = [&](int m) -> const double* {
      if (!facs[m].pT) return nullptr;
      if (m == 2 && ksh) return facs[2].pT + b.xp_k0 * R;
      return facs[m].pT + ((m == 0 && sharded) ? b.row0 * R : 0);
    };
    if (ksh && pos == 2) own_rows_buffer(1, K, b.xp_k0);   // own rows of the zeroed send buffer; the all-reduce is the all-gather
    // the reduction over T is timed like the passes (kernel_stats slot 2): it reads all of T once
    KernelStats& rs = kstats_[2];
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (profile_ && profile_reductions_) {
      if (rs.pending.size() >= 512) fold_finished(rs);
      if (rs.pending.size() < 4096) {
        e0 = take_event();
        e1 = take_event();
        AO_HIP(hipEventRecord(e0, stream_));
      }
    }
    if (pos == ia) {
      b.scratch.ensure(reduce_outer_scratch_bytes(An, Bn, R));
      b.ft.ensure(reduce_factor_scratch_bytes(Bn, R));
      const bool rode = launch_reduce_outer(b.T.p, pl.tprec, pl.nchunk, pl.trows(), An, Apad, Bn, R, fac_p(ib), facs[ib].ld, scale,
                                            out_local, ld_local, b.scratch.d(), b.ft.d(), stream_, fac_pT(ib), 0, sys);
      if (sys_done) *sys_done = rode;
    } else {
      AO_REQUIRE(pos == ib, "internal: cached contraction cannot serve this mode");
      b.ft.ensure(reduce_factor_scratch_bytes(An, R));
      const bool rode = launch_reduce_inner(b.T.p, pl.tprec, pl.nchunk, pl.trows(), An, Apad, Bn, R, fac_p(ia), facs[ia].ld, scale,
                                            out_local, ld_local, b.ft.d(), stream_, fac_pT(ia), 0, sys);
      if (sys_done) *sys_done = rode;
    }
    if (e0) {
      AO_HIP(hipEventRecord(e1, stream_));
      rs.pending.emplace_back(e0, e1);
      rs.timed++;
    }
    rs.launches++;
    rs.bytes += (double)pl.t_bytes();
  } else {
    // N-way (N > 3): contract the last mode (the one before it when pos is last) on the matrix cores with all
    // leading modes merged into the unfolding row, then fold the remaining modes one at a time over T: trailing
    // modes with reduce_outer (down to pos), leading modes with reduce_inner (up to pos), each fold leaving a
    // smaller T in the same [row][r] layout (fp64).  No partial-contraction reuse for these: N passes per iteration.
    const int N = b.nd;
    const int c = pos == N - 1 ? N - 2 : N - 1;
    int64_t lead = Ip;                                  // merged size of the modes before c (first one padded)
    for (int m = 1; m < c; ++m) lead *= b.dims[m];
    ContractPlan pl = (c == N - 1) ? make_plan(1, 0, lead, lead, b.dims[c], R, prec)
                                   : make_plan(b.dims[N - 1], lead * b.dims[c], lead, lead, b.dims[c], R, prec);
    b.T.ensure(pl.t_bytes()); b.frag.ensure(pl.frag_bytes(prec));
    timed_contract(b.X.data.p, prec, pl, c == 0 ? F0 : facs[c].p, facs[c].ld, b.frag.p, b.T.p);
    // remaining modes in memory order, with their padded extents inside T
    int rem[8], nrem = 0;
    int64_t ext[8];
    for (int m = 0; m < N; ++m)
      if (m != c) { rem[nrem] = m; ext[nrem] = m == 0 ? Ip : b.dims[m]; ++nrem; }
    const void* Tin = b.T.p;
    int tprec = pl.tprec, nchunk = pl.nchunk;
    int64_t trows = pl.trows();
    int flip = 0;
    auto tbuf = [&](int64_t rows) {
      DevBuf& d = flip ? b.tmpB : b.tmpA;
      flip ^= 1;
      d.ensure((size_t)rows * R * sizeof(double));
      return d.d();
    };
    // fold trailing modes above pos (last remaining mode first)
    while (nrem > 1 && rem[nrem - 1] != pos) {
      const int mb = rem[nrem - 1];
      int64_t Arows = 1;
      for (int q = 0; q < nrem - 1; ++q) Arows *= ext[q];
      const bool last = nrem == 2;                      // after this fold only `pos` remains (it is rem[0])
      double* dst = last ? out_local : tbuf(Arows);
      const int64_t An = last ? (rem[0] == 0 ? I : b.dims[rem[0]]) : Arows;
      b.scratch.ensure(reduce_outer_scratch_bytes(An, b.dims[mb], R));
      b.ft.ensure(reduce_factor_scratch_bytes(b.dims[mb], R));
      launch_reduce_outer(Tin, tprec, nchunk, trows, An, Arows, b.dims[mb], R, facs[mb].p, facs[mb].ld,
                          last ? scale : 1.0, dst, last ? ld_local : 0, b.scratch.d(), b.ft.d(), stream_, nullptr, last ? 0 : 1);
      if (last) { nrem = 1; break; }
      Tin = dst; tprec = AOADMM_PREC_F64; nchunk = 1; trows = Arows;
      --nrem;
    }
    // fold leading modes below pos (first remaining mode first)
    while (nrem > 1) {
      const int ma = rem[0];
      int64_t Brows = 1;
      for (int q = 1; q < nrem; ++q) Brows *= ext[q];
      const bool last = nrem == 2;                      // after this fold only `pos` remains (it is rem[1])
      double* dst = last ? out_local : tbuf(Brows);
      const int64_t An = ma == 0 ? I : b.dims[ma];
      b.ft.ensure(reduce_factor_scratch_bytes(An, R));
      launch_reduce_inner(Tin, tprec, nchunk, trows, An, ext[0], Brows, R, ma == 0 ? F0 : facs[ma].p, facs[ma].ld,
                          last ? scale : 1.0, dst, last ? ld_local : 0, b.ft.d(), stream_, nullptr, last ? 0 : 1);
      Tin = dst; tprec = AOADMM_PREC_F64; nchunk = 1; trows = Brows;
      for (int q = 0; q + 1 < nrem; ++q) { rem[q] = rem[q + 1]; ext[q] = ext[q + 1]; }
      --nrem;
    }
    b.cached_mode = -1;
  }
  if (sharded) {
    if (send) {                                        // from the block's own-rows send buffer (ld = rows) into `out`
      if (ldOut == out_rows_full) allreduce_from(send, out, out_rows_full * R);
      else for (int r = 0; r < R; ++r) allreduce_from(send + out_rows_full * r, out + ldOut * r, out_rows_full);
    } else if (ldOut == out_rows_full) allreduce(out, out_rows_full * R);
    else for (int r = 0; r < R; ++r) allreduce(out + ldOut * r, out_rows_full);
  }
}

std::vector<int> Engine::update_sequence(int p) const {
  // order in which the positions of tensor p are updated inside one outer iteration:
  // uncoupled modes first, then coupling ids ascending (cmtf_fun_AOADMM.m:10,89-93)
  std::vector<int> seq;
  const TensorInfo& t = tensors_[p];
  for (int cid = -1; cid < n_couplings_; ++cid)
    for (int i = 0; i < t.nmodes; ++i)
      if (modes_[t.modes[i]].coupling == cid) seq.push_back(i);
  return seq;
}

// ---------------------------------------------------------------------------
// per-mode pieces of the outer loop
// ---------------------------------------------------------------------------
void Engine::ensure_mode_work(ModeInfo& mi) {
  const size_t nR = (size_t)mi.rows * mi.R * sizeof(double), RR = (size_t)mi.R * mi.R * sizeof(double);
  mi.A.ensure(nR); mi.Ab.ensure(nR);
  mi.gram.ensure(RR); mi.C.ensure(RR); mi.Bsys.ensure(RR); mi.L.ensure(RR); mi.Binv.ensure(RR);
  mi.rho.ensure(64);
  mi.Zold.ensure(nR); mi.V.ensure(nR); mi.Znew.ensure(nR); mi.RHS.ensure(nR); mi.TD.ensure(nR); mi.tmp.ensure(nR);
  mi.part.ensure((size_t)admm_partials(mi.rows) * 4 * sizeof(double));
  if (mi.constrained) mi.proxws.ensure(prox_ws_bytes(mi.prox.type, mi.rows, mi.R));
  atbws_.ensure(atb_ws_bytes(mi.rows, mi.R, mi.R));
}

// Gram of the current factor (:66, :148); the same kernel leaves a row-major copy of the factor for the T
// reductions and closes the ADMM loop that produced the factor.  Call BEFORE bumping mi.version.
void Engine::compute_gram(ModeInfo& mi, const LoopEnd* close) {
  mi.facT.ensure((size_t)mi.rows * mi.R * sizeof(double));
  atb_small(mi.gram.d(), mi.fac.d(), mi.rows, mi.fac.d(), mi.rows, mi.rows, mi.R, mi.R, atbws_.d(), nullptr, stream_,
            mi.facT.d(), close);
  mi.facT_version = mi.version;
}

void Engine::prepare_mode_system(int m, int nrho, const aoadmm_options& opt) {
  ModeInfo& mi = modes_[m];
  TensorInfo& t = tensors_[mi.tensor];
  if (t.par2) {                      // first PARAFAC2 mode: same system, different A and C (:159-178)
    AO_REQUIRE(mi.pos == 0, "internal: only the first PARAFAC2 mode goes through the CP-style system");
    par2_prepare_modeA(m, nrho, opt);
    return;
  }
  FactorRef facs[8];
  for (int i = 0; i < t.nmodes; ++i) {
    const ModeInfo& o = modes_[t.modes[i]];
    facs[i] = factor_ref(o);
  }
  std::vector<int> seq = update_sequence(mi.tensor);
  SysBuild sb;
  sb.ngram = 0;
  for (int i = 0; i < t.nmodes; ++i)
    if (i != mi.pos) sb.grams[sb.ngram++] = modes_[t.modes[i]].gram.d();     // :98-103, :109,:112
  sb.Cpre = nullptr;
  sb.w = t.weight;
  sb.ridge = has_ridge_ ? mi.ridge : 0.0;
  sb.bsum_half = opt.bsum ? opt.bsum_weight / 2 : 0.0;
  sb.rho_scale = 1.0;
  sb.nrho = nrho;
  sb.R = mi.R;
  sb.C = mi.C.d(); sb.rho = mi.rho.d(); sb.Bsys = mi.Bsys.d(); sb.L = mi.L.d();
  sb.Binv = nrho > 0 ? mi.Binv.d() : nullptr;
  sb.ctl = ctl_of_mode(m);
  const int cty = mi.coupling >= 0 ? couplings_[mi.coupling].type : -1;
  if (cty == 2) sb.Madd = mi.HHt.d();
  // The system needs the Gram matrices only: it rides in the launch of the reduction that finishes the MTTKRP (one
  // extra workgroup) when that path is taken, else it gets its own launch behind the MTTKRP.
  static const bool no_rider = getenv("AOADMM_NO_SYS_RIDER") != nullptr;       // development switch
  bool rode = false;
  block_mttkrp(t.blk, mi.pos, facs, mi.R, t.weight, mi.A.d(), mi.rows, opt.use_dimtree != 0, seq.data(), (int)seq.size(), true,
               false, no_rider ? nullptr : &sb, &rode);
  if (!rode) sys_build(sb, stream_);
  if (cty == 1 || cty == 5) {                       // B = V diag(mu) V' for the Sylvester solve of the inner loop
    mi.eV.ensure((size_t)mi.R * mi.R * sizeof(double)); mi.eMu.ensure((size_t)mi.R * sizeof(double));
    sym_eig_small(mi.Bsys.d(), mi.R, mi.eMu.d(), mi.eV.d(), stream_);
  }
  t.last_pos = mi.pos;                                                        // :121-123
  mi.Aeff = mi.A.d();
  if (opt.bsum) {                                                             // :124-127
    Coef c[2] = {coef(1.0), coef(opt.bsum_weight / 2)};
    const double* x[2] = {mi.A.d(), mi.fac.d()};
    ew_lincomb(mi.Ab.d(), mi.rows * mi.R, 2, c, x, nullptr, stream_);
    mi.Aeff = mi.Ab.d();
  }
}

// see the end of the outer loop in solve(): the first uncoupled CP mode of the next iteration, prepared ahead
void Engine::prepare_next_first_mode(const aoadmm_options& opt) {
  static const bool off = getenv("AOADMM_NO_PREPARE_AHEAD") != nullptr;      // development switch
  if (off) return;
  for (int p = 0; p < n_tensors_; ++p)
    for (int m = 0; m < n_modes_; ++m) {
      const ModeInfo& mi = modes_[m];
      if (mi.coupling != -1 || mi.tensor != p) continue;
      if (tensors_[p].par2 && mi.pos != 0) return;  // a PARAFAC2 B_k or C mode comes first: nothing ahead
      prepare_mode_system(m, mi.constrained ? 1 : 0, opt);   // (the first PARAFAC2 mode goes through the same call, :159-178)
      prepared_mode_ = m;
      return;
    }
}

void Engine::update_uncoupled_cp_mode(int m, const aoadmm_options& opt) {
  ModeInfo& mi = modes_[m];
  if (prepared_mode_ == m) prepared_mode_ = -1;     // MTTKRP and system were enqueued at the end of the last iteration
  else prepare_mode_system(m, mi.constrained ? 1 : 0, opt);
  AdmmCtl* ctl = ctl_of_mode(m);
  LoopEnd le;
  GramFold gf;
  if (!mi.constrained) {
    // G.fac{m} = A{m}/B{m}  (:134): B is symmetric positive definite -> Cholesky solve
    row_solve(mi.fac.d(), mi.rows, mi.Aeff, mi.rows, mi.L.d(), mi.rows, mi.R, nullptr, stream_);
  } else if (admm_loop_wg_ok(mi.rows, mi.R, mi.prox.type, opt.MaxInnerIters)) {
    // short mode: loop, Gram matrix and row-major copy in one launch of one workgroup
    WgLoopU wa;
    wa.A = mi.Aeff; wa.Binv = mi.Binv.d(); wa.L = mi.L.d(); wa.rho = mi.rho.d(); wa.rho_prox = mi.rho.d();
    wa.fac = mi.fac.d(); wa.Z = mi.Z.d(); wa.mu = mi.mu.d();
    wa.rows = mi.rows; wa.R = mi.R; wa.per_row = 0;
    wa.ptype = mi.prox.type; wa.p0 = mi.prox.p0; wa.p1 = mi.prox.p1;
    wa.max_inner = opt.MaxInnerIters; wa.tol_pr = opt.innerRelPrTol_constr; wa.tol_du = opt.innerRelDualTol_constr;
    wa.ctl = ctl;
    mi.facT.ensure((size_t)mi.rows * mi.R * sizeof(double));
    wa.gram = mi.gram.d(); wa.facT = mi.facT.d();
    admm_loop_wg(wa, stream_);
    mi.version++;
    mi.facT_version = mi.version;
    return;
  } else {
    AdmmMode am;
    am.A = mi.Aeff; am.L = mi.L.d(); am.Binv = mi.Binv.d(); am.rho = mi.rho.d();
    am.fac = mi.fac.d(); am.Z = mi.Z.d(); am.mu = mi.mu.d();
    am.rows = mi.rows; am.R = mi.R; am.prox = mi.prox;
    mi.facT.ensure((size_t)mi.rows * mi.R * sizeof(double));
    atbws_.ensure((size_t)cdiv(mi.rows, 16) * mi.R * mi.R * sizeof(double));
    gf.ws = atbws_.d(); gf.At = mi.facT.d();
    admm_constrained_loop(am, mi.part.d(), mi.V.d(), mi.Znew.d(), mi.proxws.d(), ctl, opt.MaxInnerIters,
                          opt.innerRelPrTol_constr, opt.innerRelDualTol_constr, stream_, &le, &gf);
  }
  mi.version++;
  if (gf.nb > 0) {                                                            // :148, partials left by the loop's last launch
    atb_fin(mi.gram.d(), atbws_.d(), gf.nb, mi.R * mi.R, nullptr, stream_);
    mi.facT_version = mi.version;
  } else {
    compute_gram(mi, le.ctl ? &le : nullptr);                                 // :148
  }
}

// The six linear couplings (cmtf_fun_AOADMM.m:625-1075) in one form:  Tf_m(C_m) = Sd_m(Delta)
//   type 0: C = Delta | 1: H*C = Delta | 2: C*H = Delta | 3: C = H*Delta | 4: C = Delta*H | 5: H*C = Delta*H2
// Sd: the Delta-side image for mode m (shape img_rows x img_cols)
static const double* image_d(double* dst, const CouplingInfo& ci, const double* D, const ModeInfo& mi,
                             const AdmmCtl* ctl, hipStream_t s) {
  switch (ci.type) {
    case 3: gemm_small(dst, mi.rows, mi.H.d(), mi.hr, D, ci.rows, mi.rows, (int)ci.rows, mi.R, 0, coef(1.0), 0.0, ctl, s); return dst;
    case 4: gemm_small(dst, mi.rows, D, ci.rows, mi.H.d(), mi.hr, ci.rows, (int)ci.cols, mi.R, 0, coef(1.0), 0.0, ctl, s); return dst;
    case 5: gemm_small(dst, ci.rows, D, ci.rows, mi.H2.d(), mi.h2r, ci.rows, (int)ci.cols, mi.R, 0, coef(1.0), 0.0, ctl, s); return dst;
    default: return D;                               // types 0, 1, 2: Sd is the identity
  }
}
// Tf: the factor-side image (same shape); types 0, 3, 4 are the identity and return F itself
static const double* image_f(double* dst, const CouplingInfo& ci, const double* F, const ModeInfo& mi,
                             const AdmmCtl* ctl, hipStream_t s) {
  if (ci.type == 1 || ci.type == 5) {
    gemm_small(dst, mi.hr, mi.H.d(), mi.hr, F, mi.rows, mi.hr, (int)mi.rows, mi.R, 0, coef(1.0), 0.0, ctl, s);
    return dst;
  }
  if (ci.type == 2) {
    gemm_small(dst, mi.rows, F, mi.rows, mi.H.d(), mi.hr, mi.rows, mi.R, (int)mi.hc, 0, coef(1.0), 0.0, ctl, s);
    return dst;
  }
  return F;
}
// Tf': adjoint of the factor-side map applied to Y (img shape) -> rows x R ; identity for types 0, 3, 4
static const double* adjoint_f(double* dst, const CouplingInfo& ci, const double* Y, const ModeInfo& mi,
                               const AdmmCtl* ctl, hipStream_t s) {
  if (ci.type == 1 || ci.type == 5) {               // H' * Y
    gemm_small(dst, mi.rows, mi.Ht.d(), mi.hc, Y, mi.img_rows, mi.rows, (int)mi.hr, mi.R, 0, coef(1.0), 0.0, ctl, s);
    return dst;
  }
  if (ci.type == 2) {                                // Y * H'
    gemm_small(dst, mi.rows, Y, mi.rows, mi.H.d(), mi.hr, mi.rows, (int)mi.hc, mi.R, 1, coef(1.0), 0.0, ctl, s);
    return dst;
  }
  return Y;
}

__global__ void coupling_coefs_k(double* coef, const double* const* rhos, int n, AdmmCtl* ctl) {
  // coef[j] = rho_j / sum rho  (:661-675); also opens the coupled loop (what ctl_reset does: one launch fewer)
  if (threadIdx.x == 1) {
    ctl->active = 1;
    ctl->iters = 0;
    ctl->res[0] = ctl->res[1] = ctl->res[2] = ctl->res[3] = 0.0;
  }
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int j = 0; j < n; ++j) s += rhos[j][0];
    for (int j = 0; j < n; ++j) coef[j] = 1.0 / s * rhos[j][0];
    coef[n] = s;
  }
}

// mu_Delta += Tf(C) - Sd(Delta) (:679 and the same line of every case) with the sums the coupling residuals need in
// the same pass: out[0] = ||Tf(C) - Sd(Delta)||^2, out[1] = ||mu_Delta||^2, out[3] = ||den||^2 (den = Tf(C) or C,
// :1099-1210); out[2] (the dual numerator) is filled by the caller.  One workgroup for n <= 2048, else per-block
// partial sums added in block order by coupling_dual_fin_k.
__global__ __launch_bounds__(256) void coupling_dual_k(double* muD, const double* tf, const double* td, int64_t ni,
                                                       const double* den, int64_t nden, double* out, double* ws,
                                                       const AdmmCtl* ctl) {
  if (ctl != nullptr && ctl->active == 0) return;
  __shared__ double sh4[4];
  double s0 = 0, s1 = 0, s2 = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t i = first; i < ni; i += stride) {
    const double g = tf[i] - td[i];
    const double m = muD[i] + g;
    muD[i] = m;
    s0 += g * g; s1 += m * m;
  }
  for (int64_t i = first; i < nden; i += stride) s2 += den[i] * den[i];
  s0 = block256_sum(s0, sh4); s1 = block256_sum(s1, sh4); s2 = block256_sum(s2, sh4);
  if (threadIdx.x == 0) {
    if (gridDim.x == 1) { out[0] = s0; out[1] = s1; out[3] = s2; }
    else { double* w = ws + 3 * (int64_t)blockIdx.x; w[0] = s0; w[1] = s1; w[2] = s2; }
  }
}
__global__ void coupling_dual_fin_k(double* out, const double* ws, int nb, const AdmmCtl* ctl) {
  if (ctl != nullptr && ctl->active == 0) return;
  if (threadIdx.x >= 3) return;
  double t = 0.0;
  for (int b = 0; b < nb; ++b) t += ws[3 * b + threadIdx.x];
  out[threadIdx.x == 2 ? 3 : threadIdx.x] = t;
}

// Delta(k,:) = sum_j rho_j(k) * (C_j + mu_j)(k,:) / sum_j rho_j(k)   (:661-675): rho_j is a K-vector for a PARAFAC2
// C mode (vec[j] = 1) and a scalar otherwise
struct RowMeanArgs { const double* fac[8]; const double* mu[8]; const double* rho[8]; int vec[8]; int n; int64_t rows; int cols; };
__global__ void coupling_rowmean_k(double* Delta, RowMeanArgs a, const AdmmCtl* ctl) {
  if (ctl != nullptr && ctl->active == 0) return;
  const int64_t tot = a.rows * a.cols;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = e % a.rows;
    double acc = 0.0, sr = 0.0;
    for (int j = 0; j < a.n; ++j) {
      const double rj = a.vec[j] ? a.rho[j][k] : a.rho[j][0];
      acc += rj * (a.fac[j][e] + a.mu[j][e]);
      sr += rj;
    }
    Delta[e] = 1.0 / sr * acc;
  }
}

// out(k,c) = rho_k * in(k,c)  (rows of a K x cols matrix scaled by the rho vector of a PARAFAC2 C mode)
__global__ void rows_scale_k(double* out, const double* in, const double* rho, int64_t rows, int64_t cols,
                             const AdmmCtl* ctl) {
  if (ctl != nullptr && ctl->active == 0) return;
  const int64_t tot = rows * cols;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (int64_t)gridDim.x * blockDim.x)
    out[e] = rho[e % rows] * in[e];
}
static void rows_scale(double* out, const double* in, const double* rho, int64_t rows, int64_t cols, const AdmmCtl* ctl,
                       hipStream_t s) {
  int64_t nb = cdiv(rows * cols, 256);
  if (nb > 1024) nb = 1024;
  rows_scale_k<<<(unsigned)nb, 256, 0, s>>>(out, in, rho, rows, cols, ctl);
  AO_KERNEL_CHECK();
}
// Delta(k,:) = BB(k,:) / (AA + rho_k*AAA)   (:957-961): one workgroup per row, q x q system in LDS
__global__ void delta_rowwise_solve_k(double* Delta, const double* BB, int64_t rows, int q, const double* AA,
                                      const double* AAA, const double* rho, AdmmCtl* ctl) {
  if (ctl->active == 0) return;
  extern __shared__ double sh[];                      // q*q matrix, then q right-hand side
  double* M = sh;
  double* x = sh + q * q;
  const int64_t k = blockIdx.x;
  for (int e = threadIdx.x; e < q * q; e += blockDim.x) M[e] = AA[e] + rho[k] * AAA[e];
  for (int c = threadIdx.x; c < q; c += blockDim.x) x[c] = BB[k + rows * c];
  __syncthreads();
  const bool ok = chol_lds(M, q);
  if (!ok) { if (threadIdx.x == 0) ctl->notpd = 1; return; }
  if (threadIdx.x == 0) {                             // x * inv(L*L'): forward with L, backward with L'
    for (int c = 0; c < q; ++c) {
      double v = x[c];
      for (int p = 0; p < c; ++p) v -= M[c + q * p] * x[p];
      x[c] = v / M[c + q * c];
    }
    for (int c = q - 1; c >= 0; --c) {
      double v = x[c];
      for (int p = c + 1; p < q; ++p) v -= M[p + q * c] * x[p];
      x[c] = v / M[c + q * c];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < q; c += blockDim.x) Delta[k + rows * c] = x[c];
}

struct AAArgs { const double* H[8]; const double* rho[8]; int R[8]; int n; int Rc; };
__global__ void coupling_AA_k(double* AA, AAArgs a) {
  // AA = sum_j rho_j * H_j * H_j'   (:941-954 ; :1033-1047 with H2 and the common rhoC)
  const int Rc = a.Rc;
  for (int e = threadIdx.x; e < Rc * Rc; e += blockDim.x) {
    const int i = e % Rc, k = e / Rc;
    double acc = 0.0;
    for (int j = 0; j < a.n; ++j) {
      double t = 0.0;
      for (int q = 0; q < a.R[j]; ++q) t += a.H[j][i + Rc * q] * a.H[j][k + Rc * q];
      acc += a.rho[j][0] * t;
    }
    AA[e] = acc;
  }
}

// ---------------------------------------------------------------------------
// Couplings of type 0 (C = Delta) and 4 (C = Delta*H) are row-local: row i of every coupled factor, of Delta and of
// the duals only ever meets row i.  One thread per row then does a whole step in registers, which turns the
// 27 launches of an inner iteration (two modes, generic path below) into 11.  RMAX bounds both R and cols(Delta).
struct RowCouple {
  // per mode
  const double* Aeff; const double* L; const double* rho; const double* H;   // H: q x R (type 4), unused for type 0
  double* fac; double* muD; const double* Z; const double* mu;
  int R, constrained;
};
template <int RMAX>
__global__ __launch_bounds__(64) void couple_primal_rows_k(RowCouple m, const double* Delta, int64_t rows, int q, int type,
                                                           const AdmmCtl* ctl) {
  if (ctl != nullptr && ctl->active == 0) return;
  extern __shared__ double sh[];                      // L (R*R), H (q*R)
  const int R = m.R;
  double* Lsh = sh;
  double* Hsh = sh + R * R;
  for (int e = threadIdx.x; e < R * R; e += blockDim.x) Lsh[e] = m.L[e];
  if (type == 4)
    for (int e = threadIdx.x; e < q * R; e += blockDim.x) Hsh[e] = m.H[e];
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  const double rh = m.rho[0] / 2;
  double d[RMAX], x[RMAX];
#pragma unroll
  for (int c = 0; c < RMAX; ++c) d[c] = c < q ? Delta[i + rows * c] : 0.0;
#pragma unroll
  for (int r = 0; r < RMAX; ++r) {
    x[r] = 0.0;
    if (r < R) {
      double td;
      if (type == 4) {                                // (Delta*H)(i,r)  (:925)
        td = 0.0;
#pragma unroll
        for (int c = 0; c < RMAX; ++c)
          if (c < q) td += d[c] * Hsh[c + q * r];
      } else {
        td = d[r];                                    // :647
      }
      double v = m.Aeff[i + rows * r] + rh * (td - m.muD[i + rows * r]);
      if (m.constrained) v += rh * (m.Z[i + rows * r] - m.mu[i + rows * r]);
      x[r] = v;
    }
  }
#pragma unroll
  for (int r = 0; r < RMAX; ++r)                      // x * inv(L*L')  (:651, :929)
    if (r < R) {
      double v = x[r];
#pragma unroll
      for (int p = 0; p < RMAX; ++p)
        if (p < r) v -= Lsh[r + R * p] * x[p];
      x[r] = v / Lsh[r + R * r];
    }
#pragma unroll
  for (int r = RMAX - 1; r >= 0; --r)
    if (r < R) {
      double v = x[r];
#pragma unroll
      for (int p = 0; p < RMAX; ++p)
        if (p > r && p < R) v -= Lsh[p + R * r] * x[p];
      x[r] = v / Lsh[r + R * r];
    }
#pragma unroll
  for (int r = 0; r < RMAX; ++r)
    if (r < R) m.fac[i + rows * r] = x[r];
}

struct RowDelta {
  const double* fac[8]; const double* muD[8]; const double* rho[8]; const double* H[8];
  int R[8];
  int n;
};
// Delta_old = Delta ; Delta = weighted mean (type 0, :661-675) or BB / AA (type 4, :939-963) ; dD = Delta - Delta_old
template <int RMAX>
__global__ __launch_bounds__(64) void couple_delta_rows_k(RowDelta a, double* Delta, double* DeltaOld, double* dD,
                                                          const double* coefs, const double* LAA, int64_t rows, int q,
                                                          int type, const AdmmCtl* ctl) {
  if (ctl != nullptr && ctl->active == 0) return;
  extern __shared__ double sh[];                      // LAA (q*q), then H_j (q*R_j) back to back
  double* Lsh = sh;
  if (type == 4) {
    for (int e = threadIdx.x; e < q * q; e += blockDim.x) Lsh[e] = LAA[e];
    int off = q * q;
    for (int j = 0; j < a.n; ++j) {
      for (int e = threadIdx.x; e < q * a.R[j]; e += blockDim.x) sh[off + e] = a.H[j][e];
      off += q * a.R[j];
    }
  }
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  double bb[RMAX];
#pragma unroll
  for (int c = 0; c < RMAX; ++c) bb[c] = 0.0;
  int off = q * q;
  for (int j = 0; j < a.n; ++j) {
    if (type == 4) {
      const double rj = a.rho[j][0];
      double t[RMAX];
#pragma unroll
      for (int r = 0; r < RMAX; ++r) t[r] = r < a.R[j] ? a.fac[j][i + rows * r] + a.muD[j][i + rows * r] : 0.0;
      const double* Hj = sh + off;
#pragma unroll
      for (int c = 0; c < RMAX; ++c)
        if (c < q) {
          double acc = 0.0;
#pragma unroll
          for (int r = 0; r < RMAX; ++r)
            if (r < a.R[j]) acc += t[r] * Hj[c + q * r];
          bb[c] = (j == 0 ? 0.0 : bb[c]) + rj * acc;                               // :955, same order as the gemm path
        }
      off += q * a.R[j];
    } else {
      const double cj = coefs[j];                     // rho_j / sum rho
#pragma unroll
      for (int c = 0; c < RMAX; ++c)
        if (c < q) {
          const double v = cj * a.fac[j][i + rows * c] + cj * a.muD[j][i + rows * c];
          bb[c] = j == 0 ? v : bb[c] + v;
        }
    }
  }
  if (type == 4) {                                    // Delta(i,:) = bb * inv(LAA*LAA')
#pragma unroll
    for (int c = 0; c < RMAX; ++c)
      if (c < q) {
        double v = bb[c];
#pragma unroll
        for (int p = 0; p < RMAX; ++p)
          if (p < c) v -= Lsh[c + q * p] * bb[p];
        bb[c] = v / Lsh[c + q * c];
      }
#pragma unroll
    for (int c = RMAX - 1; c >= 0; --c)
      if (c < q) {
        double v = bb[c];
#pragma unroll
        for (int p = 0; p < RMAX; ++p)
          if (p > c && p < q) v -= Lsh[p + q * c] * bb[p];
        bb[c] = v / Lsh[c + q * c];
      }
  }
#pragma unroll
  for (int c = 0; c < RMAX; ++c)
    if (c < q) {
      const double old = Delta[i + rows * c];
      DeltaOld[i + rows * c] = old;
      Delta[i + rows * c] = bb[c];
      dD[i + rows * c] = bb[c] - old;
    }
}

// mu_Delta += C - Sd(Delta) and the four sums of the coupling residuals (:1099-1115, :1175-1191) for one mode:
// out[0] = ||C - Sd(Delta)||^2, out[1] = ||mu_Delta||^2, out[2] = ||Sd(dD)||^2, out[3] = ||C||^2
template <int RMAX>
__global__ __launch_bounds__(256) void couple_dual_rows_k(RowCouple m, const double* Delta, const double* dD, int64_t rows,
                                                          int q, int type, double* out, double* ws, const AdmmCtl* ctl) {
  if (ctl != nullptr && ctl->active == 0) return;
  extern __shared__ double sh[];                      // H (q*R)
  __shared__ double sh4[4];
  const int R = m.R;
  if (type == 4)
    for (int e = threadIdx.x; e < q * R; e += blockDim.x) sh[e] = m.H[e];
  __syncthreads();
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (int64_t)gridDim.x * blockDim.x) {
    double d[RMAX], dd[RMAX];
#pragma unroll
    for (int c = 0; c < RMAX; ++c) { d[c] = c < q ? Delta[i + rows * c] : 0.0; dd[c] = c < q ? dD[i + rows * c] : 0.0; }
#pragma unroll
    for (int r = 0; r < RMAX; ++r)
      if (r < R) {
        double td, tdd;
        if (type == 4) {
          td = 0.0; tdd = 0.0;
#pragma unroll
          for (int c = 0; c < RMAX; ++c)
            if (c < q) { td += d[c] * sh[c + q * r]; tdd += dd[c] * sh[c + q * r]; }
        } else { td = d[r]; tdd = dd[r]; }
        const double f = m.fac[i + rows * r];
        const double g = f - td;
        const double mm = m.muD[i + rows * r] + g;                                  // :679, :967
        m.muD[i + rows * r] = mm;
        s0 += g * g; s1 += mm * mm; s2 += tdd * tdd; s3 += f * f;
      }
  }
  s0 = block256_sum(s0, sh4); s1 = block256_sum(s1, sh4); s2 = block256_sum(s2, sh4); s3 = block256_sum(s3, sh4);
  if (threadIdx.x == 0) {
    double* o = gridDim.x == 1 ? out : ws + 4 * (int64_t)blockIdx.x;
    o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3;
  }
}
__global__ void couple_dual_fin_k(double* out, const double* ws, int nb, const AdmmCtl* ctl) {
  if (ctl != nullptr && ctl->active == 0) return;
  if (threadIdx.x >= 4) return;
  double t = 0.0;
  for (int b = 0; b < nb; ++b) t += ws[4 * b + threadIdx.x];
  out[threadIdx.x] = t;
}

// ---------------------------------------------------------------------------
// The whole inner loop of a row-local coupling (types 0 and 4) in ONE launch, for the sizes the example scripts use
// (rows of Delta up to a few thousand, ranks up to 16).  For these couplings every step of an inner iteration --
// the primal solves of all coupled modes (:647-651, :925-929), the Delta update (:661-675, :939-963), the coupling
// duals (:679, :967) and, with an element-/row-wise prox, update_constraint (:1420-1429) -- touches row i of every
// matrix only, so thread i carries row i through the iteration without meeting another thread; the workgroup meets
// once per iteration to add up the residual sums (:1099-1115, :1175-1191, :1079-1096) and to evaluate the while
// condition (:630).  The launch-per-step form (couple_primal / couple_delta / couple_dual + constraint_update +
// finalize: 12 launches per inner iteration for two constrained modes) is kept for larger problems.
struct WgLoopMode {
  const double* Aeff; const double* L; const double* rho; const double* H;
  double *fac, *muD, *Z, *mu, *Zold;
  double* slots;        // 8 residual sums of this mode (see FinalizeArgs)
  int R, constrained, ptype;
  double p0, p1;
};
struct WgLoopArgs {
  WgLoopMode m[4];
  int n, q, type, max_inner;
  int64_t rows;
  double *Delta, *DeltaOld, *dD;
  const double* coefs;
  const double* LAA;
  double tol_pr_coupl, tol_pr_constr, tol_du_coupl, tol_du_constr;
  AdmmCtl* ctl;
  int self_start = 0;       // registers kernel: opens the loop itself and takes rho_j / sum rho from the modes' rho
                            // (no ctl_reset / coupling_coefs_k launch in front of it)
};
template <int RMAX>
__global__ __launch_bounds__(256) void couple_loop_wg_k(WgLoopArgs a) {
  extern __shared__ double sh[];                      // LAA (q*q) | per mode: L (R*R), H (q*R)
  __shared__ double red[4][32];
  __shared__ int go;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int q = a.q, type = a.type;
  const int64_t rows = a.rows;
  int offL[4], offH[4];
  {
    int off = q * q;
    for (int j = 0; j < a.n; ++j) { offL[j] = off; off += a.m[j].R * a.m[j].R; offH[j] = off; off += q * a.m[j].R; }
    if (type == 4)
      for (int e = t; e < q * q; e += 256) sh[e] = a.LAA[e];
    for (int j = 0; j < a.n; ++j) {
      const int R = a.m[j].R;
      for (int e = t; e < R * R; e += 256) sh[offL[j] + e] = a.m[j].L[e];
      if (type == 4)
        for (int e = t; e < q * R; e += 256) sh[offH[j] + e] = a.m[j].H[e];
    }
  }
  if (t == 0) go = a.ctl->active;
  __syncthreads();
  int it = 0;
  while (go) {
    double sums[4][8];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < 8; ++k) sums[j][k] = 0.0;
    for (int64_t i = t; i < rows; i += 256) {
      double d[RMAX];
#pragma unroll
      for (int c = 0; c < RMAX; ++c) d[c] = c < q ? a.Delta[i + rows * c] : 0.0;
      // ---- primal updates
      for (int j = 0; j < a.n; ++j) {
        const WgLoopMode& m = a.m[j];
        const int R = m.R;
        const double* Lsh = sh + offL[j];
        const double* Hsh = sh + offH[j];
        const double rh = m.rho[0] / 2;
        double x[RMAX];
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
          x[r] = 0.0;
          if (r < R) {
            double td;
            if (type == 4) {                          // (Delta*H)(i,r)  (:925)
              td = 0.0;
#pragma unroll
              for (int c = 0; c < RMAX; ++c)
                if (c < q) td += d[c] * Hsh[c + q * r];
            } else {
              td = d[r];                              // :647
            }
            double v = m.Aeff[i + rows * r] + rh * (td - m.muD[i + rows * r]);
            if (m.constrained) v += rh * (m.Z[i + rows * r] - m.mu[i + rows * r]);
            x[r] = v;
          }
        }
#pragma unroll
        for (int r = 0; r < RMAX; ++r)                // x * inv(L*L')  (:651, :929)
          if (r < R) {
            double v = x[r];
#pragma unroll
            for (int p = 0; p < RMAX; ++p)
              if (p < r) v -= Lsh[r + R * p] * x[p];
            x[r] = v / Lsh[r + R * r];
          }
#pragma unroll
        for (int r = RMAX - 1; r >= 0; --r)
          if (r < R) {
            double v = x[r];
#pragma unroll
            for (int p = 0; p < RMAX; ++p)
              if (p > r && p < R) v -= Lsh[p + R * r] * x[p];
            x[r] = v / Lsh[r + R * r];
          }
#pragma unroll
        for (int r = 0; r < RMAX; ++r)
          if (r < R) m.fac[i + rows * r] = x[r];
      }
      // ---- Delta
      double bb[RMAX];
#pragma unroll
      for (int c = 0; c < RMAX; ++c) bb[c] = 0.0;
      for (int j = 0; j < a.n; ++j) {
        const WgLoopMode& m = a.m[j];
        if (type == 4) {
          const double rj = m.rho[0];
          double tt[RMAX];
#pragma unroll
          for (int r = 0; r < RMAX; ++r) tt[r] = r < m.R ? m.fac[i + rows * r] + m.muD[i + rows * r] : 0.0;
          const double* Hj = sh + offH[j];
#pragma unroll
          for (int c = 0; c < RMAX; ++c)
            if (c < q) {
              double acc = 0.0;
#pragma unroll
              for (int r = 0; r < RMAX; ++r)
                if (r < m.R) acc += tt[r] * Hj[c + q * r];
              bb[c] = (j == 0 ? 0.0 : bb[c]) + rj * acc;                           // :955
            }
        } else {
          const double cj = a.coefs[j];               // rho_j / sum rho
#pragma unroll
          for (int c = 0; c < RMAX; ++c)
            if (c < q) {
              const double v = cj * m.fac[i + rows * c] + cj * m.muD[i + rows * c];
              bb[c] = j == 0 ? v : bb[c] + v;
            }
        }
      }
      if (type == 4) {                                // Delta(i,:) = bb * inv(LAA*LAA')
        const double* Lsh = sh;
#pragma unroll
        for (int c = 0; c < RMAX; ++c)
          if (c < q) {
            double v = bb[c];
#pragma unroll
            for (int p = 0; p < RMAX; ++p)
              if (p < c) v -= Lsh[c + q * p] * bb[p];
            bb[c] = v / Lsh[c + q * c];
          }
#pragma unroll
        for (int c = RMAX - 1; c >= 0; --c)
          if (c < q) {
            double v = bb[c];
#pragma unroll
            for (int p = 0; p < RMAX; ++p)
              if (p > c && p < q) v -= Lsh[p + q * c] * bb[p];
            bb[c] = v / Lsh[c + q * c];
          }
      }
      double dd[RMAX];
#pragma unroll
      for (int c = 0; c < RMAX; ++c) {
        dd[c] = 0.0;
        if (c < q) {
          a.DeltaOld[i + rows * c] = d[c];
          a.Delta[i + rows * c] = bb[c];
          dd[c] = bb[c] - d[c];
          a.dD[i + rows * c] = dd[c];
        }
      }
      // ---- coupling duals, constraints, residual sums
      for (int j = 0; j < a.n; ++j) {
        const WgLoopMode& m = a.m[j];
        const int R = m.R;
        const double* Hsh = sh + offH[j];
        double f[RMAX];
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
          f[r] = 0.0;
          if (r < R) {
            double td, tdd;
            if (type == 4) {
              td = 0.0; tdd = 0.0;
#pragma unroll
              for (int c = 0; c < RMAX; ++c)
                if (c < q) { td += bb[c] * Hsh[c + q * r]; tdd += dd[c] * Hsh[c + q * r]; }
            } else { td = bb[r]; tdd = dd[r]; }
            f[r] = m.fac[i + rows * r];
            const double g = f[r] - td;
            const double mm = m.muD[i + rows * r] + g;                              // :679, :967
            m.muD[i + rows * r] = mm;
            sums[j][4] += g * g; sums[j][5] += mm * mm; sums[j][6] += tdd * tdd; sums[j][7] += f[r] * f[r];
          }
        }
        if (m.constrained) {                          // update_constraint (:1420-1429)
          const double rho = m.rho[0];
          double zo[RMAX], mu[RMAX], z[RMAX];
#pragma unroll
          for (int r = 0; r < RMAX; ++r) {
            zo[r] = r < R ? m.Z[i + rows * r] : 0.0;
            mu[r] = r < R ? m.mu[i + rows * r] : 0.0;
            z[r] = f[r] + mu[r];
          }
          if (m.ptype == AOADMM_C_SIMPLEX_ROW) {
            simplex_regs<RMAX>(z, R, m.p0);
          } else {
#pragma unroll
            for (int r = 0; r < RMAX; ++r) z[r] = prox_elem(m.ptype, z[r], m.p0, m.p1, rho);
          }
#pragma unroll
          for (int r = 0; r < RMAX; ++r)
            if (r < R) {
              const double mn = mu[r] + f[r] - z[r];
              m.Zold[i + rows * r] = zo[r];
              m.Z[i + rows * r] = z[r];
              m.mu[i + rows * r] = mn;
              const double dz = z[r] - zo[r];
              sums[j][0] += (f[r] - z[r]) * (f[r] - z[r]); sums[j][1] += f[r] * f[r]; sums[j][2] += mn * mn; sums[j][3] += dz * dz;
            }
        } else {
#pragma unroll
          for (int r = 0; r < RMAX; ++r) sums[j][1] += f[r] * f[r];
        }
      }
    }
    // ---- the workgroup's sums (fixed order: lanes by DPP tree, then the four waves in order)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const double v = wave_sum(sums[j][k]);
        if (lane == 0) red[w][j * 8 + k] = v;
      }
    __syncthreads();
    if (t < 32) {
      const double tot = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
      red[0][t] = tot;
      const int j = t >> 3;
      if (j < a.n) a.m[j].slots[t & 7] = tot;
    }
    __syncthreads();
    if (t == 0) {                                     // eval_res_ADMM_coupl_case0/4 + eval_res_ADMM_constr, while condition
      double prc = 0, duc = 0, prz = 0, duz = 0;
      int nz = 0;
      for (int j = 0; j < a.n; ++j) {
        const double* sj = &red[0][j * 8];
        prc += sqrt(sj[4]) / sqrt(sj[7]);
        const double sc = sqrt(sj[5]);
        duc += sc > 0 ? sqrt(sj[6]) / sc : sqrt(sj[6]);
        if (a.m[j].constrained) {
          prz += sqrt(sj[0]) / sqrt(sj[1]);
          const double sz = sqrt(sj[2]);
          duz += sz > 0 ? sqrt(sj[3]) / sz : sqrt(sj[3]);
          ++nz;
        }
      }
      prc /= a.n; duc /= a.n;
      if (nz) { prz /= nz; duz /= nz; }
      ++it;
      a.ctl->res[0] = prc; a.ctl->res[1] = prz; a.ctl->res[2] = duc; a.ctl->res[3] = duz;
      a.ctl->iters = it;
      const int cont = (it < a.max_inner && (prc > a.tol_pr_coupl || prz > a.tol_pr_constr || duc > a.tol_du_coupl ||
                                             duz > a.tol_du_constr)) ? 1 : 0;
      a.ctl->active = cont;
      go = cont;
    }
    __syncthreads();
  }
}

// Register-resident form of couple_loop_wg_k for rows <= 256 (one row per thread) and NM coupled modes: the rows of A,
// fac, mu_Delta, Z, mu of every coupled mode and the row of Delta are loaded once, live in registers for the whole
// loop and are stored once.  An inner iteration is then arithmetic plus one workgroup reduction, with no memory round
// trip (the global-memory form re-reads its own stores from L2 several times per iteration: 25 us per iteration
// against a few us here at 50 rows x 4 columns).
// Everything is padded to RMAX with zeros -- the small matrices in LDS (L_j, H_j, L_AA as RMAX x RMAX blocks), the
// reciprocal diagonals (0 beyond the rank) and the register rows -- so the loop body is straight-line code: a padded
// column contributes exact zeros to every sum and is never stored.  (The first version tested `r < R` and `c < q` at
// every step: ~5000 instructions, 700 of them branches, 10 us per inner iteration; PMC: 15 cycles per instruction on
// the one wave per SIMD.)  T4: coupling type 4 (C = Delta*H), else type 0 (C = Delta).
template <int RMAX, int NM, bool T4>
__global__ __launch_bounds__(256) void couple_loop_wg_regs_k(WgLoopArgs a) {
  constexpr int RR = RMAX * RMAX;
  __shared__ double Lsh[NM][RR];                      // L_j, column-major with leading dimension RMAX
  __shared__ double Hsh[NM][RR];                      // H_j(c, r) at c + RMAX*r
  __shared__ double LAAsh[RR];
  __shared__ double red[4][8 * NM];
  __shared__ double invd[NM + 1][RMAX];               // reciprocal diagonals of L_j and of LAA: the substitutions multiply
  __shared__ int go;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int q = a.q;
  const int64_t rows = a.rows;
  const int64_t i = t;
  const bool have = i < rows;
  const int64_t ic = have ? i : rows - 1;             // clamped: padding threads compute on a valid row, store nothing
  for (int e = t; e < RR; e += 256) {
    const int r = e % RMAX, c = e / RMAX;
    LAAsh[e] = (T4 && r < q && c < q) ? a.LAA[r + q * c] : 0.0;
#pragma unroll
    for (int j = 0; j < NM; ++j) {
      const int R = a.m[j].R;
      Lsh[j][e] = (r < R && c < R) ? a.m[j].L[r + R * c] : 0.0;
      Hsh[j][e] = (T4 && r < q && c < R) ? a.m[j].H[r + q * c] : 0.0;
    }
  }
  __syncthreads();
  if (t < RMAX) {
#pragma unroll
    for (int j = 0; j < NM; ++j) invd[j][t] = t < a.m[j].R ? 1.0 / Lsh[j][t + RMAX * t] : 0.0;
    invd[NM][t] = (T4 && t < q) ? 1.0 / LAAsh[t + RMAX * t] : 0.0;
  }
  double d[RMAX], av[NM][RMAX], f[NM][RMAX], md[NM][RMAX], z[NM][RMAX], mu[NM][RMAX], zo[NM][RMAX], rh[NM], rho[NM], cj[NM];
  ElemProx ep[NM];
#pragma unroll
  for (int c = 0; c < RMAX; ++c) d[c] = c < q ? a.Delta[ic + rows * c] : 0.0;
#pragma unroll
  for (int j = 0; j < NM; ++j) {
    const WgLoopMode& m = a.m[j];
    rho[j] = m.rho[0];
    rh[j] = rho[j] / 2;
    cj[j] = (T4 || a.self_start) ? 0.0 : a.coefs[j];  // rho_j / sum rho
    ep[j] = elem_prox_of(m.ptype, m.p0, m.p1, rho[j]);
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
      const bool ok = r < m.R;
      const int64_t o = ic + rows * (ok ? r : 0);
      av[j][r] = ok ? m.Aeff[o] : 0.0;
      f[j][r] = ok ? m.fac[o] : 0.0;
      md[j][r] = ok ? m.muD[o] : 0.0;
      z[j][r] = (ok && m.constrained) ? m.Z[o] : 0.0;
      mu[j][r] = (ok && m.constrained) ? m.mu[o] : 0.0;
      zo[j][r] = z[j][r];
    }
  }
  if (!T4 && a.self_start) {                          // coupling_coefs_k's arithmetic: 1 / sum(rho) * rho_j, modes in order
    double srho = 0.0;
#pragma unroll
    for (int j = 0; j < NM; ++j) srho += rho[j];
#pragma unroll
    for (int j = 0; j < NM; ++j) cj[j] = 1.0 / srho * rho[j];
  }
  double dold[RMAX], dd[RMAX];
#pragma unroll
  for (int c = 0; c < RMAX; ++c) { dold[c] = d[c]; dd[c] = 0.0; }
  if (t == 0) go = a.self_start ? 1 : a.ctl->active;
  __syncthreads();
  int it = 0;
  bool ran = false;
  while (go) {
    ran = true;
    double sums[NM][8];
#pragma unroll
    for (int j = 0; j < NM; ++j)
#pragma unroll
      for (int k = 0; k < 8; ++k) sums[j][k] = 0.0;
    // ---- primal updates
#pragma unroll
    for (int j = 0; j < NM; ++j) {
      const WgLoopMode& m = a.m[j];
      double x[RMAX];
#pragma unroll
      for (int r = 0; r < RMAX; ++r) {
        double td;
        if (T4) {                                     // (Delta*H)(i,r)  (:925)
          td = 0.0;
#pragma unroll
          for (int c = 0; c < RMAX; ++c) td += d[c] * Hsh[j][c + RMAX * r];
        } else {
          td = d[r];                                  // :647
        }
        double v = av[j][r] + rh[j] * (td - md[j][r]);
        if (m.constrained) v += rh[j] * (z[j][r] - mu[j][r]);
        x[r] = v;
      }
#pragma unroll
      for (int r = 0; r < RMAX; ++r) {                // x * inv(L*L')  (:651, :929)
        double v = x[r];
#pragma unroll
        for (int p = 0; p < r; ++p) v -= Lsh[j][r + RMAX * p] * x[p];
        x[r] = v * invd[j][r];
      }
#pragma unroll
      for (int r = RMAX - 1; r >= 0; --r) {
        double v = x[r];
#pragma unroll
        for (int p = r + 1; p < RMAX; ++p) v -= Lsh[j][p + RMAX * r] * x[p];
        x[r] = v * invd[j][r];
      }
#pragma unroll
      for (int r = 0; r < RMAX; ++r) f[j][r] = x[r];
    }
    // ---- Delta
    double bb[RMAX];
#pragma unroll
    for (int c = 0; c < RMAX; ++c) bb[c] = 0.0;
#pragma unroll
    for (int j = 0; j < NM; ++j) {
      if (T4) {
#pragma unroll
        for (int c = 0; c < RMAX; ++c) {
          double acc = 0.0;
#pragma unroll
          for (int r = 0; r < RMAX; ++r) acc += (f[j][r] + md[j][r]) * Hsh[j][c + RMAX * r];
          bb[c] = (j == 0 ? 0.0 : bb[c]) + rho[j] * acc;                           // :955
        }
      } else {
#pragma unroll
        for (int c = 0; c < RMAX; ++c) {
          const double v = cj[j] * f[j][c] + cj[j] * md[j][c];
          bb[c] = j == 0 ? v : bb[c] + v;
        }
      }
    }
    if (T4) {                                         // Delta(i,:) = bb * inv(LAA*LAA')
#pragma unroll
      for (int c = 0; c < RMAX; ++c) {
        double v = bb[c];
#pragma unroll
        for (int p = 0; p < c; ++p) v -= LAAsh[c + RMAX * p] * bb[p];
        bb[c] = v * invd[NM][c];
      }
#pragma unroll
      for (int c = RMAX - 1; c >= 0; --c) {
        double v = bb[c];
#pragma unroll
        for (int p = c + 1; p < RMAX; ++p) v -= LAAsh[p + RMAX * c] * bb[p];
        bb[c] = v * invd[NM][c];
      }
    }
#pragma unroll
    for (int c = 0; c < RMAX; ++c) {
      const double nv = (T4 || c < q) ? bb[c] : 0.0;  // type 0: columns beyond q carry nothing
      dold[c] = d[c];
      dd[c] = nv - d[c];
      d[c] = nv;
    }
    // ---- coupling duals, constraints, residual sums
#pragma unroll
    for (int j = 0; j < NM; ++j) {
      const WgLoopMode& m = a.m[j];
      const int R = m.R;
#pragma unroll
      for (int r = 0; r < RMAX; ++r) {
        double td, tdd;
        if (T4) {
          td = 0.0; tdd = 0.0;
#pragma unroll
          for (int c = 0; c < RMAX; ++c) { td += d[c] * Hsh[j][c + RMAX * r]; tdd += dd[c] * Hsh[j][c + RMAX * r]; }
        } else { td = r < R ? d[r] : 0.0; tdd = r < R ? dd[r] : 0.0; }
        const double g = f[j][r] - td;
        const double mm = md[j][r] + g;                                             // :679, :967
        md[j][r] = mm;
        if (have) { sums[j][4] += g * g; sums[j][5] += mm * mm; sums[j][6] += tdd * tdd; sums[j][7] += f[j][r] * f[j][r]; }
      }
      if (m.constrained) {                            // update_constraint (:1420-1429)
        double zn[RMAX];
#pragma unroll
        for (int r = 0; r < RMAX; ++r) { zo[j][r] = z[j][r]; zn[r] = f[j][r] + mu[j][r]; }
        if (m.ptype == AOADMM_C_SIMPLEX_ROW) {
          simplex_regs<RMAX>(zn, R, m.p0);
        } else {
#pragma unroll
          for (int r = 0; r < RMAX; ++r) zn[r] = r < R ? elem_prox(ep[j], zn[r]) : 0.0;
        }
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
          const double mn = mu[j][r] + f[j][r] - zn[r];
          const double dz = zn[r] - zo[j][r];
          if (have) {
            sums[j][0] += (f[j][r] - zn[r]) * (f[j][r] - zn[r]); sums[j][1] += f[j][r] * f[j][r]; sums[j][2] += mn * mn;
            sums[j][3] += dz * dz;
          }
          z[j][r] = zn[r];
          mu[j][r] = mn;
        }
      } else if (have) {
#pragma unroll
        for (int r = 0; r < RMAX; ++r) sums[j][1] += f[j][r] * f[j][r];
      }
    }
    // ---- the workgroup's sums (fixed order: lanes by DPP tree, then the four waves in order)
#pragma unroll
    for (int j = 0; j < NM; ++j)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const double v = wave_sum(sums[j][k]);
        if (lane == 0) red[w][j * 8 + k] = v;
      }
    __syncthreads();
    if (t < 8 * NM) {
      const double tot = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
      red[0][t] = tot;
      a.m[t >> 3].slots[t & 7] = tot;
    }
    __syncthreads();
    // eval_res_ADMM_coupl_case0/4 + eval_res_ADMM_constr: the 4*NM ratios (two square roots and a division each, ~100
    // dependent fp64 instructions) on 4*NM lanes side by side, then one lane adds them in mode order and decides --
    // the whole workgroup waits for this
    if (t < 4 * NM) {
      const int j = t >> 2, which = t & 3;            // 0: primal coupling, 1: dual coupling, 2: primal constr., 3: dual constr.
      const double* sj = &red[0][j * 8];
      double v;
      if (which == 0) v = sqrt(sj[4]) / sqrt(sj[7]);
      else if (which == 2) v = sqrt(sj[0]) / sqrt(sj[1]);
      else {
        const double num = sqrt(which == 1 ? sj[6] : sj[3]), sc = sqrt(which == 1 ? sj[5] : sj[2]);
        v = sc > 0 ? num / sc : num;
      }
      red[1][t] = v;
    }
    __syncthreads();
    if (t == 0) {                                     // while condition
      double prc = 0, duc = 0, prz = 0, duz = 0;
      int nz = 0;
      for (int j = 0; j < NM; ++j) {
        prc += red[1][4 * j];
        duc += red[1][4 * j + 1];
        if (a.m[j].constrained) {
          prz += red[1][4 * j + 2];
          duz += red[1][4 * j + 3];
          ++nz;
        }
      }
      prc /= NM; duc /= NM;
      if (nz) { prz /= nz; duz /= nz; }
      ++it;
      a.ctl->res[0] = prc; a.ctl->res[1] = prz; a.ctl->res[2] = duc; a.ctl->res[3] = duz;
      a.ctl->iters = it;
      const int cont = (it < a.max_inner && (prc > a.tol_pr_coupl || prz > a.tol_pr_constr || duc > a.tol_du_coupl ||
                                             duz > a.tol_du_constr)) ? 1 : 0;
      a.ctl->active = cont;
      go = cont;
    }
    __syncthreads();
  }
  if (!ran || !have) return;
#pragma unroll
  for (int c = 0; c < RMAX; ++c)
    if (c < q) { a.Delta[i + rows * c] = d[c]; a.DeltaOld[i + rows * c] = dold[c]; a.dD[i + rows * c] = dd[c]; }
#pragma unroll
  for (int j = 0; j < NM; ++j) {
    const WgLoopMode& m = a.m[j];
#pragma unroll
    for (int r = 0; r < RMAX; ++r)
      if (r < m.R) {
        const int64_t o = i + rows * r;
        m.fac[o] = f[j][r];
        m.muD[o] = md[j][r];
        if (m.constrained) { m.Z[o] = z[j][r]; m.mu[o] = mu[j][r]; m.Zold[o] = zo[j][r]; }
      }
  }
}

void Engine::coupled_admm(int c, const aoadmm_options& opt) {
  CouplingInfo& ci = couplings_[c];
  AdmmCtl* ctl = ctl_of_coupling(c);
  const int n = (int)ci.modes.size();
  const int ty = ci.type;
  const int64_t nD = ci.rows * ci.cols;
  ci.DeltaOld.ensure(nD * 8); ci.BB.ensure(nD * 8); ci.dD.ensure(nD * 8); ci.tmp.ensure(nD * 8);
  ci.coef.ensure(64 * 8);
  const int64_t qa = ty == 3 ? ci.rows : ci.cols;    // order of the Delta normal equations (types 3 / 4, 5)
  ci.AA.ensure((size_t)qa * qa * 8); ci.LAA.ensure((size_t)qa * qa * 8);
  double* resid = slots_.d() + n_modes_ * kSlotsPerMode + 2 * n_tensors_;
  for (int j = 0; j < n; ++j) {                       // image-shaped work buffers
    ModeInfo& mi = modes_[ci.modes[j]];
    const size_t nimg = (size_t)std::max(mi.rows * mi.R, mi.img_rows * mi.img_cols) * sizeof(double);
    mi.TD.ensure(nimg); mi.TF.ensure(nimg); mi.tmp.ensure(nimg); mi.W1.ensure(nimg); mi.W2.ensure(nimg);
  }
  // the one-launch registers loop (couple_loop_wg_regs_k) opens the loop and forms rho_j / sum rho itself
  bool regs_path = false;
  {
    static const bool off = getenv("AOADMM_GENERIC_COUPLING") != nullptr, no_wg = getenv("AOADMM_NO_WG_LOOP") != nullptr;
    int rmax = (int)ci.cols;
    bool local_prox = true, pc = false;
    for (int j = 0; j < n; ++j) {
      const ModeInfo& mi = modes_[ci.modes[j]];
      rmax = std::max(rmax, mi.R);
      local_prox = local_prox && (!mi.constrained || prox_is_fusable(mi.prox.type));
      pc = pc || (tensors_[mi.tensor].par2 && mi.pos == 2);
    }
    regs_path = (ty == 0 || ty == 4) && !pc && !off && !no_wg && local_prox && n <= 3 && ci.rows <= 256 && rmax <= 8;
  }
  // reset the loop control (the per-mode sys_build calls reset their own blocks); types 0-2: in coupling_coefs_k below
  if (!(ty == 0 || ty == 1 || ty == 2) && !regs_path) ctl_reset(ctl, stream_);
  // per-outer-iteration constants
  std::vector<const double*> hp(n);
  bool any_pc = false;                                // a PARAFAC2 C mode in this coupling (types 0 and 1 only)
  auto pc_block = [&](const ModeInfo& mi) -> Par2Block* {
    return (tensors_[mi.tensor].par2 && mi.pos == 2) ? &tensors_[mi.tensor].p2 : nullptr;
  };
  for (int j = 0; j < n; ++j) {
    const ModeInfo& mj = modes_[ci.modes[j]];
    Par2Block* pb = pc_block(mj);
    hp[j] = pb ? pb->rhosum.d() : mj.rho.d();         // type 1 weighs a C mode with sum(rho) (:736)
    any_pc = any_pc || pb != nullptr;
  }
  DevBuf& rho_ptrs = ci.rho_ptrs;                     // pointers never change once the work buffers exist
  if (ci.rho_ptrs_host != hp) {
    rho_ptrs.ensure(8 * sizeof(double*));
    AO_HIP(hipMemcpyAsync(rho_ptrs.p, hp.data(), n * sizeof(double*), hipMemcpyHostToDevice, stream_));
    AO_HIP(hipStreamSynchronize(stream_));            // hp is a local
    ci.rho_ptrs_host = hp;
  }
  const double* rho_last = modes_[ci.modes[n - 1]].rho.d();   // type 5: rhoC = mean(rho{mm}) with the stale loop variable (:1032)
  if (ty == 0 || ty == 1 || ty == 2) {
    if (!regs_path) {
      coupling_coefs_k<<<1, 64, 0, stream_>>>(ci.coef.d(), rho_ptrs.as<const double*>(), n, ctl);
      AO_KERNEL_CHECK();
    }
  } else if (ty == 4 || ty == 5) {
    AAArgs aa, aaa;                                   // aaa: the PARAFAC2 C mode's H*H' kept apart (:946-948)
    aa.n = 0; aa.Rc = (int)ci.cols; aaa.n = 0; aaa.Rc = (int)ci.cols;
    for (int j = 0; j < n; ++j) {
      const ModeInfo& mj = modes_[ci.modes[j]];
      AAArgs& dst = pc_block(mj) ? aaa : aa;
      dst.H[dst.n] = ty == 4 ? mj.H.d() : mj.H2.d();
      dst.rho[dst.n] = (&dst == &aaa) ? ones_.d() : (ty == 4 ? hp[j] : rho_last);
      dst.R[dst.n] = mj.R;
      dst.n++;
    }
    coupling_AA_k<<<1, 256, 0, stream_>>>(ci.AA.d(), aa);
    AO_KERNEL_CHECK();
    if (aaa.n > 0) {                                  // LAA holds AAA; the per-row systems are factored in the Delta step
      coupling_AA_k<<<1, 256, 0, stream_>>>(ci.LAA.d(), aaa);
      AO_KERNEL_CHECK();
    } else
    chol_only(ci.LAA.d(), ci.AA.d(), (int)ci.cols, ctl, stream_);
  }
  // ---- row-local fast path (types 0 and 4, ranks and cols(Delta) up to 16, no PARAFAC2 C mode)
  {
    int rmax = (int)ci.cols;
    for (int j = 0; j < n; ++j) rmax = std::max(rmax, modes_[ci.modes[j]].R);
    static const bool off = getenv("AOADMM_GENERIC_COUPLING") != nullptr;      // development switch
    if ((ty == 0 || ty == 4) && !any_pc && rmax <= 16 && !off) {
      const int q = (int)ci.cols;
      const int64_t rows = ci.rows;
      const unsigned rb = (unsigned)cdiv(rows, 64);
      int64_t nr = cdiv(rows, 2048);
      if (nr > 64) nr = 64;
      RowCouple rc[8];
      RowDelta rd;
      rd.n = n;
      size_t lds_delta = (size_t)q * q;
      for (int j = 0; j < n; ++j) {
        ModeInfo& mi = modes_[ci.modes[j]];
        rc[j].Aeff = mi.Aeff; rc[j].L = mi.L.d(); rc[j].rho = mi.rho.d(); rc[j].H = ty == 4 ? mi.H.d() : nullptr;
        rc[j].fac = mi.fac.d(); rc[j].muD = mi.muD.d(); rc[j].Z = mi.Z.d(); rc[j].mu = mi.mu.d();
        rc[j].R = mi.R; rc[j].constrained = mi.constrained ? 1 : 0;
        rd.fac[j] = mi.fac.d(); rd.muD[j] = mi.muD.d(); rd.rho[j] = mi.rho.d(); rd.H[j] = rc[j].H; rd.R[j] = mi.R;
        lds_delta += (size_t)q * mi.R;
      }
      auto by_rmax = [&](auto&& launch) {
        if (rmax <= 4) launch(std::integral_constant<int, 4>());
        else if (rmax <= 8) launch(std::integral_constant<int, 8>());
        else launch(std::integral_constant<int, 16>());
      };
      // small problems: the whole loop in one launch of one workgroup (couple_loop_wg_k)
      bool local_prox = true;
      for (int j = 0; j < n; ++j) {
        const ModeInfo& mi = modes_[ci.modes[j]];
        local_prox = local_prox && (!mi.constrained || prox_is_fusable(mi.prox.type));
      }
      static const bool no_wg = getenv("AOADMM_NO_WG_LOOP") != nullptr;           // development switch
      if (n <= 4 && rows <= 2048 && local_prox && !no_wg) {
        WgLoopArgs wa;
        wa.n = n; wa.q = q; wa.type = ty; wa.max_inner = opt.MaxInnerIters; wa.rows = rows;
        wa.Delta = ci.Delta.d(); wa.DeltaOld = ci.DeltaOld.d(); wa.dD = ci.dD.d(); wa.coefs = ci.coef.d(); wa.LAA = ci.LAA.d();
        wa.tol_pr_coupl = opt.innerRelPrTol_coupl; wa.tol_pr_constr = opt.innerRelPrTol_constr;
        wa.tol_du_coupl = opt.innerRelDualTol_coupl; wa.tol_du_constr = opt.innerRelDualTol_constr;
        wa.ctl = ctl;
        size_t lds = (size_t)q * q;
        for (int j = 0; j < n; ++j) {
          ModeInfo& mi = modes_[ci.modes[j]];
          WgLoopMode& wm = wa.m[j];
          wm.Aeff = mi.Aeff; wm.L = mi.L.d(); wm.rho = mi.rho.d(); wm.H = ty == 4 ? mi.H.d() : nullptr;
          wm.fac = mi.fac.d(); wm.muD = mi.muD.d(); wm.Z = mi.Z.d(); wm.mu = mi.mu.d(); wm.Zold = mi.Zold.d();
          wm.slots = resid + (int64_t)ci.modes[j] * kResidPerMode;
          wm.R = mi.R; wm.constrained = mi.constrained ? 1 : 0; wm.ptype = mi.prox.type; wm.p0 = mi.prox.p0; wm.p1 = mi.prox.p1;
          lds += (size_t)mi.R * mi.R + (size_t)q * mi.R;
        }
        AO_REQUIRE(regs_path == (rows <= 256 && rmax <= 8 && n <= 3), "coupled loop: path prediction and launch disagree");
        if (rows <= 256 && rmax <= 8 && n <= 3) {     // one row per thread: the state stays in registers
          wa.self_start = 1;
#define AO_WGR(RM, NMM) { if (ty == 4) couple_loop_wg_regs_k<RM, NMM, true><<<1, 256, 0, stream_>>>(wa); \
                          else couple_loop_wg_regs_k<RM, NMM, false><<<1, 256, 0, stream_>>>(wa); }
          if (rmax <= 4) { if (n == 1) AO_WGR(4, 1) else if (n == 2) AO_WGR(4, 2) else AO_WGR(4, 3) }
          else { if (n == 1) AO_WGR(8, 1) else if (n == 2) AO_WGR(8, 2) else AO_WGR(8, 3) }
#undef AO_WGR
        } else {
          by_rmax([&](auto tag) { couple_loop_wg_k<decltype(tag)::value><<<1, 256, lds * sizeof(double), stream_>>>(wa); });
        }
        AO_KERNEL_CHECK();
        return;
      }
      for (int it = 0; it < opt.MaxInnerIters; ++it) {
        for (int j = 0; j < n; ++j) {                 // primal: Sd(Delta), right-hand side and row solve in one kernel
          const size_t lds = ((size_t)rc[j].R * rc[j].R + (size_t)q * rc[j].R) * sizeof(double);
          by_rmax([&](auto tag) {
            couple_primal_rows_k<decltype(tag)::value><<<rb, 64, lds, stream_>>>(rc[j], ci.Delta.d(), rows, q, ty, ctl);
          });
          AO_KERNEL_CHECK();
        }
        by_rmax([&](auto tag) {                       // Delta_old, Delta, dD
          couple_delta_rows_k<decltype(tag)::value><<<rb, 64, lds_delta * sizeof(double), stream_>>>(
              rd, ci.Delta.d(), ci.DeltaOld.d(), ci.dD.d(), ci.coef.d(), ci.LAA.d(), rows, q, ty, ctl);
        });
        AO_KERNEL_CHECK();
        FinalizeArgs fa;
        fa.nmodes = n;
        fa.max_inner = opt.MaxInnerIters;
        fa.tol_pr_coupl = opt.innerRelPrTol_coupl; fa.tol_pr_constr = opt.innerRelPrTol_constr;
        fa.tol_du_coupl = opt.innerRelDualTol_coupl; fa.tol_du_constr = opt.innerRelDualTol_constr;
        for (int j = 0; j < n; ++j) {                 // duals, constraints, residual sums
          const int m = ci.modes[j];
          ModeInfo& mi = modes_[m];
          double* sl = resid + (int64_t)m * kResidPerMode;
          by_rmax([&](auto tag) {
            couple_dual_rows_k<decltype(tag)::value><<<(unsigned)nr, 256, (size_t)q * mi.R * sizeof(double), stream_>>>(
                rc[j], ci.Delta.d(), ci.dD.d(), rows, q, ty, sl + 4, redws_.d(), ctl);
          });
          AO_KERNEL_CHECK();
          if (nr > 1) {
            couple_dual_fin_k<<<1, 64, 0, stream_>>>(sl + 4, redws_.d(), (int)nr, ctl);
            AO_KERNEL_CHECK();
          }
          if (mi.constrained)
            constraint_update(mi.prox, mi.fac.d(), mi.Z.d(), mi.mu.d(), mi.Zold.d(), mi.V.d(), mi.rows, mi.R, mi.rho.d(), 1.0,
                              mi.proxws.d(), sl, redws_.d(), ctl, stream_);
          else
            sumsq_diff(sl + 1, mi.fac.d(), nullptr, mi.rows * mi.R, redws_.d(), ctl, stream_);
          fa.slots[j] = sl;
          fa.constrained[j] = mi.constrained ? 1 : 0;
          fa.coupled[j] = 1;
        }
        admm_finalize_generic(fa, ctl, stream_);
      }
      return;
    }
  }
  for (int it = 0; it < opt.MaxInnerIters; ++it) {
    // ---- primal updates (:635-658, :713-730, :783-800, :853-870, :913-936, :1004-1020)
    for (int j = 0; j < n; ++j) {
      ModeInfo& mi = modes_[ci.modes[j]];
      const int64_t nm = mi.rows * mi.R, ni = mi.img_rows * mi.img_cols;
      // Sd(Delta): after the first inner iteration the image computed in the dual step below is still current
      const double* TD = (it == 0 || ty == 0 || ty == 1 || ty == 2) ? image_d(mi.TD.d(), ci, ci.Delta.d(), mi, ctl, stream_)
                                                                     : mi.TD.d();
      Par2Block* pb = pc_block(mi);
      if (pb && ty != 1 && ty != 5) {
        // row k: A_inner = a_k + rho_k/2*Tf'(Sd(Delta) - mu_Delta)(k,:) [+ rho_k/2*(Z - mu)(k,:)], solved with L_k
        // (:638-645, :785-792, :850-857, :916-923); Tf' is the identity except for type 2 (right-multiplication by H')
        if (ty == 2) {
          Coef c2[2] = {coef(1.0), coef(-1.0)};
          const double* x2[2] = {TD, mi.muD.d()};
          ew_lincomb(mi.tmp.d(), ni, 2, c2, x2, ctl, stream_);
          const double* adj = adjoint_f(mi.TF.d(), ci, mi.tmp.d(), mi, ctl, stream_);
          Coef cf[3] = {coef(1.0), coef(1.0), coef(-1.0)};
          const double* x[3] = {adj, mi.Z.d(), mi.mu.d()};
          ew_lincomb(mi.RHS.d(), nm, mi.constrained ? 3 : 1, cf, x, ctl, stream_);
        } else {
          Coef cf[4] = {coef(1.0), coef(-1.0), coef(1.0), coef(-1.0)};
          const double* x[4] = {TD, mi.muD.d(), mi.Z.d(), mi.mu.d()};
          ew_lincomb(mi.RHS.d(), nm, mi.constrained ? 4 : 2, cf, x, ctl, stream_);
        }
        par2_c_rowsolve(pb->ac.d(), pb->rhoc.d(), pb->Lc.d(), mi.RHS.d(), nullptr, 1, pb->dims_all(), mi.fac.d(), ctl, stream_);
        continue;
      }
      if (ty == 0 || ty == 3 || ty == 4) {
        Coef cf[5] = {coef(1.0), coef(mi.rho.d(), 0.5), coef(mi.rho.d(), -0.5), coef(mi.rho.d(), 0.5), coef(mi.rho.d(), -0.5)};
        const double* x[5] = {mi.Aeff, TD, mi.muD.d(), mi.Z.d(), mi.mu.d()};
        ew_lincomb(mi.RHS.d(), nm, mi.constrained ? 5 : 3, cf, x, ctl, stream_);
      } else {
        Coef c2[2] = {coef(1.0), coef(-1.0)};
        const double* x2[2] = {TD, mi.muD.d()};
        ew_lincomb(mi.tmp.d(), ni, 2, c2, x2, ctl, stream_);                   // Sd(Delta) - mu_Delta
        const double* adj = adjoint_f(mi.TF.d(), ci, mi.tmp.d(), mi, ctl, stream_);
        Coef cf[4] = {coef(1.0), coef(mi.rho.d(), 0.5), coef(mi.rho.d(), 0.5), coef(mi.rho.d(), -0.5)};
        const double* x[4] = {mi.Aeff, adj, mi.Z.d(), mi.mu.d()};
        ew_lincomb(mi.RHS.d(), nm, mi.constrained ? 4 : 2, cf, x, ctl, stream_);
      }
      if (pb && (ty == 1 || ty == 5)) {
        // vec(C') = (blkdiag(B_k) + rhoC/2*kron(H'H,I) [+ rhoC/2*I]) \ vec(A_inner') (:714-722); mi.rho holds rhoC
        if (pb->hth_diag)                             // H'H diagonal: the system is K row systems (solver_par2.hip)
          par2_c_rowsolve(mi.RHS.d(), pb->rhoc.d(), pb->Lc.d(), nullptr, nullptr, 0, pb->dims_all(), mi.fac.d(), ctl, stream_);
        else
          dense_symv_rows(pb->Minv.d(), mi.RHS.d(), mi.fac.d(), pb->K, pb->R, ctl, stream_);
      } else if (ty == 1 || ty == 5) {
        // sylvester(B2, B, A_inner) (:707, :1016) with B2 = rho/2*H'H (+ rho/2*I if constrained) = U (..) U',
        // B = V diag(mu) V':  X = U * ((U' A_inner V) ./ (beta_i + mu_j)) * V'
        gemm_small(mi.W1.d(), mi.rows, mi.eUt.d(), mi.rows, mi.RHS.d(), mi.rows, mi.rows, (int)mi.rows, mi.R, 0, coef(1.0), 0.0, ctl, stream_);
        gemm_small(mi.W2.d(), mi.rows, mi.W1.d(), mi.rows, mi.eV.d(), mi.R, mi.rows, mi.R, mi.R, 0, coef(1.0), 0.0, ctl, stream_);
        sylv_scale(mi.W2.d(), mi.rows, mi.R, mi.eLam.d(), mi.eMu.d(), mi.rho.d(), 1.0, mi.constrained ? 1.0 : 0.0, ctl, stream_);
        gemm_small(mi.W1.d(), mi.rows, mi.W2.d(), mi.rows, mi.eV.d(), mi.R, mi.rows, mi.R, mi.R, 1, coef(1.0), 0.0, ctl, stream_);
        gemm_small(mi.fac.d(), mi.rows, mi.eU.d(), mi.rows, mi.W1.d(), mi.rows, mi.rows, (int)mi.rows, mi.R, 0, coef(1.0), 0.0, ctl, stream_);
      } else {
        row_solve(mi.fac.d(), mi.rows, mi.RHS.d(), mi.rows, mi.L.d(), mi.rows, mi.R, ctl, stream_);
      }
    }
    // ---- Delta update
    {
      Coef c1[1] = {coef(1.0)};
      const double* x1[1] = {ci.Delta.d()};
      ew_lincomb(ci.DeltaOld.d(), nD, 1, c1, x1, ctl, stream_);
    }
    if ((ty == 0 || ty == 2) && any_pc) {             // per-row weights rho_j(k) (:666-675, :805-811)
      RowMeanArgs ra;
      ra.n = n; ra.rows = ci.rows; ra.cols = (int)ci.cols;
      for (int j = 0; j < n; ++j) {
        ModeInfo& mi = modes_[ci.modes[j]];
        Par2Block* pb = pc_block(mi);
        ra.fac[j] = image_f(mi.TF.d(), ci, mi.fac.d(), mi, ctl, stream_); ra.mu[j] = mi.muD.d();
        ra.rho[j] = pb ? pb->rhoc.d() : mi.rho.d();
        ra.vec[j] = pb ? 1 : 0;
      }
      int64_t nb = cdiv(nD, 256);
      if (nb > 1024) nb = 1024;
      coupling_rowmean_k<<<(unsigned)nb, 256, 0, stream_>>>(ci.Delta.d(), ra, ctl);
      AO_KERNEL_CHECK();
    } else if (ty == 0 || ty == 1 || ty == 2) {       // weighted mean of Tf(C_j) + mu_j (:661-675, :735-741, :805-811)
      for (int j = 0; j < n; ++j) {
        ModeInfo& mi = modes_[ci.modes[j]];
        const double* tf = image_f(mi.TF.d(), ci, mi.fac.d(), mi, ctl, stream_);
        if (j == 0) {
          Coef cf[2] = {coef(ci.coef.d() + j, 1.0), coef(ci.coef.d() + j, 1.0)};
          const double* x[2] = {tf, mi.muD.d()};
          ew_lincomb(ci.Delta.d(), nD, 2, cf, x, ctl, stream_);
        } else {
          Coef cf[3] = {coef(1.0), coef(ci.coef.d() + j, 1.0), coef(ci.coef.d() + j, 1.0)};
          const double* x[3] = {ci.Delta.d(), tf, mi.muD.d()};
          ew_lincomb(ci.Delta.d(), nD, 3, cf, x, ctl, stream_);
        }
      }
    } else if (ty == 3) {                             // Delta = AA \ BB (:875-885)
      for (int j = 0; j < n; ++j) {
        ModeInfo& mi = modes_[ci.modes[j]];
        Coef cf[2] = {coef(1.0), coef(1.0)};
        const double* x[2] = {mi.fac.d(), mi.muD.d()};
        ew_lincomb(mi.tmp.d(), mi.rows * mi.R, 2, cf, x, ctl, stream_);
        if (Par2Block* pb = pc_block(mi)) {           // rows weighted by rho_k: H'*diag(rho)*H and H'*diag(rho)*(C + mu)
          pb->Hs.ensure((size_t)mi.hr * mi.hc * 8);
          rows_scale(pb->Hs.d(), mi.H.d(), pb->rhoc.d(), mi.hr, mi.hc, ctl, stream_);
          rows_scale(mi.tmp.d(), mi.tmp.d(), pb->rhoc.d(), mi.rows, mi.R, ctl, stream_);
          gemm_small(ci.AA.d(), ci.rows, mi.Ht.d(), mi.hc, pb->Hs.d(), mi.hr, ci.rows, (int)mi.rows, (int)ci.rows, 0,
                     coef(1.0), j == 0 ? 0.0 : 1.0, ctl, stream_);
          gemm_small(ci.BB.d(), ci.rows, mi.Ht.d(), mi.hc, mi.tmp.d(), mi.rows, ci.rows, (int)mi.rows, mi.R, 0,
                     coef(1.0), j == 0 ? 0.0 : 1.0, ctl, stream_);
          continue;
        }
        gemm_small(ci.AA.d(), ci.rows, mi.Ht.d(), mi.hc, mi.H.d(), mi.hr, ci.rows, (int)mi.rows, (int)ci.rows, 0,
                   coef(mi.rho.d(), 1.0), j == 0 ? 0.0 : 1.0, ctl, stream_);
        gemm_small(ci.BB.d(), ci.rows, mi.Ht.d(), mi.hc, mi.tmp.d(), mi.rows, ci.rows, (int)mi.rows, mi.R, 0,
                   coef(mi.rho.d(), 1.0), j == 0 ? 0.0 : 1.0, ctl, stream_);
      }
      spd_solve_left(ci.AA.d(), ci.rows, ci.BB.d(), (int)ci.cols, ctl, stream_);
      Coef c1[1] = {coef(1.0)};
      const double* x1[1] = {ci.BB.d()};
      ew_lincomb(ci.Delta.d(), nD, 1, c1, x1, ctl, stream_);
    } else {                                          // types 4, 5: Delta = BB / AA (:939-963, :1026-1054)
      for (int j = 0; j < n; ++j) {
        ModeInfo& mi = modes_[ci.modes[j]];
        const double* tf = image_f(mi.TF.d(), ci, mi.fac.d(), mi, ctl, stream_);
        Coef cf[2] = {coef(1.0), coef(1.0)};
        const double* x[2] = {tf, mi.muD.d()};
        ew_lincomb(mi.tmp.d(), mi.img_rows * mi.img_cols, 2, cf, x, ctl, stream_);
        // BB += rho_j * (Tf(C_j) + mu_j) * H_j'   (:955 ; :1048 with H2 and rhoC)
        if (ty == 4 && pc_block(mi)) {                // rows weighted by rho_k (:955)
          rows_scale(mi.tmp.d(), mi.tmp.d(), pc_block(mi)->rhoc.d(), mi.rows, mi.R, ctl, stream_);
          gemm_small(ci.BB.d(), ci.rows, mi.tmp.d(), mi.rows, mi.H.d(), mi.hr, ci.rows, mi.R, (int)ci.cols, 1,
                     coef(1.0), j == 0 ? 0.0 : 1.0, ctl, stream_);
        } else if (ty == 4)
          gemm_small(ci.BB.d(), ci.rows, mi.tmp.d(), mi.rows, mi.H.d(), mi.hr, ci.rows, mi.R, (int)ci.cols, 1,
                     coef(mi.rho.d(), 1.0), j == 0 ? 0.0 : 1.0, ctl, stream_);
        else
          gemm_small(ci.BB.d(), ci.rows, mi.tmp.d(), mi.img_rows, mi.H2.d(), mi.h2r, ci.rows, mi.R, (int)ci.cols, 1,
                     coef(rho_last, 1.0), j == 0 ? 0.0 : 1.0, ctl, stream_);
      }
      if (any_pc) {                                   // Delta(k,:) = BB(k,:) / (AA + rho_k*AAA)  (:957-961, :1049-1052)
        const Par2Block* pb = nullptr;
        for (int j = 0; j < n; ++j)
          if (Par2Block* q = pc_block(modes_[ci.modes[j]])) pb = q;
        const int q = (int)ci.cols;
        delta_rowwise_solve_k<<<(unsigned)ci.rows, 64, (size_t)(q * q + q) * sizeof(double), stream_>>>(
            ci.Delta.d(), ci.BB.d(), ci.rows, q, ci.AA.d(), ci.LAA.d(), pb->rhoc.d(), ctl);
        AO_KERNEL_CHECK();
      } else
      row_solve(ci.Delta.d(), ci.rows, ci.BB.d(), ci.rows, ci.LAA.d(), ci.rows, (int)ci.cols, ctl, stream_);
    }
    {
      Coef cf[2] = {coef(1.0), coef(-1.0)};
      const double* x[2] = {ci.Delta.d(), ci.DeltaOld.d()};
      ew_lincomb(ci.dD.d(), nD, 2, cf, x, ctl, stream_);
    }
    // ---- duals, constraints, residual pieces (:678-692 and the same block of every case)
    FinalizeArgs fa;
    fa.nmodes = n;
    fa.max_inner = opt.MaxInnerIters;
    fa.tol_pr_coupl = opt.innerRelPrTol_coupl; fa.tol_pr_constr = opt.innerRelPrTol_constr;
    fa.tol_du_coupl = opt.innerRelDualTol_coupl; fa.tol_du_constr = opt.innerRelDualTol_constr;
    for (int j = 0; j < n; ++j) {
      const int m = ci.modes[j];
      ModeInfo& mi = modes_[m];
      double* sl = resid + (int64_t)m * kResidPerMode;
      const int64_t nm = mi.rows * mi.R, ni = mi.img_rows * mi.img_cols;
      const double* TD = image_d(mi.TD.d(), ci, ci.Delta.d(), mi, ctl, stream_);
      const double* tf = image_f(mi.TF.d(), ci, mi.fac.d(), mi, ctl, stream_);
      {
        // mu_Delta += Tf(C) - Sd(Delta); sl[4] = ||Tf(C) - Sd(Delta)||^2, sl[5] = ||mu_Delta||^2, sl[7] = ||den||^2 with
        // den = H*C (:1125) / C*H (:1143) for types 1, 2, else C
        const bool img_den = ty == 1 || ty == 2;
        int64_t nr = cdiv(std::max(ni, nm), 2048);
        if (nr > 64) nr = 64;
        coupling_dual_k<<<(unsigned)nr, 256, 0, stream_>>>(mi.muD.d(), tf, TD, ni, img_den ? tf : mi.fac.d(), img_den ? ni : nm,
                                                           sl + 4, redws_.d(), ctl);
        AO_KERNEL_CHECK();
        if (nr > 1) {
          coupling_dual_fin_k<<<1, 64, 0, stream_>>>(sl + 4, redws_.d(), (int)nr, ctl);
          AO_KERNEL_CHECK();
        }
      }
      if (mi.constrained) {
        Par2Block* pb = pc_block(mi);                 // a C mode's prox gets max(rho) (:1423-1424)
        constraint_update(mi.prox, mi.fac.d(), mi.Z.d(), mi.mu.d(), mi.Zold.d(), mi.V.d(), mi.rows, mi.R,
                          pb ? pb->rhomax.d() : mi.rho.d(), 1.0, mi.proxws.d(), sl, redws_.d(), ctl, stream_);
      } else
        sumsq_diff(sl + 1, mi.fac.d(), nullptr, nm, redws_.d(), ctl, stream_);
      const double* dimg = image_d(mi.tmp.d(), ci, ci.dD.d(), mi, ctl, stream_);
      sumsq_diff(sl + 6, dimg, nullptr, ni, redws_.d(), ctl, stream_);
      fa.slots[j] = sl;
      fa.constrained[j] = mi.constrained ? 1 : 0;
      fa.coupled[j] = 1;
    }
    admm_finalize_generic(fa, ctl, stream_);
  }
}

// ---------------------------------------------------------------------------
// objective (CMTF_AOADMM_func_eval, :1213-1363)
// ---------------------------------------------------------------------------
void Engine::eval_objective_enqueue(bool first) {
  double* S = slots_.d();
  ReduceBatch rb;                                  // every plain reduction of this evaluation in one launch
  auto add = [&](int kind, double* slot, const double* x, const double* y, int64_t n) {
    ReduceTask k;
    k.kind = kind; k.slot = slot; k.x = x; k.y = y; k.n = n;
    rb.add(k);
  };
  for (int p = 0; p < n_tensors_; ++p) {
    TensorInfo& t = tensors_[p];
    const bool masked = t.par2 ? t.p2.has_mask : t.blk.has_mask;
    if (masked && first) em_pass_enqueue(p, 0);  // observed-entry residual (:1224-1226, :1249-1252); later
                                                 // evaluations reuse the statistics of the EM update pass
    if (t.par2) {
      par2_objective_enqueue(t);                 // direct residual (:1262-1264) + internal-coupling gaps (:1355)
      t.eval_shortcut = !masked && !first && t.last_pos == 0;   // remembered for finish_eval: last_pos may move on before
      if (t.eval_shortcut) {                                // shortcut through last_mttkrp / last_had (:1254-1260)
        ModeInfo& lm = modes_[t.modes[0]];
        double* sp = S + n_modes_ * kSlotsPerMode + 2 * p;
        add(RT_DOT, sp + 0, lm.A.d(), lm.fac.d(), lm.rows * lm.R);
        add(RT_DOT, sp + 1, lm.C.d(), lm.gram.d(), (int64_t)lm.R * lm.R);
      }
      continue;
    }
    if (masked) continue;
    if (first) {
      // cp_func.m:47-55 / pca_func.m:29-39: same formula with the first mode's MTTKRP
      ModeInfo& m0 = modes_[t.modes[0]];
      FactorRef facs[8];
      for (int i = 0; i < t.nmodes; ++i) {
        const ModeInfo& o = modes_[t.modes[i]];
        facs[i] = factor_ref(o);
      }
      std::vector<int> seq = update_sequence(p);
      block_mttkrp(t.blk, 0, facs, m0.R, t.weight, m0.A.d(), m0.rows, true, seq.data(), (int)seq.size());
      SysBuild sb;
      sb.ngram = 0;
      for (int i = 1; i < t.nmodes; ++i) sb.grams[sb.ngram++] = modes_[t.modes[i]].gram.d();
      sb.Cpre = nullptr; sb.w = t.weight; sb.ridge = 0; sb.bsum_half = 0; sb.rho_scale = 1; sb.nrho = 1; sb.R = m0.R;
      sb.C = m0.C.d(); sb.rho = m0.rho.d(); sb.Bsys = m0.Bsys.d(); sb.L = m0.L.d(); sb.ctl = nullptr;
      sys_build(sb, stream_);
      t.last_pos = 0;
    }
    ModeInfo& lm = modes_[t.modes[t.last_pos]];
    double* sp = S + n_modes_ * kSlotsPerMode + 2 * p;
    add(RT_DOT, sp + 0, lm.A.d(), lm.fac.d(), lm.rows * lm.R);                 // f_2 * w
    add(RT_DOT, sp + 1, lm.C.d(), lm.gram.d(), (int64_t)lm.R * lm.R);          // f_3
  }
  for (int m = 0; m < n_modes_; ++m) {
    ModeInfo& mi = modes_[m];
    if (mi.slabs) continue;                      // per-slab ratios come from par2_b_gaps
    double* sm = S + (int64_t)m * kSlotsPerMode;
    const int64_t nm = mi.rows * mi.R;
    add(RT_SUMSQ_DIFF, sm + 0, mi.fac.d(), nullptr, nm);
    if (mi.constrained) {
      add(RT_SUMSQ_DIFF, sm + 1, mi.fac.d(), mi.Z.d(), nm);
      const int ty = mi.prox.type;
      if (ty == AOADMM_C_L2_REG) {
        reg_value(sm + 3, ty, mi.prox.p0, mi.fac.d(), mi.rows, mi.R, redws_.d(), stream_);
      } else if (ty == AOADMM_C_QUADRATIC) {       // eta*trace(x'*L*x) (:67): L*x into the prox workspace, then <x, L*x>
        gemm_small(mi.proxws.d(), mi.rows, mi.prox.Lmat, mi.rows, mi.fac.d(), mi.rows, mi.rows, (int)mi.rows, mi.R, 0,
                   coef(1.0), 0.0, nullptr, stream_);
        ReduceTask k;
        k.kind = RT_DOT; k.slot = sm + 3; k.x = mi.fac.d(); k.y = mi.proxws.d(); k.n = nm; k.scale = mi.prox.p0;
        rb.add(k);
      } else if (ty == AOADMM_C_L1_REG || ty == AOADMM_C_L0_REG || ty == AOADMM_C_RIDGE || ty == AOADMM_C_GL_SMOOTH ||
                 ty == AOADMM_C_TV) {
        ReduceTask k;
        k.kind = RT_REG; k.aux = ty; k.slot = sm + 3; k.x = mi.fac.d(); k.rows = mi.rows; k.R = mi.R; k.scale = mi.prox.p0;
        rb.add(k);
      }
    }
    if (mi.coupling >= 0) {                        // :1303-1329
      CouplingInfo& ci = couplings_[mi.coupling];
      const size_t nimg = (size_t)std::max(nm, mi.img_rows * mi.img_cols) * sizeof(double);
      mi.TD.ensure(nimg); mi.TF.ensure(nimg);
      const double* td = image_d(mi.TD.d(), ci, ci.Delta.d(), mi, nullptr, stream_);
      const double* tf = image_f(mi.TF.d(), ci, mi.fac.d(), mi, nullptr, stream_);
      add(RT_SUMSQ_DIFF, sm + 2, tf, td, mi.img_rows * mi.img_cols);
      if (tf != mi.fac.d()) add(RT_SUMSQ_DIFF, sm + 4, tf, nullptr, mi.img_rows * mi.img_cols);   // ||H*C|| / ||C*H||
    }
    if (rb.n >= kReduceBatchMax - 4) {           // many modes: flush and start the next batch
      reduce_batch(rb, redws_.d(), stream_);
      rb = ReduceBatch();
    }
  }
  reduce_batch(rb, redws_.d(), stream_);
}

static bool stop_one(double f, double fo, const aoadmm_options& o) {
  const double rel = fo > 0 ? std::fabs(fo - f) / fo : std::fabs(fo - f);    // evaluate_stopping_conditions.m:8-15
  return f < o.AbsFuncTol || rel < o.OuterRelTol;
}

void Engine::solve(const aoadmm_options& opt, aoadmm_result* out) {
  require_usable();
  AO_REQUIRE(model_done_, "call aoadmm_model_end first");
  AO_REQUIRE(out != nullptr, "null result");
  AO_REQUIRE(opt.MaxOuterIters >= 0 && opt.MaxInnerIters >= 1, "bad iteration limits");
  AO_HIP(hipSetDevice(device_));
  prepared_mode_ = -1;                                // nothing prepared ahead by an earlier solve is valid for this state
  allow_xp_ = opt.no_permuted_copy == 0;
  if (!allow_xp_)
    for (int p = 0; p < n_tensors_; ++p)
      drop_permuted_copies(tensors_[p].blk);
  for (int p = 0; p < n_tensors_; ++p) {
    AO_REQUIRE(tensors_[p].blk.has_data, "tensor %d has no data (Z.object{%d})", p, p + 1);
  }
  for (int m = 0; m < n_modes_; ++m) {
    ModeInfo& mi = modes_[m];
    AO_REQUIRE(mi.has_fac, "G.fac{%d} missing", m + 1);
    if (mi.constrained) AO_REQUIRE(mi.has_Z && mi.has_mu, "G.constraint_fac{%d} / constraint_dual_fac{%d} missing", m + 1, m + 1);
    if (mi.coupling >= 0) {
      AO_REQUIRE(mi.has_muD && mi.muD_rows == mi.img_rows && mi.muD_cols == mi.img_cols, "G.coupling_dual_fac{%d} missing or mis-sized", m + 1);
      AO_REQUIRE(couplings_[mi.coupling].has_state, "G.coupling_fac{%d} missing", mi.coupling + 1);
    }
    ensure_mode_work(mi);
    if (!mi.slabs) compute_gram(mi);                                         // :62-81
  }
  for (int p = 0; p < n_tensors_; ++p) {
    TensorInfo& t = tensors_[p];
    if (t.par2) {
      for (int k = 0; k < t.p2.K; ++k)
        AO_REQUIRE(t.p2.have_P[k] && t.p2.have_mu[k], "G.P{%d}{%d} / G.mu_DeltaB{%d}{%d} missing", p + 1, k + 1, p + 1, k + 1);
    }
    (void)tensor_normsq(p);          // Znorm_const{p}; a masked block needs the factors (statistics-only EM pass)
  }
  const bool has_miss = has_missing();
  for (int p = 0; p < n_tensors_; ++p) {
    TensorInfo& t = tensors_[p];
    if (!t.par2) continue;
    Par2Block& b = t.p2;
    AO_REQUIRE(b.has_DeltaB, "G.DeltaB{%d} missing", p + 1);
    for (int k = 0; k < b.K; ++k) AO_REQUIRE(b.have_P[k] && b.have_mu[k], "G.P{%d}{%d} / G.mu_DeltaB{%d}{%d} missing", p + 1, k + 1, p + 1, k + 1);
    par2_ensure_work(t);
    {
      // slabs over the ranks or every slab on every rank (aoadmm_options.par2_slab_sharding)
      const ModeInfo& mB = modes_[t.modes[1]];
      const bool can = sharded() && world_ > 1 && !b.has_mask && !(mB.constrained && mB.prox.type == AOADMM_C_TPARAFAC2) &&
                       modes_[t.modes[2]].coupling < 0;     // a coupled C mode needs every row system on every rank
      const bool want = opt.par2_slab_sharding > 0 || (opt.par2_slab_sharding == 0 && b.K / world_ >= 1024);
      const int per = (int)cdiv(b.K, world_);
      // every rank must own a slab, and every rank must reach the same verdict: otherwise repeat the block
      b.slab_sharded = can && want && (int64_t)per * (world_ - 1) < b.K;
      b.k0 = std::min(b.K, per * rank_);
      b.k1 = std::min(b.K, b.k0 + per);
    }
    par2_gram(modes_[t.modes[1]].fac.d(), b.dims(), b.GB.d(), stream_);      // :71-73
    t.last_pos = 2;
  }
  const int nctl = n_modes_ + n_couplings_;
  const int nslots = n_modes_ * kSlotsPerMode + 2 * n_tensors_;
  // pinned landing area + event: the host waits for the objective values only, not for work enqueued
  // behind them (prefetch_next_contraction)
  struct Pinned {
    void* p = nullptr; hipEvent_t ev = nullptr;
    ~Pinned() { if (p) (void)hipHostFree(p); if (ev) (void)hipEventDestroy(ev); }
  } pin;
  // per PARAFAC2 block: K + 1 slab residuals (+ the not-PD flag of sharded slabs), 4 K gap sums, K regulariser values --
  // read back with everything else behind ONE event (three more copies into pageable memory with a stream
  // synchronisation each left the GPU idle for ~60 us per outer iteration of config 4)
  const char* rb_base = readback_.as<char>();
  auto in_arena = [&](const DevBuf& d) {
    return !d.owned && static_cast<const char*>(d.p) >= rb_base && static_cast<const char*>(d.p) + d.bytes <= rb_base + readback_.bytes;
  };
  AO_REQUIRE(in_arena(slots_) && in_arena(ctls_), "read-back arena: slots / loop-control blocks are not views of it");
  AO_HIP(hipHostMalloc(&pin.p, readback_.bytes, hipHostMallocDefault));
  AO_HIP(hipEventCreateWithFlags(&pin.ev, hipEventDisableTiming));
  char* hb = static_cast<char*>(pin.p);
  double* hs = reinterpret_cast<double*>(hb + (slots_.as<char>() - rb_base));
  double* hem = hs + (em_slot(0) - slots_.d());                                // EM statistics, 4 per tensor
  AdmmCtl* hctl = reinterpret_cast<AdmmCtl*>(hb + (ctls_.as<char>() - rb_base));
  std::vector<double*> hp2v(n_tensors_, nullptr);                              // PARAFAC2 per-slab values: res | q | regv
  for (int p = 0; p < n_tensors_; ++p) {
    if (!tensors_[p].par2) continue;
    Par2Block& b = tensors_[p].p2;
    AO_REQUIRE(in_arena(b.res) && in_arena(b.q) && in_arena(b.regv) && b.q.d() == b.res.d() + b.K + 1 &&
               b.regv.d() == b.res.d() + 5 * b.K + 1, "read-back arena: PARAFAC2 block %d keeps its sums elsewhere", p);
    hp2v[p] = reinterpret_cast<double*>(hb + (b.res.as<char>() - rb_base));
  }
  (void)nslots;

  auto enqueue_readback = [&]() {
    AO_HIP(hipMemcpyAsync(hb, readback_.p, readback_.bytes, hipMemcpyDeviceToHost, stream_));
    AO_HIP(hipEventRecord(pin.ev, stream_));
  };
  auto finish_eval = [&](double f[4]) {
    AO_HIP(hipEventSynchronize(pin.ev));
    for (int i = 0; i < nctl; ++i)
      if (hctl[i].notpd)
        throw Error(AOADMM_ERR_NOT_PD, "Cholesky failed: system matrix is not positive definite (chol in cmtf_fun_AOADMM.m:142/273/362)");
    double ft = 0.0, fpar = 0.0, fcon = 0.0;
    int ncon = 0;
    for (int p = 0; p < n_tensors_; ++p) {
      TensorInfo& t = tensors_[p];
      const double* sp = hs + n_modes_ * kSlotsPerMode + 2 * p;
      const bool masked = t.par2 ? t.p2.has_mask : t.blk.has_mask;
      if (t.par2) {
        Par2Block& b = t.p2;
        const double* res = hp2v[p];
        const double* q = res + b.K + 1;
        if (b.slab_sharded && res[b.K] > 0)         // some rank's slabs hit a non-positive-definite system (the slot is only written then)
          throw Error(AOADMM_ERR_NOT_PD, "Cholesky failed in a PARAFAC2 slab system on another rank (chol in cmtf_fun_AOADMM.m:212/240)");
        double fp = 0.0;
        if (masked) fp = hem[4 * p + 2];                                                        // :1249-1252
        else if (t.eval_shortcut) fp = t.normsq - 2.0 * (sp[0] / t.weight) + sp[1];               // :1254-1260
        else for (int k = 0; k < b.K; ++k) fp += res[k];                                        // :1262-1264
        ft += t.weight * fp;                                                                    // :1267
        const ModeInfo& mB = modes_[t.modes[1]];
        double gp = 0.0, gz = 0.0, nb2 = 0.0;
        for (int k = 0; k < b.K; ++k) {
          const double nb = std::sqrt(q[4 * k + 1]);
          gp += std::sqrt(q[4 * k]) / nb;                                                       // :1355
          gz += std::sqrt(q[4 * k + 2]) / nb;                                                   // :1337
          nb2 += q[4 * k + 1];
        }
        fpar += gp;
        if (mB.constrained && mB.prox.type == AOADMM_C_TPARAFAC2) {        // t_smoothness_penalty.m via reg_func (:1276-1277)
          double pen = 0.0;
          for (int k = 1; k < b.K; ++k) pen += q[4 * k + 3];
          ft += mB.prox.p0 * pen;
        }
        if (mB.constrained && prox_has_reg_value(mB.prox.type)) {          // sum_k reg_func(B_k) (:1279-1281)
          const double* rv = res + 5 * b.K + 1;
          for (int k = 0; k < b.K; ++k) ft += rv[k];
        }
        if (mB.constrained) {
          const double g = gz / b.K;                                                            // :1339
          fcon += g;
          if (g != 0.0) ++ncon;
          if (has_ridge_) ft += mB.ridge * nb2;                                                 // :1292-1295 (quirk: only if constrained)
        }
      } else if (masked) {
        ft += t.weight * hem[4 * p + 2];                                       // :1224-1226 = w * ||miss.*(X - M)||^2
      } else {
        const double f2 = sp[0] / t.weight;                                   // last_mttkrp = A*1/w (:121)
        ft += t.weight * (t.normsq - 2.0 * f2 + sp[1]);                        // :1235-1241
      }
    }
    if (fpar > 0) {                                                            // :1360-1362 (quirk: K of the LAST tensor)
      const TensorInfo& tl = tensors_[n_tensors_ - 1];
      fpar /= tl.par2 ? tl.p2.K : 1;
    }
    std::vector<double> cp(n_couplings_, 0.0);
    for (int m = 0; m < n_modes_; ++m) {
      const ModeInfo& mi = modes_[m];
      if (mi.slabs) continue;
      const double* sm = hs + (int64_t)m * kSlotsPerMode;
      const double nf = std::sqrt(sm[0]);
      if (mi.constrained) {
        const int ty = mi.prox.type;
        if (prox_has_reg_value(ty)) ft += sm[3];                               // reg_func (:1272-1288)
        const double g = std::sqrt(sm[1]) / nf;                               // :1341
        fcon += g;
        if (g != 0.0) ++ncon;
      }
      if (has_ridge_) ft += mi.ridge * sm[0];                                  // :1297
      if (mi.coupling >= 0) {                                                  // :1309-1323
        const int cty = couplings_[mi.coupling].type;
        const double den = (cty == 1 || cty == 2 || cty == 5) ? std::sqrt(sm[4]) : nf;
        cp[mi.coupling] += std::sqrt(sm[2]) / den;
      }
    }
    double fc = 0.0; int nc = 0;
    for (double v : cp) { fc += v; if (v != 0.0) ++nc; }
    if (fc > 0) fc /= nc;                                                      // :1327-1329
    if (fcon > 0) fcon /= ncon;                                                // :1346-1348
    f[0] = ft; f[1] = fc; f[2] = fcon; f[3] = fpar;
  };

  double f[4], fo[4];
  eval_objective_enqueue(true);                                                // :32
  enqueue_readback();
  finish_eval(f);
  if (out->func_val_conv) out->func_val_conv[0] = f[0];
  if (out->func_coupl_conv) out->func_coupl_conv[0] = f[1];
  if (out->func_constr_conv) out->func_constr_conv[0] = f[2];
  if (out->func_PAR2_coupl) out->func_PAR2_coupl[0] = f[3];
  if (out->time_at_it) out->time_at_it[0] = 0.0;
  double f_rel_missing = std::nan("");                                          // :30
  if (out->func_rel_missing) out->func_rel_missing[0] = f_rel_missing;
  const bool report = progress_fn_ != nullptr && progress_every_ > 0;
  if (report) progress_fn_(progress_user_, 0, f, f_rel_missing);               // :53-59
  const auto t0 = std::chrono::steady_clock::now();

  int iter = 1;
  bool stop = false;
  while (iter <= opt.MaxOuterIters && !stop) {                                 // :87
    for (ModeInfo& mq : modes_) mq.quad.dirty = true;   // rho moves once per outer iteration ('quadratic regularization', non-symmetric L)
    if (iter == 3)                                      // by now every pass of the schedule has run once: all copies exist
      for (int p = 0; p < n_tensors_; ++p)
        if (!tensors_[p].par2) maybe_release_natural(tensors_[p]);
    for (int cid = -1; cid < n_couplings_; ++cid) {                            // :89 (0 = uncoupled first)
      std::vector<int> cm;
      for (int m = 0; m < n_modes_; ++m)
        if (modes_[m].coupling == cid) cm.push_back(m);
      if (cm.empty()) continue;
      std::set<int> ps;
      for (int m : cm) ps.insert(modes_[m].tensor);
      for (int p : ps)                                                         // :91
        for (int m : cm)                                                       // :93
          if (modes_[m].tensor == p) {
            const bool par2 = tensors_[p].par2;
            if (par2 && modes_[m].pos == 1) par2_update_B(m, opt, iter);             // :191-218
            else if (par2 && modes_[m].pos == 2 && cid < 0) par2_update_C(m, opt);   // :219-248
            else if (par2 && modes_[m].pos == 2) par2_prepare_C_coupled(m, couplings_[cid].type, opt);
            else if (cid < 0) update_uncoupled_cp_mode(m, opt);
            else {
              // system of a coupled mode: +rho/2*I (types 0, 3, 4: :269, :336, :358), +rho/2*H*H' (type 2, :314),
              // nothing for the Sylvester types 1, 5 (:288-293, :377-382); +rho/2*I more if constrained
              const int cty = couplings_[cid].type;
              const int con = modes_[m].constrained ? 1 : 0;
              prepare_mode_system(m, (cty == 0 || cty == 3 || cty == 4) ? 1 + con : (cty == 2 ? con : 0), opt);
            }
          }
      if (cid >= 0) {
        coupled_admm(cid, opt);                                                // :277 / :366
        for (int m : cm) { modes_[m].version++; compute_gram(modes_[m]); }      // :393-403
      }
    }
    if (has_miss)                                                              // EM imputation (:408-441)
      for (int p = 0; p < n_tensors_; ++p)
        if (tensors_[p].par2 ? tensors_[p].p2.has_mask : tensors_[p].blk.has_mask)
          em_pass_enqueue(p, 1, opt.use_dimtree != 0 && iter < opt.MaxOuterIters);
    for (int i = 0; i < 4; ++i) fo[i] = f[i];
    if (iter < opt.MaxOuterIters) {
      // The objective needs nothing the first tensor pass of the next iteration writes (frag, T), and that pass does
      // not depend on the stopping decision: the pass goes onto the main stream, the objective kernels and their
      // read-back onto the side stream behind an event, and the main stream takes up its small kernels again only
      // when the objective is through (they overwrite what it reads).  ~50 us per iteration off the critical path.
      // (Only when a pass is actually launched: the cross-stream wait alone costs ~40 us.)
      AO_HIP(hipEventRecord(side_ev_, stream_));
      if (prefetch_next_contraction(opt)) {
        AO_HIP(hipStreamWaitEvent(side_, side_ev_, 0));
        std::swap(stream_, side_);
        try {
          eval_objective_enqueue(false);                                       // :447
          enqueue_readback();
        } catch (...) { std::swap(stream_, side_); throw; }
        std::swap(stream_, side_);
        AO_HIP(hipStreamWaitEvent(stream_, pin.ev, 0));
      } else {
        eval_objective_enqueue(false);                                         // :447
        enqueue_readback();
        // No pass to hide behind: the host now waits ~45 us for the read-back before it can enqueue anything, and the
        // GPU would sit idle.  The MTTKRP (reductions over the cached T) and the system build of the next iteration's
        // first mode depend on no stopping decision and write only that mode's scratch (A, C, rho, B, L, inv, ctl --
        // behind the read-back of this iteration's loop counters in stream order): enqueue them now.
        if (!has_miss) prepare_next_first_mode(opt);
      }
    } else {
      eval_objective_enqueue(false);                                           // :447
      enqueue_readback();
    }
    finish_eval(f);
    if (out->func_val_conv) out->func_val_conv[iter] = f[0];
    if (out->func_coupl_conv) out->func_coupl_conv[iter] = f[1];
    if (out->func_constr_conv) out->func_constr_conv[iter] = f[2];
    if (out->func_PAR2_coupl) out->func_PAR2_coupl[iter] = f[3];
    if (out->time_at_it)
      out->time_at_it[iter] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (out->innerIters) {
      for (int m = 0; m < n_modes_; ++m) {
        const ModeInfo& mi = modes_[m];
        double v;
        if (mi.coupling >= 0) v = hctl[n_modes_ + mi.coupling].iters;          // :392
        else if (mi.slabs) v = hctl[m].iters;                                  // :215
        else if (mi.constrained) v = hctl[m].iters;                            // :146
        else v = 1;                                                            // :138
        out->innerIters[(int64_t)(iter - 1) * n_modes_ + m] = v;
      }
    }
    stop = stop_one(f[0], fo[0], opt) && stop_one(f[1], fo[1], opt) && stop_one(f[2], fo[2], opt) &&
           stop_one(f[3], fo[3], opt);                                         // :456
    if (has_miss) {
      double num = 0.0, den = 0.0;
      for (int p = 0; p < n_tensors_; ++p)
        if (tensors_[p].par2 ? tensors_[p].p2.has_mask : tensors_[p].blk.has_mask) { num += hem[4 * p]; den += hem[4 * p + 1]; }
      f_rel_missing = den > 0 ? std::sqrt(num / den) : std::sqrt(num);        // :436-440
      if (out->func_rel_missing) out->func_rel_missing[iter] = f_rel_missing;
      stop = stop && (f_rel_missing < opt.OuterRelTol);                        // :457-459
    }
    if (report && iter % progress_every_ == 0) progress_fn_(progress_user_, iter, f, f_rel_missing);   // :462-468
    ++iter;
  }
  out->f_tensors = f[0]; out->f_couplings = f[1]; out->f_constraints = f[2]; out->f_PAR2_couplings = f[3];
  out->f_rel_missing = f_rel_missing;
  for (int p = 0; p < n_tensors_; ++p)
    if (tensors_[p].par2) par2_gather_slabs(tensors_[p]);
  AO_HIP(hipStreamSynchronize(stream_));
  out->OuterIterations = iter - 1;
  out->exit_code = iter > opt.MaxOuterIters ? 0 : 1;                           // make_exit_flag.m:4-5
  for (int i = 0; i < 4; ++i) out->exit_abs[i] = f[i] < opt.AbsFuncTol ? 1 : 0;
}

// Y = X_(n) X_(n)' of the RESIDENT data of tensor p (cmtf_nvecs.m:31-56, init_coupled_AOADMM_CMTF.m:50-73): the Gram
// matrix whose leading eigenvectors initialise mode `pos` with init_options.nvecs = 1, without another transfer of the
// tensor.  CP blocks (matrices, 3-way): any mode; PARAFAC2 blocks: pos 0 = [X_1 ... X_K] X_k' summed, pos 1 = X_k' X_k
// of slab `slab`.  With a communicator the first mode of a row-sharded block has no local answer (its Gram matrix pairs
// rows of different ranks): AOADMM_ERR_UNSUPPORTED, the caller takes aoadmm_op_unfold_gram with the host array.
void Engine::resident_unfold_gram(int p, int pos, int slab, double* out_host) {
  require_usable();
  AO_REQUIRE(model_done_, "call aoadmm_model_end first");
  AO_REQUIRE(p >= 0 && p < n_tensors_, "tensor %d out of range", p);     // out_host may be null: ranks > 0 of a multi-device context
  AO_HIP(hipSetDevice(device_));
  TensorInfo& t = tensors_[p];
  UnfoldGramArgs a;
  int prec = AOADMM_PREC_F64;
  bool reduce = false;
  if (t.par2) {
    const Par2Block& b = t.p2;
    AO_REQUIRE(pos == 0 || pos == 1, "PARAFAC2 block: mode 1 (all slabs) or mode 2 (one slab)");
    for (int k = 0; k < b.K; ++k) AO_REQUIRE(b.have_slab[k], "slab %d of tensor %d has no data", k, p);
    if (pos == 0) { a.X = b.X.p; a.n = b.I; a.sa = 1; a.n1 = b.Jtot; a.s1 = b.I; a.n2 = 1; a.s2 = 0; }
    else {
      AO_REQUIRE(slab >= 0 && slab < b.K, "slab %d out of range", slab);
      const int64_t Jk = b.off_h[slab + 1] - b.off_h[slab];
      a.X = b.X.d() + (int64_t)b.I * b.off_h[slab]; a.n = Jk; a.sa = b.I; a.n1 = b.I; a.s1 = 1; a.n2 = 1; a.s2 = 0;
    }
  } else {
    const CpBlock& b = t.blk;
    AO_REQUIRE(b.has_data, "tensor %d has no data", p);
    AO_REQUIRE((b.nd == 2 || b.nd == 3) && pos >= 0 && pos < b.nd, "unfold_gram handles matrices and 3-way tensors");
    if (sharded() && pos == 0)
      throw Error(AOADMM_ERR_UNSUPPORTED, "resident unfold_gram: the first mode of a row-sharded block pairs rows of different ranks");
    if (b.x_released)
      throw Error(AOADMM_ERR_UNSUPPORTED, "resident unfold_gram: the natural-layout array was released (only the pass copies are resident)");
    const int64_t I = b.dims[0], Ip = b.X.pad0, J = b.dims[1], K = b.nd == 3 ? b.dims[2] : 1;
    prec = b.X.prec;
    a.X = b.X.data.p;
    if (pos == 0) { a.n = I; a.sa = 1; a.n1 = J * K; a.s1 = Ip; a.n2 = 1; a.s2 = 0; }
    else if (pos == 1) { a.n = J; a.sa = Ip; a.n1 = I; a.s1 = 1; a.n2 = K; a.s2 = Ip * J; }
    else { a.n = K; a.sa = Ip * J; a.n1 = Ip * J; a.s1 = 1; a.n2 = 1; a.s2 = 0; }   // padding rows are zeros
    reduce = sharded();                                 // partial sums over this rank's rows
  }
  DevBuf ws, y;
  ws.alloc(unfold_gram_ws_bytes(a));
  y.alloc((size_t)a.n * a.n * sizeof(double));
  unfold_gram(a, prec, ws.d(), y.d(), stream_);
  if (reduce) allreduce(y.d(), a.n * a.n);
  if (out_host) AO_HIP(hipMemcpyAsync(out_host, y.p, (size_t)a.n * a.n * sizeof(double), hipMemcpyDeviceToHost, stream_));
  AO_HIP(hipStreamSynchronize(stream_));
}

void Engine::resident_mttkrp(int p, int pos, double* out_host, float* ms) {
  AO_REQUIRE(model_done_, "call aoadmm_model_end first");
  AO_REQUIRE(p >= 0 && p < n_tensors_, "tensor %d out of range", p);
  AO_HIP(hipSetDevice(device_));
  TensorInfo& t = tensors_[p];
  AO_REQUIRE(pos >= 0 && pos < t.nmodes, "tensor mode %d out of range", pos);
  FactorRef facs[8];
  for (int i = 0; i < t.nmodes; ++i) {
    ModeInfo& o = modes_[t.modes[i]];
    AO_REQUIRE(o.has_fac, "G.fac{%d} missing", t.modes[i] + 1);
    facs[i] = factor_ref(o);
  }
  ModeInfo& mi = modes_[t.modes[pos]];
  ensure_mode_work(mi);
  hipEvent_t e0, e1;
  AO_HIP(hipEventCreate(&e0)); AO_HIP(hipEventCreate(&e1));
  AO_HIP(hipEventRecord(e0, stream_));
  t.blk.cached_mode = -1;                              // a full MTTKRP: tensor pass + reduction, on the pass's resident copy
  block_mttkrp(t.blk, pos, facs, mi.R, 1.0, mi.A.d(), mi.rows, true, nullptr, 0, true, true);
  t.blk.cached_mode = -1;                              // the solver's own factors may differ from what this pass used
  AO_HIP(hipEventRecord(e1, stream_));
  AO_HIP(hipEventSynchronize(e1));
  float tms = 0.f;
  AO_HIP(hipEventElapsedTime(&tms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (ms) *ms = tms;
  if (out_host) {
    AO_HIP(hipMemcpyAsync(out_host, mi.A.p, (size_t)mi.rows * mi.R * sizeof(double), hipMemcpyDeviceToHost, stream_));
    AO_HIP(hipStreamSynchronize(stream_));
  }
}

}  // namespace aoadmm
