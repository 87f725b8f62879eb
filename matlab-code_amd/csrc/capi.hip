// extern "C" boundary of libaoadmm_hip.so (include/aoadmm_hip.h).  No exception
// leaves this file; every entry point returns a status and records the message.
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <thread>

#include "solver.h"
#include "misc.h"

using namespace aoadmm;

static thread_local std::string g_last_error;

// One process, several GPUs (the shape a MATLAB session needs: SURVEY 8b): the context owns one engine per device
// and one host thread per engine.  Every model/data/state/solve call is handed to all workers at once -- the ranks
// must enter the collectives inside those calls together -- and returns when the last worker is done.
struct MultiCtx {
  std::vector<std::unique_ptr<Engine>> eng;
  std::vector<std::thread> th;
  std::mutex m;
  std::condition_variable cv_job, cv_done;
  std::function<void(Engine&, int)> job;
  uint64_t gen = 0;
  int pending = 0;
  bool quit = false;
  bool poisoned = false;                  // set when run() had to abort communicators; never cleared
  std::vector<int> code;
  std::vector<std::string> msg;
  std::vector<char> done;
  // options.Display = 'iter': rank 0's worker queues the rows, the CALLING thread delivers them from inside run()
  // (a MEX callback may only touch MATLAB from the interpreter's thread)
  struct ProgressRow { int iter; double f[4]; double frm; };
  std::deque<ProgressRow> rows;
  aoadmm_progress_fn user_fn = nullptr;
  void* user_arg = nullptr;
  static void queue_row(void* self, int iter, const double f[4], double frm) {
    MultiCtx* mc = static_cast<MultiCtx*>(self);
    ProgressRow r;
    r.iter = iter; r.frm = frm;
    for (int i = 0; i < 4; ++i) r.f[i] = f[i];
    std::lock_guard<std::mutex> lk(mc->m);
    mc->rows.push_back(r);
    mc->cv_done.notify_all();
  }

  void worker(int r, int device) {
    uint64_t seen = 0;
    for (;;) {
      std::function<void(Engine&, int)> f;
      {
        std::unique_lock<std::mutex> lk(m);
        cv_job.wait(lk, [&] { return quit || gen != seen; });
        if (quit) return;
        seen = gen;
        f = job;
      }
      int c = AOADMM_OK;
      std::string w;
      try {
        (void)hipSetDevice(device);
        f(*eng[r], r);
      } catch (const Error& e) { c = e.code; w = e.what(); }
      catch (const std::bad_alloc&) { c = AOADMM_ERR_NOMEM; w = "host allocation failed"; }
      catch (const std::exception& e) { c = AOADMM_ERR_INVALID; w = e.what(); }
      catch (...) { c = AOADMM_ERR_INVALID; w = "unknown failure"; }
      {
        std::lock_guard<std::mutex> lk(m);
        code[r] = c; msg[r] = w; done[r] = 1;
        --pending;
        cv_done.notify_all();             // also on failure: run() starts its grace period for the peers
      }
    }
  }
  // run f on every engine concurrently; progress rows are delivered on this (the caller's) thread while it waits.
  // Throws the failure of the rank that failed by itself: a rank that only reports the abort it received
  // (AOADMM_ERR_RCCL) is passed over when another rank has the cause.
  void run(const std::function<void(Engine&, int)>& f) {
    {
      std::unique_lock<std::mutex> lk(m);
      // sticky: once a peer's communicator had to be aborted the ranks are out of step for good (the aborted ones would
      // skip collectives the failed one still enters), so every later call fails at once instead of computing on
      if (poisoned) throw Error(AOADMM_ERR_RCCL, "multi-device context unusable: a rank failed alone and its peers' communicators were aborted; destroy the context");
      job = f;
      pending = (int)eng.size();
      done.assign(eng.size(), 0);
      code.assign(eng.size(), AOADMM_OK);
      ++gen;
      cv_job.notify_all();
      // A rank that failed alone leaves its peers waiting for it inside a collective.  Failures that every rank
      // detects (bad arguments) finish everywhere within milliseconds; if some rank is still busy kAbortGrace after
      // another one failed, its communicator is aborted so that it returns (the context is unusable afterwards).
      const auto kAbortGrace = std::chrono::seconds(5);
      bool failing = false, aborted = false;
      std::chrono::steady_clock::time_point t_fail;
      for (;;) {
        cv_done.wait_for(lk, std::chrono::milliseconds(500), [&] { return pending == 0 || !rows.empty(); });
        if (!aborted && pending > 0) {
          bool any = false;
          for (size_t r = 0; r < eng.size(); ++r) any = any || (done[r] && code[r] != AOADMM_OK);
          if (any && !failing) { failing = true; t_fail = std::chrono::steady_clock::now(); }
          if (failing && std::chrono::steady_clock::now() - t_fail > kAbortGrace) {
            for (size_t r = 0; r < eng.size(); ++r)
              if (!done[r]) eng[r]->comm_abort();
            aborted = true;
            poisoned = true;
          }
        }
        while (!rows.empty()) {
          const ProgressRow r = rows.front();
          rows.pop_front();
          aoadmm_progress_fn fn = user_fn;
          void* arg = user_arg;
          lk.unlock();
          if (fn) fn(arg, r.iter, r.f, r.frm);
          lk.lock();
        }
        if (pending == 0) break;
      }
    }
    int first = -1;
    for (size_t r = 0; r < eng.size(); ++r)
      if (code[r] != AOADMM_OK && (first < 0 || (code[first] == AOADMM_ERR_RCCL && code[r] != AOADMM_ERR_RCCL))) first = (int)r;
    if (first >= 0) throw Error(code[first], fmt("rank %d: %s", first, msg[first].c_str()));
  }
  ~MultiCtx() {
    {
      std::lock_guard<std::mutex> lk(m);
      quit = true;
      cv_job.notify_all();
    }
    for (auto& t : th)
      if (t.joinable()) t.join();
  }
};

struct aoadmm_ctx {
  Engine* eng;          // the engine (single device) or rank 0's engine (multi-device)
  MultiCtx* multi;
};

// f(engine, rank) on the one engine, or on every engine of a multi-device context at once
template <class F>
static void on_engines(aoadmm_ctx* ctx, F&& f) {
  if (ctx->multi) ctx->multi->run(std::function<void(Engine&, int)>(f));
  else f(*ctx->eng, 0);
}

template <class F>
static int guarded(F&& f) {
  try {
    f();
    return AOADMM_OK;
  } catch (const Error& e) {
    g_last_error = e.what();
    return e.code;
  } catch (const std::bad_alloc&) {
    g_last_error = "host allocation failed";
    return AOADMM_ERR_NOMEM;
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return AOADMM_ERR_INVALID;
  } catch (...) {
    g_last_error = "unknown failure";
    return AOADMM_ERR_INVALID;
  }
}

#define CTX_OR_FAIL(ctx)                                   \
  if (!(ctx) || !(ctx)->eng) {                             \
    g_last_error = "null context";                         \
    return AOADMM_ERR_INVALID;                             \
  }

extern "C" {

int aoadmm_abi_version(void) { return AOADMM_ABI_VERSION; }
const char* aoadmm_last_error(void) { return g_last_error.c_str(); }

int aoadmm_device_count(int* n) {
  return guarded([&] {
    AO_REQUIRE(n != nullptr, "null pointer");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    *n = (e == hipSuccess) ? c : 0;
  });
}

int aoadmm_create(aoadmm_ctx** ctx, int device) {
  return guarded([&] {
    AO_REQUIRE(ctx != nullptr, "null pointer");
    *ctx = nullptr;
    Engine* e = new Engine(device);
    aoadmm_ctx* c = new aoadmm_ctx;
    c->eng = e;
    c->multi = nullptr;
    *ctx = c;
  });
}

int aoadmm_create_multi(aoadmm_ctx** ctx, int n_devices, const int* devices) {
  return guarded([&] {
    AO_REQUIRE(ctx != nullptr && devices != nullptr && n_devices >= 1 && n_devices <= 64, "bad arguments");
    *ctx = nullptr;
    std::unique_ptr<MultiCtx> mc(new MultiCtx);
    const int n = n_devices;
    mc->eng.resize(n); mc->code.assign(n, AOADMM_OK); mc->msg.resize(n); mc->done.assign(n, 0);
    for (int r = 0; r < n; ++r) mc->eng[r].reset(new Engine(devices[r]));
    for (int r = 0; r < n; ++r) mc->th.emplace_back(&MultiCtx::worker, mc.get(), r, devices[r]);
    bool distinct = true;
    for (int a = 0; a < n; ++a)
      for (int b = a + 1; b < n; ++b) distinct = distinct && devices[a] != devices[b];
    if (n > 1) {
      if (distinct) {                                 // RCCL over xGMI, one rank per device
        ncclUniqueId uid;
        ncclResult_t r = ncclGetUniqueId(&uid);
        if (r != ncclSuccess) throw Error(AOADMM_ERR_RCCL, fmt("ncclGetUniqueId failed: %s", ncclGetErrorString(r)));
        char id[128];
        std::memset(id, 0, sizeof id);
        std::memcpy(id, &uid, sizeof(uid));
        mc->run([&](Engine& e, int rank) { e.comm_init(id, rank, n); });
      } else {                                        // a device listed twice: bring-up/test transport (see comm_init_local)
        static std::atomic<int> next_key{1 << 20};
        const int key = next_key++;
        mc->run([&](Engine& e, int rank) { e.comm_init_local(key, rank, n); });
      }
    }
    aoadmm_ctx* c = new aoadmm_ctx;
    c->eng = mc->eng[0].get();
    c->multi = mc.release();
    *ctx = c;
  });
}

int aoadmm_destroy(aoadmm_ctx* ctx) {
  return guarded([&] {
    if (!ctx) return;
    if (ctx->multi) delete ctx->multi;                // joins the workers, destroys every engine
    else delete ctx->eng;
    delete ctx;
  });
}

int aoadmm_synchronize(aoadmm_ctx* ctx) {
  CTX_OR_FAIL(ctx);
  return guarded([&] {
    on_engines(ctx, [&](Engine& e, int) {
      AO_HIP(hipSetDevice(e.device()));
      AO_HIP(hipStreamSynchronize(e.stream()));
    });
  });
}

int aoadmm_set_progress(aoadmm_ctx* ctx, aoadmm_progress_fn fn, void* user, int every) {
  CTX_OR_FAIL(ctx);
  return guarded([&] {                               // rank 0 reports; the caller's thread delivers (MultiCtx::run)
    if (ctx->multi) {
      MultiCtx* mc = ctx->multi;
      { std::lock_guard<std::mutex> lk(mc->m); mc->user_fn = fn; mc->user_arg = user; }
      const bool on = fn != nullptr && every > 0;
      on_engines(ctx, [&](Engine& e, int r) {
        const bool mine = on && r == 0;
        e.set_progress(mine ? &MultiCtx::queue_row : nullptr, mine ? static_cast<void*>(mc) : nullptr, mine ? every : 0);
      });
    } else {
      ctx->eng->set_progress(fn, user, every);
    }
  });
}

int aoadmm_comm_unique_id(char id[128]) {
  return guarded([&] {
    AO_REQUIRE(id != nullptr, "null pointer");
    ncclUniqueId uid;
    ncclResult_t r = ncclGetUniqueId(&uid);
    if (r != ncclSuccess) throw Error(AOADMM_ERR_RCCL, fmt("ncclGetUniqueId failed: %s", ncclGetErrorString(r)));
    std::memset(id, 0, 128);
    std::memcpy(id, &uid, sizeof(uid));
  });
}
int aoadmm_comm_init_rank(aoadmm_ctx* ctx, const char id[128], int rank, int world) {
  CTX_OR_FAIL(ctx);
  return guarded([&] {
    AO_REQUIRE(!ctx->multi, "a multi-device context owns its communicator");
    ctx->eng->comm_init(id, rank, world);
  });
}
int aoadmm_comm_init_rank_share(aoadmm_ctx* ctx, const char id[128], int rank, int world) {
  CTX_OR_FAIL(ctx);
  return guarded([&] {
    AO_REQUIRE(!ctx->multi, "a multi-device context owns its communicator");
    AO_REQUIRE(id != nullptr, "aoadmm_comm_init_rank_share needs the id from aoadmm_comm_unique_id");
    ctx->eng->comm_init(id, rank, world, true);
  });
}
int aoadmm_comm_init_local(aoadmm_ctx* ctx, int key, int rank, int world) {
  CTX_OR_FAIL(ctx);
  return guarded([&] {
    AO_REQUIRE(!ctx->multi, "a multi-device context owns its communicator");
    ctx->eng->comm_init_local(key, rank, world);
  });
}
int aoadmm_comm_rank(aoadmm_ctx* ctx, int* rank, int* world) {
  CTX_OR_FAIL(ctx);
  return guarded([&] {
    if (rank) *rank = ctx->eng->rank();
    if (world) *world = ctx->eng->world();
  });
}
int aoadmm_comm_info(aoadmm_ctx* ctx, int* nccl_version, int* comm_ranks, char* lib_path, int lib_path_cap) {
  CTX_OR_FAIL(ctx);
  return guarded([&] { ctx->eng->comm_info(nccl_version, comm_ranks, lib_path, lib_path_cap); });
}

int aoadmm_model_begin(aoadmm_ctx* ctx, int n_modes, int n_tensors, int n_couplings) {
  CTX_OR_FAIL(ctx);
  return guarded([&] { on_engines(ctx, [&](Engine& e, int) { e.model_begin(n_modes, n_tensors, n_couplings); }); });
}
int aoadmm_model_set_mode(aoadmm_ctx* ctx, int mode, int64_t rows, int rank) {
  CTX_OR_FAIL(ctx);
  return guarded([&] { on_engines(ctx, [&](Engine& e, int) { e.set_mode(mode, rows, rank); }); });
}
int aoadmm_model_set_mode_slabs(aoadmm_ctx* ctx, int mode, int K, const int64_t* rows_k, int rank) {
  CTX_OR_FAIL(ctx);
  return guarded([&] {
    AO_REQUIRE(rows_k != nullptr, "null pointer");
    on_engines(ctx, [&](Engine& e, int) { e.set_mode_slabs(mode, K, rows_k, rank); });
  });
}
int aoadmm_model_add_cp(aoadmm_ctx* ctx, int p, int n, const int* modes, double weight) {
  CTX_OR_FAIL(ctx);
  return guarded([&] {
    AO_REQUIRE(modes != nullptr, "null pointer");
    on_engines(ctx, [&](Engine& e, int) { e.add_cp(p, n, modes, weight); });
  });
}
int aoadmm_model_add_par2(aoadmm_ctx* ctx, int p, const int* modes3, double weight) {
  CTX_OR_FAIL(ctx);
  return guarded([&] {
    AO_REQUIRE(modes3 != nullptr, "null pointer");
    on_engines(ctx, [&](Engine& e, int) { e.add_par2(p, modes3, weight); });
  });
}
int aoadmm_model_set_constraint(aoadmm_ctx* ctx, int mode, int constraint, const double* params, int n_params,
                                const double* Lmat) {
  CTX_OR_FAIL(ctx);
  return guarded([&] {
    AO_REQUIRE(n_params == 0 || params != nullptr, "null parameters");
    on_engines(ctx, [&](Engine& e, int) { e.set_constraint(mode, constraint, params, n_params, Lmat); });
  });
}
int aoadmm_model_set_coupling(aoadmm_ctx* ctx, int mode, int coupling, const double* H, int64_t hr, int64_t hc,
                              const double* H2, int64_t h2r, int64_t h2c) {
  CTX_OR_FAIL(ctx);
  return guarded([&] { on_engines(ctx, [&](Engine& e, int) { e.set_coupling(mode, coupling, H, hr, hc, H2, h2r, h2c); }); });
}
int aoadmm_model_set_coupling_type(aoadmm_ctx* ctx, int coupling, int type) {
  CTX_OR_FAIL(ctx);
  return guarded([&] { on_engines(ctx, [&](Engine& e, int) { e.set_coupling_type(coupling, type); }); });
}
int aoadmm_model_set_ridge(aoadmm_ctx* ctx, const double* ridge) {
  CTX_OR_FAIL(ctx);
  return guarded([&] { on_engines(ctx, [&](Engine& e, int) { e.set_ridge(ridge); }); });
}
int aoadmm_model_end(aoadmm_ctx* ctx) {
  CTX_OR_FAIL(ctx);
  return guarded([&] { on_engines(ctx, [&](Engine& e, int) { e.model_end(); }); });
}

int aoadmm_tensor_upload(aoadmm_ctx* ctx, int p, const double* data, int precision) {
  CTX_OR_FAIL(ctx);
  return guarded([&] {
    AO_REQUIRE(data != nullptr, "null data");
    on_engines(ctx, [&](Engine& e, int) { e.tensor_upload(p, data, precision, 0, -1); });
  });
}
int aoadmm_tensor_upload_rows(aoadmm_ctx* ctx, int p, const double* block, int64_t row_offset, int64_t local_rows,
                              int precision) {
  CTX_OR_FAIL(ctx);
  return guarded([&] {
    AO_REQUIRE(block != nullptr && local_rows > 0, "null/empty block");
    // one block cannot be every rank's block: a multi-device context shards the FULL array itself
    AO_REQUIRE(!ctx->multi, "aoadmm_tensor_upload_rows is for one-process-per-GPU contexts; give a multi-device context "
                            "the full array through aoadmm_tensor_upload");
    ctx->eng->tensor_upload(p, block, precision, row_offset, local_rows);
  });
}
int aoadmm_par2_slab_upload(aoadmm_ctx* ctx, int p, int k, const double* Xk) {
  CTX_OR_FAIL(ctx);
  return guarded([&] { on_engines(ctx, [&](Engine& e, int) { e.par2_slab_upload(p, k, Xk); }); });
}
int aoadmm_tensor_synth(aoadmm_ctx* ctx, int p, int rank, uint64_t seed, double noise, int precision) {
  CTX_OR_FAIL(ctx);
  return guarded([&] { on_engines(ctx, [&](Engine& e, int) { e.tensor_synth(p, rank, seed, noise, precision); }); });
}
int aoadmm_tensor_mask_upload(aoadmm_ctx* ctx, int p, const uint8_t* mask) {
  CTX_OR_FAIL(ctx);
  return guarded([&] { on_engines(ctx, [&](Engine& e, int) { e.tensor_mask_upload(p, mask); }); });
}
int aoadmm_par2_slab_mask_upload(aoadmm_ctx* ctx, int p, int k, const uint8_t* mask_k) {
  CTX_OR_FAIL(ctx);
  return guarded([&] { on_engines(ctx, [&](Engine& e, int) { e.par2_slab_mask_upload(p, k, mask_k); }); });
}
int aoadmm_tensor_normsq(aoadmm_ctx* ctx, int p, double* out) {
  CTX_OR_FAIL(ctx);
  return guarded([&] {
    AO_REQUIRE(out != nullptr, "null pointer");
    on_engines(ctx, [&](Engine& e, int r) {
      const double v = e.tensor_normsq(p);           // collective: every rank computes, rank 0 answers
      if (r == 0) *out = v;
    });
  });
}

int aoadmm_state_set(aoadmm_ctx* ctx, int field, int index, int slab, const double* host, int64_t rows,
                     int64_t cols) {
  CTX_OR_FAIL(ctx);
  return guarded([&] { on_engines(ctx, [&](Engine& e, int) { e.state_set(field, index, slab, host, rows, cols); }); });
}
int aoadmm_state_get(aoadmm_ctx* ctx, int field, int index, int slab, double* host, int64_t rows, int64_t cols) {
  CTX_OR_FAIL(ctx);
  return guarded([&] {
    ctx->eng->state_get(field, index, slab, host, rows, cols);   // multi-device: the state is replicated, rank 0 answers
  });
}

int aoadmm_solve(aoadmm_ctx* ctx, const aoadmm_options* opt, aoadmm_result* out) {
  CTX_OR_FAIL(ctx);
  return guarded([&] {
    AO_REQUIRE(opt != nullptr && out != nullptr, "null options/result");
    on_engines(ctx, [&](Engine& e, int r) {
      if (r == 0) { e.solve(*opt, out); return; }
      aoadmm_result mine;                             // the other ranks keep their (identical) results to themselves
      std::memset(&mine, 0, sizeof mine);
      e.solve(*opt, &mine);
    });
  });
}
int aoadmm_resident_mttkrp(aoadmm_ctx* ctx, int p, int tensor_mode, double* out_host_or_null, float* elapsed_ms) {
  CTX_OR_FAIL(ctx);
  return guarded([&] {
    on_engines(ctx, [&](Engine& e, int r) { e.resident_mttkrp(p, tensor_mode, r == 0 ? out_host_or_null : nullptr, r == 0 ? elapsed_ms : nullptr); });
  });
}
int aoadmm_kernel_stats(aoadmm_ctx* ctx, int which, int reset, double* contract_ms, int64_t* contract_launches,
                        double* contract_bytes, double* contract_flops) {
  CTX_OR_FAIL(ctx);
  return guarded([&] {
    on_engines(ctx, [&](Engine& e, int r) {
      if (r == 0) e.kernel_stats(which, reset, contract_ms, contract_launches, contract_bytes, contract_flops);
      else e.kernel_stats(which, reset, nullptr, nullptr, nullptr, nullptr);
    });
  });
}

// ---------------------------------------------------------------------------
// op level
// ---------------------------------------------------------------------------
static void h2d(DevBuf& b, const double* h, int64_t n, hipStream_t s) {
  b.alloc((size_t)n * sizeof(double));
  AO_HIP(hipMemcpyAsync(b.p, h, (size_t)n * sizeof(double), hipMemcpyHostToDevice, s));
}
static void d2h(double* h, const DevBuf& b, int64_t n, hipStream_t s) {
  AO_HIP(hipMemcpyAsync(h, b.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s));
  AO_HIP(hipStreamSynchronize(s));
}

int aoadmm_op_mttkrp(aoadmm_ctx* ctx, const double* X, int ndims, const int64_t* dims, const double* const* U, int R,
                     int n, int precision, double* out) {
  CTX_OR_FAIL(ctx);
  return guarded([&] {
    AO_REQUIRE(X && dims && U && out, "null pointer");
    AO_REQUIRE(ndims >= 2 && ndims <= 8 && n >= 0 && n < ndims, "bad order/mode");
    Engine& e = *ctx->eng;
    AO_HIP(hipSetDevice(e.device()));
    CpBlock blk;
    e.block_upload(blk, ndims, dims, X, precision, 0, dims[0]);
    std::vector<DevBuf> fac(ndims);
    FactorRef refs[8];
    for (int m = 0; m < ndims; ++m) {
      AO_REQUIRE(U[m] != nullptr, "null factor");
      h2d(fac[m], U[m], dims[m] * R, e.stream());
      refs[m] = FactorRef{fac[m].d(), dims[m], 1};
    }
    DevBuf o;
    o.alloc((size_t)dims[n] * R * sizeof(double));
    // host in / host out on ONE engine with the whole tensor: no collective, whatever communicator the engine is in
    e.block_mttkrp(blk, n, refs, R, 1.0, o.d(), dims[n], false, nullptr, 0, false, true);
    d2h(out, o, dims[n] * R, e.stream());
  });
}

int aoadmm_op_unfold_gram(aoadmm_ctx* ctx, const double* X, int ndims, const int64_t* dims, int n, int precision,
                          double* out) {
  CTX_OR_FAIL(ctx);
  return guarded([&] {
    AO_REQUIRE(X && dims && out, "null pointer");
    AO_REQUIRE((ndims == 2 || ndims == 3) && n >= 0 && n < ndims, "unfold_gram handles matrices and 3-way tensors");
    Engine& e = *ctx->eng;
    AO_HIP(hipSetDevice(e.device()));
    CpBlock blk;
    e.block_upload(blk, ndims, dims, X, precision, 0, dims[0]);
    const int64_t I = dims[0], Ip = blk.X.pad0, J = dims[1], K = ndims == 3 ? dims[2] : 1;
    UnfoldGramArgs a;
    a.X = blk.X.data.p;
    if (n == 0) { a.n = I; a.sa = 1; a.n1 = J * K; a.s1 = Ip; a.n2 = 1; a.s2 = 0; }
    else if (n == 1) { a.n = J; a.sa = Ip; a.n1 = I; a.s1 = 1; a.n2 = K; a.s2 = Ip * J; }
    else { a.n = K; a.sa = Ip * J; a.n1 = Ip * J; a.s1 = 1; a.n2 = 1; a.s2 = 0; }   // padding rows are zeros
    DevBuf ws, y;
    ws.alloc(unfold_gram_ws_bytes(a));
    y.alloc((size_t)a.n * a.n * sizeof(double));
    unfold_gram(a, precision, ws.d(), y.d(), e.stream());
    d2h(out, y, a.n * a.n, e.stream());
  });
}

int aoadmm_resident_unfold_gram(aoadmm_ctx* ctx, int p, int tensor_mode, int slab, double* out) {
  CTX_OR_FAIL(ctx);
  return guarded([&] {
    AO_REQUIRE(out != nullptr, "null pointer");
    // every engine of a multi-device context enters (the Gram matrix of a sharded block is all-reduced); rank 0 reports
    on_engines(ctx, [&](Engine& e, int r) { e.resident_unfold_gram(p, tensor_mode, slab, r == 0 ? out : nullptr); });
  });
}

int aoadmm_op_gram(aoadmm_ctx* ctx, const double* F, int64_t rows, int R, double* out) {
  CTX_OR_FAIL(ctx);
  return guarded([&] {
    AO_REQUIRE(F && out && rows > 0 && R > 0 && R <= kMaxRank, "bad arguments");
    Engine& e = *ctx->eng;
    AO_HIP(hipSetDevice(e.device()));
    DevBuf f, g, ws;
    h2d(f, F, rows * R, e.stream());
    g.alloc((size_t)R * R * 8);
    ws.alloc(atb_ws_bytes(rows, R, R));
    atb_small(g.d(), f.d(), rows, f.d(), rows, rows, R, R, ws.d(), nullptr, e.stream());
    d2h(out, g, (int64_t)R * R, e.stream());
  });
}

int aoadmm_op_chol(aoadmm_ctx* ctx, const double* B, int R, double* L) {
  CTX_OR_FAIL(ctx);
  return guarded([&] {
    AO_REQUIRE(B && L && R > 0 && R <= kMaxRank, "bad arguments");
    Engine& e = *ctx->eng;
    AO_HIP(hipSetDevice(e.device()));
    DevBuf b, l, c;
    h2d(b, B, (int64_t)R * R, e.stream());
    l.alloc((size_t)R * R * 8);
    c.alloc(sizeof(AdmmCtl));
    AO_HIP(hipMemsetAsync(c.p, 0, sizeof(AdmmCtl), e.stream()));
    AO_HIP(hipMemsetAsync(l.p, 0, (size_t)R * R * 8, e.stream()));
    chol_only(l.d(), b.d(), R, c.as<AdmmCtl>(), e.stream());
    AdmmCtl h;
    AO_HIP(hipMemcpyAsync(&h, c.p, sizeof h, hipMemcpyDeviceToHost, e.stream()));
    AO_HIP(hipStreamSynchronize(e.stream()));
    if (h.notpd) throw Error(AOADMM_ERR_NOT_PD, "Matrix must be positive definite.");
    d2h(L, l, (int64_t)R * R, e.stream());
  });
}

static ProxSpec make_spec(int constraint, const double* params, int np) {
  ProxSpec ps;
  ps.type = constraint;
  if (np > 0) ps.p0 = params[0];
  if (np > 1) ps.p1 = params[1];
  return ps;
}

int aoadmm_op_prox(aoadmm_ctx* ctx, int constraint, const double* params, int n_params, const double* Lmat,
                   const double* X, int64_t rows, int R, double rho, double* out) {
  CTX_OR_FAIL(ctx);
  return guarded([&] {
    AO_REQUIRE(X && out && rows > 0 && R > 0 && R <= kMaxRank, "bad arguments");
    Engine& e = *ctx->eng;
    AO_HIP(hipSetDevice(e.device()));
    DevBuf x, z, r, ws;
    h2d(x, X, rows * R, e.stream());
    z.alloc((size_t)rows * R * 8);
    h2d(r, &rho, 1, e.stream());
    ws.alloc(prox_ws_bytes(constraint, rows, R));
    ProxSpec ps = make_spec(constraint, params, n_params);
    QuadPrep qp;
    if (constraint == AOADMM_C_QUADRATIC) { qp.build(Lmat, rows, e.stream()); qp.attach(ps); }
    prox_apply(ps, x.d(), rows, z.d(), rows, rows, R, r.d(), 1.0, ws.d(), nullptr, e.stream());
    d2h(out, z, rows * R, e.stream());
  });
}

int aoadmm_op_admm_constrained(aoadmm_ctx* ctx, const double* A, const double* Bsys, double rho, int constraint,
                               const double* params, int n_params, const double* Lmat, int64_t rows, int R,
                               int max_inner, double tol_pr, double tol_du, double* fac, double* Z, double* mu,
                               int* inner_iters) {
  CTX_OR_FAIL(ctx);
  return guarded([&] {
    AO_REQUIRE(A && Bsys && fac && Z && mu && rows > 0 && R > 0 && R <= kMaxRank && max_inner >= 1, "bad arguments");
    Engine& e = *ctx->eng;
    hipStream_t s = e.stream();
    AO_HIP(hipSetDevice(e.device()));
    DevBuf a, b, l, rh, f, z, m, part, V, Zn, ws, ctl, Cd;
    h2d(a, A, rows * R, s); h2d(b, Bsys, (int64_t)R * R, s);
    h2d(f, fac, rows * R, s); h2d(z, Z, rows * R, s); h2d(m, mu, rows * R, s);
    l.alloc((size_t)R * R * 8); rh.alloc(64); Cd.alloc((size_t)R * R * 8);
    part.alloc((size_t)admm_partials(rows) * 4 * 8 + 2048);
    V.alloc((size_t)rows * R * 8); Zn.alloc((size_t)rows * R * 8);
    ws.alloc(prox_ws_bytes(constraint, rows, R));
    ctl.alloc(sizeof(AdmmCtl));
    AO_HIP(hipMemsetAsync(ctl.p, 0, sizeof(AdmmCtl), s));
    // B + rho/2*I and its Cholesky (cmtf_fun_AOADMM.m:141-142): feed sys_build with C = Bsys, w = 1
    // but keep the caller's rho: do it by hand
    std::vector<double> Bh(Bsys, Bsys + (size_t)R * R);
    for (int i = 0; i < R; ++i) Bh[i + (size_t)R * i] += rho / 2;
    DevBuf bb;
    h2d(bb, Bh.data(), (int64_t)R * R, s);
    h2d(rh, &rho, 1, s);
    AO_HIP(hipStreamSynchronize(s));
    chol_only(l.d(), bb.d(), R, ctl.as<AdmmCtl>(), s);
    ctl_reset(ctl.as<AdmmCtl>(), s);
    AdmmMode am;
    am.A = a.d(); am.L = l.d(); am.rho = rh.d(); am.fac = f.d(); am.Z = z.d(); am.mu = m.d();
    am.rows = rows; am.R = R; am.prox = make_spec(constraint, params, n_params);
    QuadPrep qp;
    if (constraint == AOADMM_C_QUADRATIC) { qp.build(Lmat, rows, s); qp.attach(am.prox); }
    admm_constrained_loop(am, part.d(), V.d(), Zn.d(), ws.d(), ctl.as<AdmmCtl>(), max_inner, tol_pr, tol_du, s);
    AdmmCtl h;
    AO_HIP(hipMemcpyAsync(&h, ctl.p, sizeof h, hipMemcpyDeviceToHost, s));
    AO_HIP(hipStreamSynchronize(s));
    if (h.notpd) throw Error(AOADMM_ERR_NOT_PD, "Matrix must be positive definite.");
    if (inner_iters) *inner_iters = h.iters;
    d2h(fac, f, rows * R, s); d2h(Z, z, rows * R, s); d2h(mu, m, rows * R, s);
  });
}

}  // extern "C"
