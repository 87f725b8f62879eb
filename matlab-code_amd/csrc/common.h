// Internal helpers shared by the HIP translation units of libaoadmm_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <cstdarg>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/aoadmm_hip.h"

namespace aoadmm {

// ---- error plumbing: exceptions stay inside the library, the C ABI converts --
struct Error : public std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

inline std::string fmt(const char* f, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, f);
  vsnprintf(buf, sizeof buf, f, ap);
  va_end(ap);
  return std::string(buf);
}

#define AO_HIP(expr)                                                                   \
  do {                                                                                 \
    hipError_t e__ = (expr);                                                           \
    if (e__ != hipSuccess)                                                             \
      throw ::aoadmm::Error(AOADMM_ERR_HIP, ::aoadmm::fmt("%s failed: %s (%s:%d)", #expr, \
                                                           hipGetErrorString(e__), __FILE__, __LINE__)); \
  } while (0)

#define AO_REQUIRE(cond, ...)                                                      \
  do {                                                                             \
    if (!(cond)) throw ::aoadmm::Error(AOADMM_ERR_INVALID, ::aoadmm::fmt(__VA_ARGS__)); \
  } while (0)

#define AO_KERNEL_CHECK() AO_HIP(hipGetLastError())

// ---- owning device buffer ----------------------------------------------------
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  bool owned = true;                 // false: a view into another buffer (see view())
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes), owned(o.owned) { o.p = nullptr; o.bytes = 0; o.owned = true; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) { release(); p = o.p; bytes = o.bytes; owned = o.owned; o.p = nullptr; o.bytes = 0; o.owned = true; }
    return *this;
  }
  ~DevBuf() { release(); }
  void release() {
    if (p && owned) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    owned = true;
  }
  // n bytes at ptr inside a buffer someone else owns and outlives this one (ensure() up to n bytes keeps the view)
  void view(void* ptr, size_t n) {
    release();
    p = ptr; bytes = n; owned = false;
  }
  void alloc(size_t n) {
    release();
    if (n == 0) n = 16;
    hipError_t e = hipMalloc(&p, n);
    if (e != hipSuccess) {
      p = nullptr;
      throw Error(e == hipErrorOutOfMemory ? AOADMM_ERR_NOMEM : AOADMM_ERR_HIP,
                  fmt("hipMalloc(%zu bytes) failed: %s", n, hipGetErrorString(e)));
    }
    bytes = n;
  }
  void ensure(size_t n) { if (n > bytes) alloc(n); }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
  double* d() const { return as<double>(); }
};

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

constexpr int kMaxRank = 64;   // R <= 64 (two 32-wide MFMA N tiles); typical R <= 32

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to a (device, kernel) pair, not to the process: a context that
// drives several devices from one process (aoadmm_create_multi) must opt in on each of them.  Thread-safe; a
// larger request for a pair that already opted in raises the limit.  (misc.hip)
void ensure_dynamic_lds(const void* kernel, int bytes);

}  // namespace aoadmm
