// core.hpp — device-memory, error and profiling plumbing of libs360 (host side, HIP runtime).
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace s360 {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define S360_HIP(expr)                                                                                         \
  do {                                                                                                         \
    hipError_t e_ = (expr);                                                                                    \
    if (e_ != hipSuccess)                                                                                      \
      throw ::s360::Error(-3, std::string(#expr) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" +          \
                                  std::to_string(__LINE__) + ")");                                            \
  } while (0)

// Grow-only device buffer: persistent across frames (sizes are fixed by rig + eqr size).
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  DevBuf() {}
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  void ensure(size_t bytes) {
    if (bytes <= cap) return;
    release();
    S360_HIP(hipMalloc(&p, bytes));
    cap = bytes;
  }
  template <typename T>
  T* as() const { return reinterpret_cast<T*>(p); }
};

// Per-kernel-family device timing with HIP events on the stream the kernels run on.
struct Profiler {
  bool on = false;
  hipStream_t st = nullptr;
  struct Rec { int name; hipEvent_t a, b; };
  std::vector<std::string> names;
  std::map<std::string, int> ids;
  std::vector<Rec> recs;
  std::vector<hipEvent_t> pool;
  hipEvent_t ev() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e;
    S360_HIP(hipEventCreate(&e));
    return e;
  }
  int begin(const char* name) {
    if (!on) return -1;
    auto it = ids.find(name);
    int id;
    if (it == ids.end()) { id = (int)names.size(); names.push_back(name); ids[name] = id; } else id = it->second;
    Rec r{id, ev(), ev()};
    S360_HIP(hipEventRecord(r.a, st));
    recs.push_back(r);
    return (int)recs.size() - 1;
  }
  // Called from ~ProfScope (implicitly noexcept): never throws; a failed record marks the profile invalid and the
  // error surfaces at the next checked HIP call on the stream.
  void end(int h) noexcept {
    if (h >= 0 && hipEventRecord(recs[h].b, st) != hipSuccess) broken = true;
  }
  bool broken = false;
  void clear() {
    for (auto& r : recs) { pool.push_back(r.a); pool.push_back(r.b); }
    recs.clear();
  }
  // sums per name; resets the record list
  void collect(std::vector<float>& ms, std::vector<int>& cnt) {
    ms.assign(names.size(), 0.f);
    cnt.assign(names.size(), 0);
    if (broken) { broken = false; clear(); throw Error(-3, "profiler: an event could not be recorded"); }
    if (!recs.empty()) S360_HIP(hipEventSynchronize(recs.back().b));
    for (auto& r : recs) {
      float t = 0;
      S360_HIP(hipEventElapsedTime(&t, r.a, r.b));
      ms[r.name] += t;
      cnt[r.name] += 1;
    }
    clear();
  }
  ~Profiler() {
    clear();
    for (auto e : pool) (void)hipEventDestroy(e);
  }
};
struct ProfScope {
  Profiler& p;
  int h;
  ProfScope(Profiler& p_, const char* n) : p(p_), h(p_.begin(n)) {}
  ~ProfScope() { p.end(h); }
};

// api.hip: is the context with this uid (s360_ctx::uid, unique per process) still alive?
bool context_alive(unsigned long long uid);

}  // namespace s360
