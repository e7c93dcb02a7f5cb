// TEST INFRASTRUCTURE: glog stand-in for compiling the reference's sources. LOG(INFO / WARNING) and VLOG are discarded,
// LOG(ERROR) goes to stderr, LOG(FATAL) and a failing CHECK print and abort like glog.
#pragma once
#include <cstdlib>
#include <iostream>
#include <sstream>
struct ShimNullLog {
  template <typename T> ShimNullLog& operator<<(const T&) { return *this; }
  ShimNullLog& operator<<(std::ostream& (*)(std::ostream&)) { return *this; }
};
struct ShimLog {
  bool fatal;
  std::ostringstream ss;
  explicit ShimLog(bool f) : fatal(f) {}
  template <typename T> ShimLog& operator<<(const T& v) { ss << v; return *this; }
  ShimLog& operator<<(std::ostream& (*f)(std::ostream&)) { ss << f; return *this; }
  ~ShimLog() {
    std::cerr << ss.str() << std::endl;
    if (fatal) std::abort();
  }
};
struct ShimVoidify { void operator&(const ShimLog&) {} void operator&(const ShimNullLog&) {} };
#define SHIM_LOG_INFO ShimNullLog()
#define SHIM_LOG_WARNING ShimNullLog()
#define SHIM_LOG_ERROR ShimLog(false)
#define SHIM_LOG_FATAL ShimLog(true)
#define LOG(x) SHIM_LOG_##x
#define VLOG(x) ShimNullLog()
#define LOG_IF(x, c) ShimNullLog()
#define SHIM_CHECK(cond, text) (cond) ? (void)0 : ShimVoidify() & ShimLog(true) << "Check failed: " text " "
#define CHECK(x) SHIM_CHECK((x), #x)
#define CHECK_EQ(a, b) SHIM_CHECK((a) == (b), #a " == " #b)
#define CHECK_NE(a, b) SHIM_CHECK((a) != (b), #a " != " #b)
#define CHECK_LT(a, b) SHIM_CHECK((a) < (b), #a " < " #b)
#define CHECK_LE(a, b) SHIM_CHECK((a) <= (b), #a " <= " #b)
#define CHECK_GT(a, b) SHIM_CHECK((a) > (b), #a " > " #b)
#define CHECK_GE(a, b) SHIM_CHECK((a) >= (b), #a " >= " #b)
template <typename T> inline T shim_check_notnull(T p, const char* text) {
  if (p == nullptr) { std::cerr << "Check failed: '" << text << "' Must be non NULL" << std::endl; std::abort(); }
  return p;
}
#define CHECK_NOTNULL(p) shim_check_notnull((p), #p)
namespace google {
inline void InitGoogleLogging(const char*) {}
inline void InstallFailureSignalHandler() {}
}  // namespace google
