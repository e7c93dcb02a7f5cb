// TEST INFRASTRUCTURE. gflags stand-in for compiling the reference's programs from /root/reference: DEFINE_* create the
// FLAGS_ variables in gflags' own namespaces (fLS, fLI, fLD, fLB) and register them; ParseCommandLineNonHelpFlags reads
// --name=value / --name value / -name value (and --name / --noname for booleans).
#pragma once
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <map>
#include <string>

namespace google {
struct FlagRef {
  char type;  // 's', 'i', 'd', 'b'
  void* p;
};
inline std::map<std::string, FlagRef>& registry() {
  static std::map<std::string, FlagRef> r;
  return r;
}
struct Registrar {
  Registrar(const char* name, char type, void* p) { registry()[name] = FlagRef{type, p}; }
};
inline void set_flag(const FlagRef& f, const std::string& v) {
  switch (f.type) {
    case 's': *static_cast<std::string*>(f.p) = v; break;
    case 'i': *static_cast<int*>(f.p) = std::atoi(v.c_str()); break;
    case 'd': *static_cast<double*>(f.p) = std::strtod(v.c_str(), nullptr); break;
    default: *static_cast<bool*>(f.p) = !(v == "0" || v == "false" || v == "no" || v == "f" || v == "n"); break;
  }
}
inline unsigned ParseCommandLineNonHelpFlags(int* argc, char*** argv, bool) {
  for (int i = 1; i < *argc; ++i) {
    std::string a = (*argv)[i];
    if (a.size() < 2 || a[0] != '-') { std::cerr << "shim gflags: unexpected argument " << a << std::endl; std::exit(1); }
    a = a.substr(a[1] == '-' ? 2 : 1);
    std::string key = a, val;
    bool has = false;
    const size_t eq = a.find('=');
    if (eq != std::string::npos) { key = a.substr(0, eq); val = a.substr(eq + 1); has = true; }
    auto it = registry().find(key);
    if (it == registry().end() && key.compare(0, 2, "no") == 0 && registry().count(key.substr(2)) && registry()[key.substr(2)].type == 'b') {
      *static_cast<bool*>(registry()[key.substr(2)].p) = false;
      continue;
    }
    if (it == registry().end()) { std::cerr << "ERROR: unknown command line flag '" << key << "'" << std::endl; std::exit(1); }
    if (!has) {
      if (it->second.type == 'b') { *static_cast<bool*>(it->second.p) = true; continue; }
      if (i + 1 >= *argc) { std::cerr << "ERROR: flag '" << key << "' is missing its argument" << std::endl; std::exit(1); }
      val = (*argv)[++i];
    }
    set_flag(it->second, val);
  }
  return 1;
}
inline unsigned ParseCommandLineFlags(int* argc, char*** argv, bool b) { return ParseCommandLineNonHelpFlags(argc, argv, b); }
inline void HandleCommandLineHelpFlags() {}
}  // namespace google
namespace gflags = google;

#define SHIM_DEFINE_FLAG(ns, type, code, name, def)                 \
  namespace ns {                                                    \
  type FLAGS_##name = def;                                          \
  static google::Registrar shim_reg_##name(#name, code, &FLAGS_##name); \
  }                                                                 \
  using ns::FLAGS_##name
#define DEFINE_string(name, def, help) SHIM_DEFINE_FLAG(fLS, std::string, 's', name, def)
#define DEFINE_int32(name, def, help) SHIM_DEFINE_FLAG(fLI, int, 'i', name, def)
#define DEFINE_double(name, def, help) SHIM_DEFINE_FLAG(fLD, double, 'd', name, def)
#define DEFINE_bool(name, def, help) SHIM_DEFINE_FLAG(fLB, bool, 'b', name, def)
#define DECLARE_string(name) namespace fLS { extern std::string FLAGS_##name; } using fLS::FLAGS_##name
#define DECLARE_int32(name) namespace fLI { extern int FLAGS_##name; } using fLI::FLAGS_##name
#define DECLARE_double(name) namespace fLD { extern double FLAGS_##name; } using fLD::FLAGS_##name
#define DECLARE_bool(name) namespace fLB { extern bool FLAGS_##name; } using fLB::FLAGS_##name
