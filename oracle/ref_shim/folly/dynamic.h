// TEST INFRASTRUCTURE. Stand-in for the part of folly::dynamic / parseJson / readFile that the reference's Camera.cpp reads
// a rig description with (Camera.cpp:44-83, 243-254), over a small JSON reader of its own. Built with -DSUPPRESS_RIG_IO:
// the serialising half of folly::dynamic is not needed.
#pragma once
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace folly {

class dynamic {
 public:
  enum Kind { NUL, BOOL, NUMBER, STRING, ARRAY, OBJECT } kind = NUL;
  double num = 0;
  bool b = false;
  std::string str;
  std::vector<dynamic> arr;
  std::vector<std::pair<std::string, dynamic>> obj;

  dynamic() {}
  const dynamic& operator[](const char* key) const { return at(std::string(key)); }
  const dynamic& operator[](const std::string& key) const { return at(key); }
  const dynamic& operator[](int i) const {
    if (kind != ARRAY || i < 0 || (size_t)i >= arr.size()) throw std::runtime_error("dynamic: bad array index");
    return arr[(size_t)i];
  }
  const dynamic& at(const std::string& key) const {
    if (kind == OBJECT)
      for (const auto& kv : obj) if (kv.first == key) return kv.second;
    throw std::runtime_error("dynamic: no key '" + key + "'");
  }
  size_t count(const char* key) const {
    if (kind == OBJECT)
      for (const auto& kv : obj) if (kv.first == key) return 1;
    return 0;
  }
  size_t size() const { return kind == ARRAY ? arr.size() : kind == OBJECT ? obj.size() : 0; }
  double asDouble() const {
    if (kind == NUMBER) return num;
    if (kind == STRING) return std::strtod(str.c_str(), nullptr);
    if (kind == BOOL) return b ? 1.0 : 0.0;
    throw std::runtime_error("dynamic: not a number");
  }
  const std::string& getString() const {
    if (kind != STRING) throw std::runtime_error("dynamic: not a string");
    return str;
  }
  std::vector<dynamic>::const_iterator begin() const { return arr.begin(); }
  std::vector<dynamic>::const_iterator end() const { return arr.end(); }
  friend std::ostream& operator<<(std::ostream& o, const dynamic& d) {
    switch (d.kind) {
      case NUMBER: return o << d.num;
      case STRING: return o << '"' << d.str << '"';
      case BOOL: return o << (d.b ? "true" : "false");
      case ARRAY: o << "["; for (size_t i = 0; i < d.arr.size(); ++i) o << (i ? "," : "") << d.arr[i]; return o << "]";
      case OBJECT: o << "{"; for (size_t i = 0; i < d.obj.size(); ++i) o << (i ? "," : "") << '"' << d.obj[i].first << "\":" << d.obj[i].second; return o << "}";
      default: return o << "null";
    }
  }
};

namespace detail {
struct JsonReader {
  const std::string& s;
  size_t p = 0;
  explicit JsonReader(const std::string& t) : s(t) {}
  [[noreturn]] void fail(const char* m) const { throw std::runtime_error(std::string("json: ") + m + " at offset " + std::to_string(p)); }
  void ws() { while (p < s.size() && (s[p] == ' ' || s[p] == '\t' || s[p] == '\n' || s[p] == '\r')) ++p; }
  dynamic value() {
    ws();
    if (p >= s.size()) fail("unexpected end");
    dynamic d;
    const char c = s[p];
    if (c == '{') {
      d.kind = dynamic::OBJECT;
      ++p; ws();
      if (p < s.size() && s[p] == '}') { ++p; return d; }
      for (;;) {
        ws();
        const dynamic k = value();
        if (k.kind != dynamic::STRING) fail("object key is not a string");
        ws();
        if (p >= s.size() || s[p] != ':') fail("expected ':'");
        ++p;
        d.obj.emplace_back(k.str, value());
        ws();
        if (p < s.size() && s[p] == ',') { ++p; continue; }
        if (p < s.size() && s[p] == '}') { ++p; return d; }
        fail("expected ',' or '}'");
      }
    }
    if (c == '[') {
      d.kind = dynamic::ARRAY;
      ++p; ws();
      if (p < s.size() && s[p] == ']') { ++p; return d; }
      for (;;) {
        d.arr.push_back(value());
        ws();
        if (p < s.size() && s[p] == ',') { ++p; continue; }
        if (p < s.size() && s[p] == ']') { ++p; return d; }
        fail("expected ',' or ']'");
      }
    }
    if (c == '"') {
      d.kind = dynamic::STRING;
      ++p;
      while (p < s.size() && s[p] != '"') {
        if (s[p] == '\\' && p + 1 < s.size()) {
          const char e = s[p + 1];
          d.str += e == 'n' ? '\n' : e == 't' ? '\t' : e;
          p += 2;
        } else d.str += s[p++];
      }
      if (p >= s.size()) fail("unterminated string");
      ++p;
      return d;
    }
    if (s.compare(p, 4, "true") == 0) { d.kind = dynamic::BOOL; d.b = true; p += 4; return d; }
    if (s.compare(p, 5, "false") == 0) { d.kind = dynamic::BOOL; d.b = false; p += 5; return d; }
    if (s.compare(p, 4, "null") == 0) { p += 4; return d; }
    char* end = nullptr;
    d.num = std::strtod(s.c_str() + p, &end);
    if (end == s.c_str() + p) fail("unexpected character");
    d.kind = dynamic::NUMBER;
    p = (size_t)(end - s.c_str());
    return d;
  }
};
}  // namespace detail

inline dynamic parseJson(const std::string& text) {
  detail::JsonReader r(text);
  dynamic d = r.value();
  r.ws();
  if (r.p != text.size()) r.fail("trailing characters");
  return d;
}
inline bool readFile(const char* path, std::string& out) {
  std::ifstream f(path, std::ios::binary);
  if (!f) { out.clear(); return false; }
  std::ostringstream ss;
  ss << f.rdbuf();
  out = ss.str();
  return true;
}

}  // namespace folly
