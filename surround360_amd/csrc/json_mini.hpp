// json_mini.hpp — the small recursive-descent JSON reader of libs360's host side (objects, arrays, strings, numbers,
// literals): rig descriptions (rig.cpp) and ISP configurations (isp.cpp). Numbers are read with strtod, i.e. they are
// the correctly rounded doubles the reference's parsers produce.
#pragma once
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "core.hpp"

namespace s360 {

struct JV {
  enum T { NUL, NUM, STR, ARR, OBJ, BOOL } t = NUL;
  double num = 0;
  std::string str;
  std::vector<JV> arr;
  std::map<std::string, JV> obj;
  const JV* get(const char* k) const {
    auto it = obj.find(k);
    return it == obj.end() ? nullptr : &it->second;
  }
};
struct JP {
  const char* s;
  const char* e;
  void ws() { while (s < e && std::isspace((unsigned char)*s)) ++s; }
  [[noreturn]] void fail(const char* m) { throw Error(S360_ERR_IO, std::string("json: ") + m); }
  JV document() {  // one value, then nothing but white space
    JV v = value();
    ws();
    if (s != e) fail("trailing characters after the document");
    return v;
  }
  JV value() {
    ws();
    if (s >= e) fail("unexpected end");
    JV v;
    if (*s == '{') {
      ++s; v.t = JV::OBJ; ws();
      if (s < e && *s == '}') { ++s; return v; }
      for (;;) {
        ws();
        JV k = string();
        ws();
        if (s >= e || *s != ':') fail("expected ':'");
        ++s;
        v.obj[k.str] = value();
        ws();
        if (s < e && *s == ',') { ++s; continue; }
        if (s < e && *s == '}') { ++s; break; }
        fail("expected ',' or '}'");
      }
    } else if (*s == '[') {
      ++s; v.t = JV::ARR; ws();
      if (s < e && *s == ']') { ++s; return v; }
      for (;;) {
        v.arr.push_back(value());
        ws();
        if (s < e && *s == ',') { ++s; continue; }
        if (s < e && *s == ']') { ++s; break; }
        fail("expected ',' or ']'");
      }
    } else if (*s == '"') {
      v = string();
    } else if (!std::strncmp(s, "true", 4)) { v.t = JV::BOOL; v.num = 1; s += 4;
    } else if (!std::strncmp(s, "false", 5)) { v.t = JV::BOOL; s += 5;
    } else if (!std::strncmp(s, "null", 4)) { s += 4;
    } else {
      char* end = nullptr;
      v.num = std::strtod(s, &end);
      if (end == s) fail("bad number");
      v.t = JV::NUM;
      s = end;
    }
    return v;
  }
  JV string() {
    if (s >= e || *s != '"') fail("expected string");
    ++s;
    JV v;
    v.t = JV::STR;
    while (s < e && *s != '"') {
      if (*s == '\\' && s + 1 < e) {
        ++s;
        switch (*s) {
          case 'n': v.str += '\n'; break;
          case 't': v.str += '\t'; break;
          case 'u':  // (ids and names of this format are ASCII: a \\uXXXX escape becomes one placeholder character)
            if (e - s < 5) fail("truncated \\u escape");
            s += 4; v.str += '?'; break;
          default: v.str += *s;
        }
        ++s;
      } else v.str += *s++;
    }
    if (s >= e) fail("unterminated string");
    ++s;
    return v;
  }
};

}  // namespace s360
