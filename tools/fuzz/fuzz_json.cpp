// mutates valid rig / ISP JSON texts and the flow .bin container; calls the C ABI's host-only entry points
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <vector>
#include "../../include/s360.h"
static std::string slurp(const char* p) { std::ifstream f(p, std::ios::binary); return std::string((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>()); }
int main(int argc, char** argv) {
  const std::string mode = argv[1];
  const std::string base = slurp(argv[2]);
  const int iters = atoi(argv[3]);
  std::mt19937 rng(atoi(argv[4]));
  const char* toks[] = {"{", "}", "[", "]", ",", ":", "\"", "\\", "\\u", "\\u12", "null", "true", "1e999", "-", "0x10", "1.5", "\"id\"", "\"cameras\"", "\"origin\"", "[1,2]", "[1,2,3,4]", "{}", "[]", "\n", "\t", "e", "E+", ".", "99999999999999999999", "\xff"};
  int ok = 0, err = 0;
  const std::string tmp = "/tmp/s360_fuzz/cur_json_" + std::string(argv[4]);
  for (int it = 0; it < iters; ++it) {
    std::string d = base;
    int nm = 1 + rng() % 6;
    for (int m = 0; m < nm && !d.empty(); ++m) {
      size_t pos = rng() % d.size();
      switch (rng() % 7) {
        case 0: d[pos] = (char)rng(); break;
        case 1: d.resize(pos); break;
        case 2: d.insert(pos, toks[rng() % (sizeof toks / sizeof *toks)]); break;
        case 3: { size_t n = 1 + rng() % 24; d.erase(pos, n); break; }
        case 4: { size_t q = rng() % d.size(); size_t n = 1 + rng() % 40; d.insert(pos, d.substr(q, n)); break; }
        case 5: { // replace a number
          size_t a = d.find_first_of("0123456789", pos); if (a == std::string::npos) break; size_t b = d.find_first_not_of("0123456789.eE+-", a);
          const char* nums[] = {"0", "-1", "1e308", "-1e308", "1e-320", "nan", "inf", "2147483648", "-2147483649", "4294967296", "0.5", "65536", "1000000"};
          d.replace(a, (b == std::string::npos ? d.size() : b) - a, nums[rng() % 13]); break; }
        default: d[pos] ^= 1 << (rng() % 8);
      }
    }
    if (mode == "rig") {
      { std::ofstream o(tmp, std::ios::binary); o.write(d.data(), d.size()); }
      std::vector<s360_camera> cams(24);
      int n = s360_rig_load_json(tmp.c_str(), cams.data(), rng() % 2 ? 24 : 3);
      if (n >= 0) { ++ok; s360_rig_find_top(cams.data(), n < 24 ? n : 24); s360_rig_find_bottom(cams.data(), n < 24 ? n : 24); s360_rig_find_bottom2(cams.data(), n < 24 ? n : 24); } else ++err;
    } else if (mode == "isp") {
      s360_isp_config cfg;
      int r = s360_isp_config_from_json(d.c_str(), &cfg);
      if (r >= 0) {
        ++ok;
        std::vector<float> ccm(9), lut(3 * 4096), vv(37 * 3), vh(53 * 3);
        s360_isp_config_tables(&cfg, ccm.data(), lut.data(), 53, 37, vh.data(), vv.data());
      } else ++err;
    } else if (mode == "flow") {
      { std::ofstream o(tmp, std::ios::binary); o.write(d.data(), d.size()); }
      std::vector<float> out(1 << 16); int w = 0, h = 0;
      int r = s360_read_flow_from_file(tmp.c_str(), out.data(), &w, &h, rng() % 2 ? out.size() : 64);
      if (r >= 0) ++ok; else ++err;
    }
  }
  printf("%s: %d accepted, %d rejected\n", mode.c_str(), ok, err);
}
