// sanitizer fuzz harness (not part of the repo's product): mutates valid PNG / JPEG files and decodes them
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <random>
#include <string>
#include <vector>
#include "png_io.hpp"
#include "jpeg_io.hpp"
int main(int argc, char** argv) {
  const std::string seedfile = argv[1];
  const int iters = atoi(argv[2]);
  const unsigned seed = argc > 3 ? atoi(argv[3]) : 1;
  std::ifstream f(seedfile, std::ios::binary);
  std::vector<unsigned char> base((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  std::mt19937 rng(seed);
  int ok = 0, err = 0;
  const std::string tmp = "/tmp/s360_fuzz/cur_" + std::to_string(seed) + (seedfile.find(".jp") != std::string::npos ? ".jpg" : ".png");
  for (int it = 0; it < iters; ++it) {
    std::vector<unsigned char> d = base;
    int nm = 1 + rng() % 8;
    for (int m = 0; m < nm; ++m) {
      int kind = rng() % 5;
      if (d.empty()) break;
      size_t pos = rng() % d.size();
      if (kind == 0) d[pos] = rng();
      else if (kind == 1) d[pos] ^= 1u << (rng() % 8);
      else if (kind == 2) d.resize(pos);  // truncate
      else if (kind == 3) { size_t n = 1 + rng() % 16; d.insert(d.begin() + pos, n, (unsigned char)rng()); }
      else { size_t n = 1 + rng() % 16; if (pos + n < d.size()) d.erase(d.begin() + pos, d.begin() + pos + n); }
    }
    { std::ofstream o(tmp, std::ios::binary); o.write((const char*)d.data(), d.size()); }
    try {
      pngio::Image im = jpegio::read_any(tmp, (it & 1) != 0);
      (void)im; ++ok;
    } catch (const std::exception& e) { ++err; }
  }
  printf("%s: %d decoded, %d rejected\n", seedfile.c_str(), ok, err);
}
