// writer -> reader round trips of host/png_io.hpp at random sizes (1 pixel to several deflate bands), 3 / 4 channels, 16-bit,
// 1..5 encoder threads; run under the sanitizers (make -C tools fuzz/png_roundtrip && mkdir -p /tmp/s360_fuzz && tools/fuzz/png_roundtrip)
#include <cstdio>
#include <random>
#include "png_io.hpp"
int main() {
  std::mt19937 rng(5); int bad = 0, n = 0;
  for (int it = 0; it < 400; ++it) {
    int w = 1 + rng() % (it % 7 == 0 ? 3000 : 70), h = 1 + rng() % (it % 5 == 0 ? 1200 : 50), c = (rng() & 1) ? 3 : 4, th = 1 + rng() % 5;
    std::vector<uint8_t> px((size_t)w * h * c);
    for (auto& v : px) v = (it & 1) ? (uint8_t)rng() : (uint8_t)((&v - px.data()) / 7);
    pngio::write("/tmp/s360_fuzz/rt.png", px.data(), w, h, c, 1, th);
    pngio::Image im = pngio::read("/tmp/s360_fuzz/rt.png", true);
    ++n; if (im.w != w || im.h != h || im.c != c || !(im.px.size() == px.size() && std::equal(px.begin(), px.end(), im.px.begin()))) { ++bad; printf("MISMATCH %d x %d x %d threads %d\n", w, h, c, th); }
    if (it % 9 == 0) {  // 16-bit RGB
      std::vector<uint16_t> p16((size_t)w * h * 3); for (auto& v : p16) v = (uint16_t)rng();
      pngio::write16("/tmp/s360_fuzz/rt16.png", p16.data(), w, h, 1, th);
      pngio::Image i8 = pngio::read("/tmp/s360_fuzz/rt16.png", false);  // high bytes
      bool ok = i8.w == w && i8.h == h; for (size_t i = 0; ok && i < p16.size(); ++i) ok = i8.px[i] == (p16[i] >> 8);
      ++n; if (!ok) { ++bad; printf("MISMATCH16 %d x %d\n", w, h); }
    }
  }
  printf("%d round trips, %d mismatches\n", n, bad);
}
