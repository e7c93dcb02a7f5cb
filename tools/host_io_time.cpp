// host_io_time — what the end-to-end stream's host side costs on THIS machine (DESIGN.md section 6 `end_to_end_files`, section 9):
// the write rate of the scratch directory against /dev/shm, PNG encode (parallel deflate, filtered scanlines) and decode
// of an 8192 x 8192 x 3 image of the synthetic world's kind of content. Ten seconds; run it first on a GPU box before
// reading the stream's numbers:  make -C tools host_io_time && tools/host_io_time [scratch dir, default /tmp]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <string>
#include <thread>
#include <vector>
#include <dlfcn.h>
#include "../host/png_io.hpp"
using clk = std::chrono::steady_clock;
static double ms(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }
int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : "/tmp";
  const int w = 8192, h = 8192;
  std::vector<uint8_t> px((size_t)w * h * 3);
  unsigned s = 12345;  // smooth structure + a few bits of noise, like a rendered frame
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x)
      for (int c = 0; c < 3; ++c) {
        s = s * 1664525u + 1013904223u;
        px[((size_t)y * w + x) * 3 + c] = (uint8_t)(128 + 90 * std::sin(x * 0.003 + c) * std::cos(y * 0.002 - c) + ((s >> 28) & 7));
      }
  std::printf("hardware threads: %u\n", std::thread::hardware_concurrency());
  for (const std::string& d : {dir, std::string("/dev/shm")}) {
    const std::string f = d + "/s360_io_time.bin";
    auto t0 = clk::now();
    FILE* fp = std::fopen(f.c_str(), "wb");
    if (!fp) { std::printf("%s: cannot write\n", d.c_str()); continue; }
    std::fwrite(px.data(), 1, (size_t)82 << 20, fp);
    std::fclose(fp);
    auto t1 = clk::now();
    std::printf("fwrite + fclose of 82 MB to %-9s %.0f ms (%.0f MB/s)\n", (d + ":").c_str(), ms(t0, t1), 82.0 * 1.048576 / (ms(t0, t1) / 1e3));
    std::remove(f.c_str());
  }
  for (int strat : {Z_DEFAULT_STRATEGY, Z_RLE, Z_HUFFMAN_ONLY})
    for (int filt : {2, 1}) {
      pngio::g_write_strategy = strat;
      pngio::g_write_filter = filt;
      const std::string f = "/dev/shm/s360_io_time_v.png";
      auto t0 = clk::now();
      pngio::write(f, px.data(), w, h, 3, 1, 4);
      auto t1 = clk::now();
      FILE* fp = std::fopen(f.c_str(), "rb"); std::fseek(fp, 0, SEEK_END); const long n = std::ftell(fp); std::fclose(fp);
      pngio::Image im = pngio::read(f, false);
      std::printf("strategy %d filter %s, 4 threads: %.0f ms, %.1f MB%s\n", strat, filt == 1 ? "Sub" : "Up", ms(t0, t1), n / 1048576.0,
                  (im.px.size() == px.size() && std::equal(px.begin(), px.end(), im.px.begin())) ? "" : "  MISMATCH");
      std::remove(f.c_str());
    }
  pngio::g_write_strategy = Z_RLE;
  pngio::g_write_filter = 1;
  for (const std::string& d : {dir, std::string("/dev/shm")}) {
    const std::string f = d + "/s360_io_time.png";
    for (int threads : {0, 16, 4}) {
      auto t0 = clk::now();
      try { pngio::write(f, px.data(), w, h, 3, 1, threads); } catch (const std::exception& e) { std::printf("%s\n", e.what()); break; }
      auto t1 = clk::now();
      FILE* fp = std::fopen(f.c_str(), "rb"); std::fseek(fp, 0, SEEK_END); const long n = std::ftell(fp); std::fclose(fp);
      std::printf("PNG encode + write to %-9s threads %-3d %.0f ms, %.1f MB\n", (d + ":").c_str(), threads, ms(t0, t1), n / 1048576.0);
    }
    auto t0 = clk::now();
    pngio::Image im = pngio::read(f, false);
    auto t1 = clk::now();
    std::printf("PNG decode from %-9s %.0f ms (%d x %d)%s\n", (d + ":").c_str(), ms(t0, t1), im.w, im.h, (im.px.size() == px.size() && std::equal(px.begin(), px.end(), im.px.begin())) ? "" : "  MISMATCH");
    std::remove(f.c_str());
  }
  // page-locked memory (s360_host_alloc) against the heap as the place decoded pixels go: a camera image's decode writes and
  // re-reads its rows (unfilter), 17 decoders at once
  void* lib = dlopen(argc > 2 ? argv[2] : "surround360_amd/libs360.so", RTLD_NOW);
  if (lib) {
    auto alloc = (void* (*)(size_t))dlsym(lib, "s360_host_alloc");
    auto release = (void (*)(void*))dlsym(lib, "s360_host_free");
    const std::string f = "/dev/shm/s360_io_cam.png";
    pngio::write(f, px.data(), 2048, 2048, 3, 1, 0);
    void* probe = alloc ? alloc(4096) : nullptr;  // (no HIP device: only the heap is timed)
    if (probe) release(probe);
    for (int pinned = 0; pinned < (probe ? 2 : 1) && alloc; ++pinned) {
      pngio::g_pixel_alloc = pinned ? alloc : nullptr;
      pngio::g_pixel_free = pinned ? release : nullptr;
      {
        std::vector<pngio::Image> im(17);
        for (int rep = 0; rep < 2; ++rep) {  // (the second pass decodes into the buffers of the first: what a stream does)
          auto t0 = clk::now();
          std::vector<std::thread> th;
          for (int k = 0; k < 17; ++k) th.emplace_back([&, k] { pngio::read_into(f, false, im[k]); });
          for (auto& t : th) t.join();
          auto t1 = clk::now();
          std::printf("17 x 2048^2 PNG decode, 17 threads, %s memory, pass %d: %.0f ms\n", pinned ? "page-locked" : "heap", rep, ms(t0, t1));
        }
        pngio::Pixels a((size_t)64 << 20), b((size_t)64 << 20);
        auto t0 = clk::now();
        std::memcpy(a.data(), px.data(), a.size());
        auto t1 = clk::now();
        std::memcpy(b.data(), a.data(), a.size());
        auto t2 = clk::now();
        std::printf("memcpy 64 MB heap -> %s: %.1f GB/s; %s -> %s: %.1f GB/s\n", pinned ? "page-locked" : "heap", 0.0671 / (ms(t0, t1) / 1e3),
                    pinned ? "page-locked" : "heap", pinned ? "page-locked" : "heap", 0.0671 / (ms(t1, t2) / 1e3));
      }
    }
    pngio::g_pixel_alloc = nullptr;
    pngio::g_pixel_free = nullptr;
    std::remove(f.c_str());
  } else {
    std::printf("(libs360.so not found: page-locked decode not timed)\n");
  }
}
