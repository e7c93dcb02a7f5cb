#include <cstdio>
#include <dirent.h>
#include <string>
#include "png_io.hpp"
#include "jpeg_io.hpp"
int main(int argc, char** argv) {
  DIR* d = opendir(argv[1]); int ok = 0, err = 0; dirent* e;
  while ((e = readdir(d))) {
    std::string n = e->d_name; if (n[0] == '.') continue;
    for (int ka = 0; ka < 2; ++ka) {
      try { pngio::Image im = jpegio::read_any(std::string(argv[1]) + "/" + n, ka); ++ok; } catch (const std::exception& x) { ++err; }
    }
    try { int w, h, dp; auto g = pngio::read_gray(std::string(argv[1]) + "/" + n, &w, &h, &dp); ++ok; } catch (const std::exception& x) { ++err; }
  }
  printf("%d decoded, %d rejected\n", ok, err);
}
