// median_microbench — launch_median5_c2 (medianBlur(flow, 5), PixFlow.h:398,411) alone on batch-shaped inputs: ms per launch
// and medians per second, for tuning the tiled kernel (outputs per thread: S360_MEDIAN_T=8 / 16). Links surround360_amd/csrc/build/median.o
// (built by the library's Makefile) so that it times the object the library ships.
//   build: make -C tools median_microbench        run (GPU box): tools/median_microbench [reps]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../surround360_amd/csrc/flow_kernels.hpp"

int main(int argc, char** argv) {
  const int reps = argc > 1 ? std::atoi(argv[1]) : 20;
  struct Shape { int B, w, h; const char* what; };
  const Shape shapes[] = {{616, 303, 442, "22 slots x 28 side flows, a top level"}, {616, 152, 221, "the same, 2 octaves down"},
                          {616, 76, 111, "4 octaves down"}, {88, 1260, 263, "22 slots x 4 pole flows, a top level"}, {28, 303, 442, "one frame's side flows"}};
  for (const Shape& s : shapes) {
    const size_t bs = (size_t)s.w * s.h, n = bs * s.B;
    std::vector<float2> h(n);
    unsigned r = 12345u;
    for (size_t i = 0; i < n; ++i) {
      r = r * 1664525u + 1013904223u;
      const float a = (float)(r >> 8) * (1.0f / 16777216.0f) - 0.5f;
      r = r * 1664525u + 1013904223u;
      h[i] = make_float2(a * 20.0f, ((float)(r >> 8) * (1.0f / 16777216.0f) - 0.5f) * 6.0f);
    }
    float2 *src = nullptr, *dst = nullptr;
    if (hipMalloc(&src, n * sizeof(float2)) != hipSuccess || hipMalloc(&dst, n * sizeof(float2)) != hipSuccess) return 1;
    hipMemcpy(src, h.data(), n * sizeof(float2), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int k = 0; k < 3; ++k) s360::launch_median5_c2(nullptr, src, dst, s.w, s.h, bs, s.B);
    hipEventRecord(e0, nullptr);
    for (int k = 0; k < reps; ++k) s360::launch_median5_c2(nullptr, src, dst, s.w, s.h, bs, s.B);
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // checksum of the output (the same for every build that computes the same medians)
    hipMemcpy(h.data(), dst, n * sizeof(float2), hipMemcpyDeviceToHost);
    unsigned long long sum = 0;
    for (size_t i = 0; i < n; ++i) sum += (unsigned long long)__builtin_bit_cast(unsigned, h[i].x) * 31u + __builtin_bit_cast(unsigned, h[i].y);
    std::printf("B %4d  %4d x %4d  %-40s %8.3f ms per launch  %7.2f G medians/s  checksum %016llx\n", s.B, s.w, s.h, s.what, ms / reps,
                2.0 * n / (ms / reps) * 1e-6, sum);
    hipFree(src);
    hipFree(dst);
  }
  return 0;
}
