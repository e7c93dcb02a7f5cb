// sweep_microbench.hip — timing harness for the sweep kernels on synthetic buffers (results are not checked here;
// parity lives in tests/). Build: make -C tools   Run on the GPU box: tools/sweep_microbench
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <string>
#include <cstdlib>
#include <random>
#include <vector>

#include "../surround360_amd/csrc/flow_kernels.hip"
#include "../surround360_amd/csrc/sweep_lock.hip"
#include "../surround360_amd/csrc/sweep_quad.hip"

using namespace s360;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static unsigned long long g_ts[8];
static FlowIdx make_idx(int B) {  // flow b: image b against image (b + B) mod 2B (device arrays, leaked: a tool)
  std::vector<int> h(2 * B);
  for (int b = 0; b < B; ++b) { h[b] = b % (2 * B); h[B + b] = (b + B) % (2 * B); }
  int* d = nullptr;
  CK(hipMalloc(&d, h.size() * sizeof(int)));
  CK(hipMemcpy(d, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice));
  return FlowIdx{d, d + B};
}
// what the kernels read per flow and pixel: {blurredFlow.x | NaN = not updated, blurredFlow.y}; hrec keeps round 3's four floats
// {I0x | NaN, I0y, blurredFlow} for the generator's sake (I0's gradient now comes from the gradient plane of image i0[b])
static std::vector<float> half_records(const std::vector<float>& hrec) {
  std::vector<float> hh(hrec.size() / 2);
  for (size_t i = 0; i < hrec.size() / 4; ++i) {
    const bool masked = hrec[4 * i] != hrec[4 * i];
    hh[2 * i] = masked ? __builtin_nanf("") : hrec[4 * i + 2];
    hh[2 * i + 1] = hrec[4 * i + 3];
  }
  return hh;
}
static float run(int w, int h, int B, bool fast, int mode /*2 lock, 3 quad*/, int reps) {
  const size_t n = (size_t)w * h;
  std::mt19937 rng(1234);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  std::vector<float> hG(2 * B * n * 2), hrec(B * n * 4), hflow(B * n * 2);
  for (auto& v : hG) v = 0.05f * U(rng);
  for (size_t i = 0; i < (size_t)B * n; ++i) {
    hrec[4 * i + 0] = 0.05f * U(rng);
    hrec[4 * i + 1] = 0.05f * U(rng);
    hrec[4 * i + 2] = 2.0f * U(rng);
    hrec[4 * i + 3] = 1.0f * U(rng);
    hflow[2 * i + 0] = hrec[4 * i + 2] + 0.3f * U(rng);
    hflow[2 * i + 1] = hrec[4 * i + 3] + 0.3f * U(rng);
  }
  if (const char* e = getenv("S360_MB_MASKROWS")) {  // the first fraction of the rows is below the alpha threshold
    const int rows = (int)(atof(e) * h);
    for (int b = 0; b < B; ++b)
      for (int y = 0; y < rows; ++y)
        for (int x = 0; x < w; ++x) hrec[4 * ((size_t)b * n + (size_t)y * w + x)] = __builtin_nanf("");
  }
  float *dG, *drec, *dflow;
  void* hand;
  unsigned* err;
  CK(hipMalloc(&dG, hG.size() * 4));
  const std::vector<float> hhalf = half_records(hrec);
  float* dhalf;
  CK(hipMalloc(&drec, hrec.size() * 4));
  CK(hipMalloc(&dhalf, hhalf.size() * 4));
  CK(hipMalloc(&dflow, hflow.size() * 4));
  const size_t hb = std::max(sweep_lock_handoff_bytes(w, h, B, sweep_lock_waves()), sweep_quad_handoff_bytes(w, h, B));
  CK(hipMalloc(&hand, hb));
  CK(hipMalloc(&err, 8));
  CK(hipMemset(err, 0, 8));
  CK(hipMemcpy(dG, hG.data(), hG.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(drec, hrec.data(), hrec.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dhalf, hhalf.data(), hhalf.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dflow, hflow.data(), hflow.size() * 4, hipMemcpyHostToDevice));
  FlowIdx idx = make_idx(B);
  PixFlowConsts pc{0.9f, 0.001f, 0.01f, 0.01f, 0.5f, 0.5f, 0};
  hipStream_t st;
  CK(hipStreamCreate(&st));
  if (fast) {
    std::vector<float> d{0.001f, (float)w, (float)h};
    if (!sweep_verify_divisors(st, d)) printf("  (fast division not verified for %d x %d)\n", w, h);
  }
  hipEvent_t a, b2;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b2));
  auto once = [&](int dir) {
    CK(hipMemsetAsync(hand, 0xFF, hb, st));  // (the library resets one arena per flow call instead)
    if (mode == 2)
      launch_sweep_lock(st, (const float4*)drec, (const float2*)dG, (float2*)dflow, hand, err, w, h, n, B, idx, dir, pc, fast);
    else
      launch_sweep_quad(st, (const float2*)dhalf, (const float2*)dG, (float2*)dflow, hand, err, w, h, n, B, idx, dir, pc, fast);
  };
  once(1);
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(a, st));
  for (int r = 0; r < reps; ++r) once(r & 1 ? -1 : 1);
  CK(hipEventRecord(b2, st));
  CK(hipEventSynchronize(b2));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b2));
  unsigned e = 0;
  CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
  if (e) printf("  (errflag set)\n");
  CK(hipMemcpy(g_ts, (char*)hand + 32, sizeof(g_ts), hipMemcpyDeviceToHost));
  hipFree(dG); hipFree(drec); hipFree(dflow); hipFree(hand); hipFree(err);
  hipStreamDestroy(st);
  return ms * 1000.f / reps;
}

// Saturated throughput: the same sweep on NS independent buffer sets / streams at once; returns Gpx/s.
static float throughput(int w, int h, int B, int mode, int NS, int reps) {
  const size_t n = (size_t)w * h;
  std::mt19937 rng(99);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  std::vector<float> hG(2 * B * n * 2), hrec(B * n * 4), hflow(B * n * 2);
  for (auto& v : hG) v = 0.05f * U(rng);
  for (size_t i = 0; i < (size_t)B * n; ++i) {
    hrec[4 * i + 0] = 0.05f * U(rng); hrec[4 * i + 1] = 0.05f * U(rng);
    hrec[4 * i + 2] = 2.0f * U(rng); hrec[4 * i + 3] = 1.0f * U(rng);
    hflow[2 * i + 0] = hrec[4 * i + 2] + 0.3f * U(rng); hflow[2 * i + 1] = hrec[4 * i + 3] + 0.3f * U(rng);
  }
  if (const char* e = getenv("S360_MB_MASKROWS")) {  // the first fraction of the rows is below the alpha threshold (like the pole flows)
    const int rows = (int)(atof(e) * h);
    for (int b = 0; b < B; ++b)
      for (int y = 0; y < rows; ++y)
        for (int x = 0; x < w; ++x) hrec[4 * ((size_t)b * n + (size_t)y * w + x)] = __builtin_nanf("");
  }
  std::vector<float*> dG(NS), drec(NS), dhalf(NS), dflow(NS), dA(NS), dbl(NS);
  std::vector<void*> hand(NS);
  std::vector<unsigned*> err(NS);
  std::vector<hipStream_t> st(NS);
  const size_t hb = std::max(sweep_lock_handoff_bytes(w, h, B, sweep_lock_waves()), sweep_quad_handoff_bytes(w, h, B));
  for (int k = 0; k < NS; ++k) {
    CK(hipMalloc(&dG[k], hG.size() * 4)); CK(hipMalloc(&drec[k], hrec.size() * 4)); CK(hipMalloc(&dhalf[k], hrec.size() * 2)); CK(hipMalloc(&dflow[k], hflow.size() * 4));
    CK(hipMalloc(&hand[k], hb)); CK(hipMalloc(&err[k], 8)); CK(hipMemset(err[k], 0, 8));
    CK(hipMalloc(&dA[k], 2 * B * n * 4)); CK(hipMalloc(&dbl[k], B * n * 8));
    {
      std::vector<float> ones(2 * B * n, 1.0f), bl(B * n * 2);
      for (size_t i = 0; i < (size_t)B * n; ++i) { bl[2 * i] = hrec[4 * i + 2]; bl[2 * i + 1] = hrec[4 * i + 3]; }
      CK(hipMemcpy(dA[k], ones.data(), ones.size() * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(dbl[k], bl.data(), bl.size() * 4, hipMemcpyHostToDevice));
    }
    CK(hipMemcpy(dG[k], hG.data(), hG.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(drec[k], hrec.data(), hrec.size() * 4, hipMemcpyHostToDevice));
    { const std::vector<float> hhalf = half_records(hrec); CK(hipMemcpy(dhalf[k], hhalf.data(), hhalf.size() * 4, hipMemcpyHostToDevice)); }
    CK(hipMemcpy(dflow[k], hflow.data(), hflow.size() * 4, hipMemcpyHostToDevice));
    CK(hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking));
  }
  FlowIdx idx = make_idx(B);
  PixFlowConsts pc{0.9f, 0.001f, 0.01f, 0.01f, 0.5f, 0.5f, 0};
  std::vector<float> d{0.001f, (float)w, (float)h};
  sweep_verify_divisors(st[0], d);
  auto once = [&](int k, int dir) {
    CK(hipMemsetAsync(hand[k], 0xFF, hb, st[k]));
    if (mode == 3)
      launch_sweep_quad(st[k], (const float2*)dhalf[k], (const float2*)dG[k], (float2*)dflow[k], hand[k], err[k], w, h, n, B, idx, dir, pc, true);
    else
      launch_sweep_lock(st[k], (const float4*)drec[k], (const float2*)dG[k], (float2*)dflow[k], hand[k], err[k], w, h, n, B, idx, dir, pc, true);
  };
  for (int k = 0; k < NS; ++k) once(k, 1);
  CK(hipDeviceSynchronize());
  const auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r)
    for (int k = 0; k < NS; ++k) once(k, r & 1 ? -1 : 1);
  CK(hipDeviceSynchronize());
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  for (int k = 0; k < NS; ++k) { hipFree(dG[k]); hipFree(drec[k]); hipFree(dflow[k]); hipFree(hand[k]); hipFree(err[k]); hipFree(dA[k]); hipFree(dbl[k]); hipStreamDestroy(st[k]); }
  return (float)((double)NS * reps * B * n / sec / 1e9);
}


// Where do 1-wave workgroups land? Every workgroup records HW_ID / XCC_ID and then idles ~spin shader cycles so that
// the workgroups of all concurrent dispatches are resident together.
__global__ __launch_bounds__(256) void k_place(unsigned* out, long long spin) {
  unsigned a, b;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(a));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(b));
  const long long t0 = __builtin_readcyclecounter();
  while ((long long)__builtin_readcyclecounter() - t0 < spin) __builtin_amdgcn_s_sleep(8);
  if ((threadIdx.x & 63) == 0) {
    const unsigned i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    out[2 * i] = a; out[2 * i + 1] = b;
  }
}
static void placement(int G0, int NS, int wavesPerWg) {
  const int G = G0 * wavesPerWg;  // waves per dispatch
  std::vector<unsigned*> d(NS);
  std::vector<hipStream_t> st(NS);
  for (int k = 0; k < NS; ++k) { CK(hipMalloc(&d[k], G * 8)); CK(hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking)); }
  for (int k = 0; k < NS; ++k) hipLaunchKernelGGL(k_place, dim3(G0), dim3(64 * wavesPerWg), 0, st[k], d[k], 2000000LL);
  CK(hipDeviceSynchronize());
  const auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < NS; ++k) hipLaunchKernelGGL(k_place, dim3(G0), dim3(64 * wavesPerWg), 0, st[k], d[k], 2000000LL);
  CK(hipDeviceSynchronize());
  printf("wall time of %d concurrent spin dispatches (2M cycles each): %.3f ms\n", NS,
         1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  std::vector<int> perSimd(8 * 128 * 4, 0), perCu(8 * 128, 0), perXcc(8, 0);
  std::vector<unsigned> h(2 * G);
  for (int k = 0; k < NS; ++k) {
    CK(hipMemcpy(h.data(), d[k], G * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < G; ++i) {
      const unsigned id = h[2 * i], xcc = h[2 * i + 1] & 7;
      const unsigned simd = (id >> 4) & 3, cu = (id >> 8) & 15, sh = (id >> 12) & 1, se = (id >> 13) & 7;
      const unsigned cuIdx = ((se & 3) * 2 + sh) * 16 + cu;  // 0..127
      perXcc[xcc]++; perCu[xcc * 128 + cuIdx]++; perSimd[(xcc * 128 + cuIdx) * 4 + simd]++;
    }
  }
  auto stats = [](const std::vector<int>& v, const char* name) {
    int used = 0, mx = 0; long tot = 0;
    std::vector<int> hist(64, 0);
    for (int x : v) { if (x) ++used; mx = std::max(mx, x); tot += x; hist[std::min(x, 63)]++; }
    printf("%-8s slots used %d, max %d, total %ld; histogram(count:slots):", name, used, mx, tot);
    for (int i = 0; i < 64; ++i) if (hist[i] && i) printf(" %d:%d", i, hist[i]);
    printf("\n");
  };
  printf("placement of %d dispatches x %d workgroups of %d wave(s)\n", NS, G0, wavesPerWg);
  printf("per XCC:"); for (int x : perXcc) printf(" %d", x); printf("\n");
  stats(perCu, "per CU"); stats(perSimd, "per SIMD");
}

int main(int argc, char** argv) {
  if (argc > 3 && std::string(argv[1]) == "place") { placement(atoi(argv[2]), atoi(argv[3]), argc > 4 ? atoi(argv[4]) : 1); return 0; }
  if (argc > 5 && std::string(argv[1]) == "tp1") {  // one configuration (for rocprofv3 --pmc runs): w h B streams
    const int w = atoi(argv[2]), h = atoi(argv[3]), B = atoi(argv[4]), ns = atoi(argv[5]), md = argc > 6 ? atoi(argv[6]) : 2;
    printf("mode %d %dx%d B=%d streams=%d: %.2f Gpx/s\n", md, w, h, B, ns, throughput(w, h, B, md, ns, 4));
    return 0;
  }
  if (argc > 4 && std::string(argv[1]) == "ts") {  // sweep_microbench_ts only: cycle phases of compute wave 0 of ticket 0 (lock kernel)
    const int w = atoi(argv[2]), h = atoi(argv[3]), B = atoi(argv[4]);
    const float us = run(w, h, B, true, 2, 1);
    const char* names[7] = {"loop top", "candidates + addresses + gather issue", "barrier", "gather wait (after the LDS preload)", "errorFunction", "selection + gradient step + LDS store", "LDS preload of the next step"};
    unsigned long long tot = 0;
    for (int i = 0; i < 7; ++i) tot += g_ts[i];
    printf("lock %dx%d B=%d: %.1f us per launch; s_memtime ticks of wave 0 / ticket 0 over %d steps (last launch)\n", w, h, B, us, w + 3);
    for (int i = 0; i < 7; ++i) printf("  %-40s %10llu ticks  %5.1f %%  %.1f per step\n", names[i], g_ts[i], 100.0 * g_ts[i] / (double)tot, g_ts[i] / (double)(w + 3));
    printf("  total %llu ticks = %.1f per step\n", tot, tot / (double)(w + 3));
    printf("  bulk service wave: %llu ticks inside events = %.1f per event (4 events per 16 steps)\n", g_ts[7], g_ts[7] / ((w + 3) / 4.0));
    return 0;
  }
  if (argc > 1 && std::string(argv[1]) == "tp") {
    printf("saturated sweep throughput, Gpx/s (one px = one pixel update of one sweep)\n");
    struct C2 { int w, h, B; const char* name; };
    const C2 cs[] = {{5040, 1052, 4, "polar L0"}, {607, 884, 28, "side L0"}, {1153, 240, 4, "polar L14"}, {140, 203, 28, "side L14"}};
    for (const C2& c : cs)
      for (int ns : {1, 4, 16}) {
        printf("%-10s B=%2d streams=%d : lock %7.2f   quad %7.2f\n", c.name, c.B, ns,
               throughput(c.w, c.h, c.B, 2, ns, 4), throughput(c.w, c.h, c.B, 3, ns, 4));
        fflush(stdout);
      }
    return 0;
  }
  struct Cfg { int w, h, B; const char* name; };
  const Cfg cfgs[] = {{127, 27, 4, "polar L35"}, {613, 128, 4, "polar L20"}, {5040, 1052, 4, "polar L0"},
                      {27, 38, 28, "side L30"}, {140, 203, 28, "side L14"}, {607, 884, 28, "side L0"}};
  printf("%-10s %6s %6s %3s | %9s | %9s | %9s |  us per launch of one flow batch alone\n", "config", "w", "h", "B", "lock", "lock ieee", "quad");
  for (const Cfg& c : cfgs) {
    const int reps = c.w > 1000 ? 6 : 20;
    printf("%-10s %6d %6d %3d | %9.1f | %9.1f | %9.1f |  lock us/step %.3f\n", c.name, c.w, c.h, c.B,
           run(c.w, c.h, c.B, true, 2, reps), run(c.w, c.h, c.B, false, 2, reps), run(c.w, c.h, c.B, true, 3, reps),
           run(c.w, c.h, c.B, true, 2, reps) / (c.w + 18));
    fflush(stdout);
  }
  return 0;
}
