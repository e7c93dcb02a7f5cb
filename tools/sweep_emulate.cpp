// DEVELOPER / TEST TOOL — not part of the product: the PixFlow sweep kernels (surround360_amd/csrc/sweep_lock.hip,
// sweep_quad.hip) compiled for the CPU over tools/hip_wave_shim and run with wave semantics (see the shim's header),
// against a plain raster-order loop of the same recurrence (PixFlow.h:388-410) written here from errorFunction
// (sweep_common.hpp: error_from). Checks the kernels' indexing, their cross-lane / cross-wave / cross-workgroup
// hand-offs, the masked-pixel short cuts and the ticket / persistent-wave logic where no GPU is attached.
//   sweep_emulate <lock|quad> <w> <h> <flows> <seed> <mask: none|random|bands|rows0|most> <fast: 0|1> [rowflags: 0|1]
// Exit status 0 = bit-identical flows. Environment: EMU_LANE_ORDER=fwd|rev|shuffle, EMU_CUS, S360_QUAD_WAVES_PER_CU,
// S360_QUAD_WIN (the kernels' own tuning switches).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../surround360_amd/csrc/sweep_common.hpp"

using namespace s360;
namespace s360 { extern unsigned long long g_quad_rounds, g_quad_fallbacks, g_quad_chunks, g_quad_fills; }

namespace {
PixFlowConsts consts() {  // OpticalFlowFactory.h:26-41
  PixFlowConsts pc;
  pc.pyrScaleFactor = 0.9f; pc.smoothnessCoef = 0.001f; pc.verticalRegularizationCoef = 0.01f;
  pc.horizontalRegularizationCoef = 0.01f; pc.gradientStepSize = 0.5f; pc.downscaleFactor = 0.5f; pc.maxPercentage = 0;
  return pc;
}
// the raster-order sweep of one flow (dir > 0: forward, left / up neighbours; dir < 0: backward, right / down)
void reference_sweep(const float4* rec, const float2* G1, float2* flow, int w, int h, int dir, const SweepConst& c) {
  const float kEps = 0.001f;
  for (int yi = 0; yi < h; ++yi) {
    const int y = dir > 0 ? yi : h - 1 - yi;
    for (int xi = 0; xi < w; ++xi) {
      const int x = dir > 0 ? xi : w - 1 - xi;
      const float4 rc = rec[(size_t)y * w + x];
      if (rc.x != rc.x) continue;  // below the alpha threshold: not updated
      auto E = [&](float fx, float fy) {
        const Foot ft = footprint(w, (float)x + fx, (float)y + fy, c);
        Texels t;
        const float2 a = G1[ft.off], b = G1[ft.off + 1], cc = G1[ft.off + w], d = G1[ft.off + w + 1];
        t.r0 = make_float4(a.x, a.y, b.x, b.y);
        t.r1 = make_float4(cc.x, cc.y, d.x, d.y);
        return error_from(t, ft, rc.x, rc.y, rc.z, rc.w, fx, fy, c);
      };
      float2 f = flow[(size_t)y * w + x];
      float cur = E(f.x, f.y);
      if (xi > 0) {
        const float2 p = flow[(size_t)y * w + x - dir];
        const float e = E(p.x, p.y);
        if (e < cur) { f = p; cur = e; }
      }
      if (yi > 0) {
        const float2 p = flow[(size_t)(y - dir) * w + x];
        const float e = E(p.x, p.y);
        if (e < cur) { f = p; cur = e; }
      }
      const float ex = E(f.x + kEps, f.y), ey = E(f.x, f.y + kEps);
      float2 r;
      r.x = f.x - c.gradStep * ((ex - cur) / kEps);
      r.y = f.y - c.gradStep * ((ey - cur) / kEps);
      flow[(size_t)y * w + x] = r;
    }
  }
}
// the check the library runs on the device before it uses the fast division (k_verify_div), here on the host
bool divisor_ok(float cc) {
  const float rc = 1.0f / cc;
  for (unsigned m = 0; m < (1u << 23); ++m) {
    const float x = __uint_as_float(0x3f800000u | m);
    if (x / cc != fdiv_m(x, cc, rc) || (-x) / cc != fdiv_m(-x, cc, rc)) return false;
  }
  return true;
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 8) {
    std::fprintf(stderr, "usage: %s <lock|quad> <w> <h> <flows> <seed> <mask> <fast> [rowflags]\n", argv[0]);
    return 2;
  }
  const std::string kernel = argv[1], mask = argv[6];
  const int w = std::atoi(argv[2]), h = std::atoi(argv[3]), B = std::atoi(argv[4]), seed = std::atoi(argv[5]);
  bool fast = std::atoi(argv[7]) != 0;
  const bool useRowflags = argc > 8 && std::atoi(argv[8]) != 0;
  const size_t bs = (size_t)w * h + 8;  // plane stride in pixels
  const PixFlowConsts pc = consts();
  const SweepConst c = make_sweep_const(pc, w, h);
  if (fast && !(divisor_ok(c.fcols) && divisor_ok(c.frows) && divisor_ok(0.001f))) {
    std::printf("fast division not proven for %d x %d: running the IEEE path\n", w, h);
    fast = false;
  }
  std::mt19937 rng((unsigned)seed);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  const int nimg = B + 1;
  std::vector<float2> G(bs * nimg);
  for (auto& g : G) g = make_float2(0.2f * U(rng), 0.2f * U(rng));
  std::vector<float4> rec(bs * B);
  std::vector<float2> flow(bs * B), want;
  std::vector<int> i0(B), i1(B);
  std::vector<unsigned> rowflags((size_t)B * h, 0xFFFFFFFFu);
  const float kNaN = __uint_as_float(0x7fc00000u);
  for (int b = 0; b < B; ++b) {
    i0[b] = b;
    i1[b] = (b + 1) % nimg;
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) {
        float4 r = make_float4(0.2f * U(rng), 0.2f * U(rng), 3.f * U(rng), 3.f * U(rng));
        float2 f = make_float2(r.z + 2.f * U(rng), r.w + 2.f * U(rng));
        const unsigned k = rng();
        if (mask == "smooth") {  // what the pipeline feeds the sweeps: a smooth disparity field with sub-pixel roughness
          const float sx = 9.f * std::sin(0.021f * x + 0.013f * y + b) + 3.f, sy = 1.5f * std::cos(0.017f * x - 0.011f * y);
          r.z = sx; r.w = sy;
          f = make_float2(sx + 0.3f * U(rng), sy + 0.3f * U(rng));
        } else
        if (k % 97 == 0) { f = make_float2(0.f, 0.f); r.z = 1e-16f; r.w = 0.f; }  // operands below the fast path's range
        if (mask != "smooth" && k % 89 == 0) f = make_float2(r.z, r.w);                              // smoothness term exactly zero
        if (mask != "smooth" && k % 83 == 0) f = make_float2(40.f * U(rng), 40.f * U(rng));          // samples clamped at the borders
        bool masked = false;
        if (mask == "random") masked = (k >> 8) % 10 < 3;
        else if (mask == "bands") masked = y < (h * 5) / 8 ? true : (k >> 8) % 10 < 2;  // like a pole flow: the upper rows have no data
        else if (mask == "rows0") masked = (y % 16 == 0 && x > w / 3 && x < 2 * w / 3) || (k >> 8) % 10 < 1;
        else if (mask == "most") masked = (k >> 8) % 100 < 97;
        if (b == 1 && mask == "bands") masked = (y >= h / 4 && y < h / 2) || (k >> 8) % 10 < 2;
        if (masked) r.x = kNaN;
        else rowflags[(size_t)b * h + y] = 0u;
        rec[bs * b + (size_t)y * w + x] = r;
        flow[bs * b + (size_t)y * w + x] = f;
      }
  }
  // What the kernels read: per flow the half-records {blurredFlow.x | NaN = not updated, blurredFlow.y}; I0's gradient is the
  // gradient plane of image i0[b] = b (the same planes the OTHER flows sample as their I1) — the records' .x .y take it
  std::vector<float2> half(bs * B);
  for (int b = 0; b < B; ++b)
    for (size_t i = 0; i < (size_t)w * h; ++i) {
      float4& r = rec[bs * b + i];
      const float2 g0 = G[bs * i0[b] + i];
      const bool masked = r.x != r.x;
      r.x = masked ? kNaN : g0.x;
      r.y = g0.y;
      half[bs * b + i] = make_float2(masked ? kNaN : r.z, r.w);
    }
  for (int dir : {1, -1}) {
    want = flow;
    for (int b = 0; b < B; ++b) reference_sweep(rec.data() + bs * b, G.data() + bs * i1[b], want.data() + bs * b, w, h, dir, c);
    std::vector<float2> got = flow;
    unsigned errflag = 0;
    FlowIdx idx{i0.data(), i1.data()};
    if (kernel == "lock") {
      std::vector<unsigned char> handoff(sweep_lock_handoff_bytes(w, h, B, sweep_lock_waves()), 0xFF);
      launch_sweep_lock(nullptr, rec.data(), G.data(), got.data(), handoff.data(), &errflag, w, h, bs, B, idx, dir, pc, fast);
    } else {
      std::vector<unsigned char> handoff(sweep_quad_handoff_bytes(w, h, B), 0xFF);
      launch_sweep_quad(nullptr, half.data(), G.data(), got.data(), handoff.data(), &errflag, w, h, bs, B, idx, dir, pc, fast,
                        useRowflags ? rowflags.data() : nullptr);
    }
    size_t bad = 0, changed = 0;
    for (size_t i = 0; i < got.size(); ++i) {
      if (std::memcmp(&got[i], &want[i], sizeof(float2)) != 0) {
        if (bad < 5) std::printf("  dir %d flow %zu pixel (%zu,%zu): got (%g,%g) want (%g,%g)\n", dir, i / bs, (i % bs) % w, (i % bs) / w,
                                 got[i].x, got[i].y, want[i].x, want[i].y);
        ++bad;
      }
      changed += std::memcmp(&flow[i], &want[i], sizeof(float2)) != 0;
    }
    std::printf("%s %dx%d flows %d dir %+d mask %s fast %d: %zu pixels updated, %zu differ, error flag %u\n", kernel.c_str(), w, h, B, dir,
                mask.c_str(), (int)fast, changed, bad, errflag);
    if (bad || errflag) return 1;
    if (kernel == "quad" && g_quad_rounds)
      std::printf("quad window: %llu rounds, fallback fraction %.4f , %llu chunks, %llu fills\n", g_quad_rounds / 64,
                  (double)g_quad_fallbacks / (double)g_quad_rounds, g_quad_chunks / 64, g_quad_fills / 64);
    flow = want;  // the backward sweep continues from the forward sweep's result, as in the pipeline
  }
  return 0;
}
