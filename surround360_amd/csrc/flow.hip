// flow.hip — FlowEngine: PixFlow::computeOpticalFlow (PixFlow.h:81-183) as a batched HIP
// launch sequence on one stream. See flow.hpp.
#include "flow.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace s360 {

PixFlowConsts pixflow_consts_by_name(const std::string& name) {
  PixFlowConsts c;
  c.pyrScaleFactor = 0.9f;
  c.smoothnessCoef = 0.001f;
  c.verticalRegularizationCoef = 0.01f;
  c.horizontalRegularizationCoef = 0.01f;
  c.gradientStepSize = 0.5f;
  c.downscaleFactor = 0.5f;
  c.maxPercentage = 0;
  if (name == "pixflow_low") return c;
  if (name == "pixflow_search_20") {
    c.maxPercentage = 20;
    return c;
  }
  throw Error(-4, "unrecognized flow algorithm name: " + name);
}

BlurTaps gaussian_taps(int n, double sigma) {
  // cv::getGaussianKernel(n, sigma, CV_32F): exp in double, taps stored as float, normalised by
  // the double sum of the float taps.
  float k[16];
  const double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
  const double scale2X = -0.5 / (sigmaX * sigmaX);
  double sum = 0;
  for (int i = 0; i < n; ++i) {
    const double x = i - (n - 1) * 0.5;
    k[i] = (float)std::exp(scale2X * x * x);
    sum += k[i];
  }
  sum = 1. / sum;
  for (int i = 0; i < n; ++i) k[i] = (float)(k[i] * sum);
  BlurTaps t;
  std::memset(&t, 0, sizeof(t));
  t.r = n / 2;
  for (int j = 0; j <= t.r; ++j) t.k[j] = k[t.r + j];
  return t;
}

void FlowLevels::build(int dw, int dh, float pyrScale) {
  w.clear(); h.clear(); off.clear();
  total = 0;
  int cw = dw, ch = dh;
  for (;;) {
    w.push_back(cw); h.push_back(ch); off.push_back(total);
    total += (size_t)cw * ch;
    const int nw = int(cw * pyrScale + 0.5f), nh = int(ch * pyrScale + 0.5f);
    if (nh <= 24 || nw <= 24 || w.size() >= 1000) break;  // kPyrMinImageSize, kPyrMaxLevels
    cw = nw; ch = nh;
  }
}

// Layout of the uploaded table (8-byte words): [N][B][images N][prev_images N][prev_flow B][out B][int i0[B], int i1[B]]
const unsigned long long* FlowEngine::batch_tables(hipStream_t st, const FlowBatch& b) {
  const size_t N = b.images.size(), B = b.out.size();
  std::vector<unsigned long long> key;
  key.reserve(2 * N + 3 * B + 2);
  key.push_back(N);
  key.push_back(B);
  for (size_t k = 0; k < N; ++k) key.push_back((unsigned long long)b.images[k]);
  for (size_t k = 0; k < N; ++k) key.push_back(b.prev_images.empty() ? 0ull : (unsigned long long)b.prev_images[k]);
  for (size_t k = 0; k < B; ++k) key.push_back(b.prev_flow.empty() ? 0ull : (unsigned long long)b.prev_flow[k]);
  for (size_t k = 0; k < B; ++k) key.push_back((unsigned long long)b.out[k]);
  {
    std::vector<int> ints(2 * B);
    for (size_t k = 0; k < B; ++k) { ints[k] = b.i0[k]; ints[B + k] = b.i1[k]; }
    const size_t at = key.size();
    key.resize(at + B);
    std::memcpy(&key[at], ints.data(), 2 * B * sizeof(int));
  }
  for (TabSlot& t : tabs_)
    if (t.key == key) return t.buf.as<unsigned long long>() + 2;
  TabSlot& t = tabs_[tab_next_];
  tab_next_ = (tab_next_ + 1) % 4;
  S360_HIP(hipStreamSynchronize(st));  // earlier launches may still read the slot that is being replaced
  t.key = key;
  t.buf.ensure(key.size() * sizeof(unsigned long long));
  S360_HIP(hipMemcpyAsync(t.buf.p, t.key.data(), key.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, st));
  S360_HIP(hipStreamSynchronize(st));
  return t.buf.as<unsigned long long>() + 2;
}

void FlowEngine::compute(hipStream_t st, const PixFlowConsts& pc, const FlowBatch& batch, int w, int h, int hint) {
  const int N = (int)batch.images.size(), B = (int)batch.out.size();
  if (B < 1 || B > kMaxFlows || (int)batch.i0.size() != B || (int)batch.i1.size() != B)
    throw Error(-1, "FlowEngine: bad batch");
  if (!batch.prev_flow.empty() && ((int)batch.prev_flow.size() != B || (int)batch.prev_images.size() != N))
    throw Error(-1, "FlowEngine: previous-frame state must cover the whole batch");
  for (int b = 0; b < B; ++b)
    if (batch.i0[b] < 0 || batch.i0[b] >= N || batch.i1[b] < 0 || batch.i1[b] >= N) throw Error(-1, "FlowEngine: image index out of range");
  Profiler& P = *prof_;
  FlowBufs& M = *bufs_;
  dw_ = int(w * pc.downscaleFactor);
  dh_ = int(h * pc.downscaleFactor);
  // every caller, not only the operator entry point (the frame stages come here directly): the sweep kernels' tap
  // footprints and window placement assume at least 2 x 2 pixels at every pyramid level
  if (dw_ < 2 || dh_ < 2)
    throw Error(-1, "image too small for PixFlow: the reference's bilinear taps need a 2x2 image after the entry downscale (PixFlow.h:457-475)");
  const size_t n0 = (size_t)dw_ * dh_;
  lv_.build(dw_, dh_, pc.pyrScaleFactor);
  const int L = (int)lv_.w.size();
  const bool usePrev = !batch.prev_flow.empty();
  const unsigned long long* tab = batch_tables(st, batch);
  const uchar4* const* imageTab = reinterpret_cast<const uchar4* const*>(tab);
  const uchar4* const* prevImageTab = reinterpret_cast<const uchar4* const*>(tab + N);
  const float2* const* prevFlowTab = reinterpret_cast<const float2* const*>(tab + 2 * N);
  float* const* outTab = reinterpret_cast<float* const*>(tab + 2 * N + B);
  FlowIdx idx;
  idx.i0 = reinterpret_cast<const int*>(tab + 2 * N + 2 * B);
  idx.i1 = idx.i0 + B;

  M.down.ensure(N * n0 * sizeof(uchar4));
  M.gray.ensure(N * n0 * sizeof(float));
  M.pyrI.ensure(2 * N * lv_.total * sizeof(float));  // per level: N grey planes, then N alpha planes
  M.G.ensure(N * n0 * sizeof(float2));
  M.flowA.ensure(B * n0 * sizeof(float2));
  M.flowB.ensure(B * n0 * sizeof(float2));
  if (sweep_fast_ < 0) {
    const char* d = std::getenv("S360_SWEEP_DIV");  // "ieee": IEEE division / sqrt expansions instead of the verified fast ones (same bits)
    sweep_fast_ = !(d && std::string(d) == "ieee");
  }
  bool fastOk = false;
  if (sweep_fast_) {
    std::vector<float> divs;
    divs.push_back(0.001f);
    for (int l = 0; l < L; ++l) {
      divs.push_back((float)lv_.w[l]);
      divs.push_back((float)lv_.h[l]);
    }
    fastOk = sweep_verify_divisors(st, divs);
  }
  M.rec.ensure(B * n0 * (sweep_mode_ == 3 ? sizeof(float2) : sizeof(float4)));  // half-records / full records (flow_kernels.hpp)
  // Band hand-off granules + ticket counters of every sweep launch of this call (2 per level): one arena, reset to
  // all-ones ("not written") by ONE memset instead of one per launch.
  auto handoff_bytes = [&](int l) {
    const size_t b = sweep_mode_ == 3 ? sweep_quad_handoff_bytes(lv_.w[l], lv_.h[l], B)
                                      : sweep_lock_handoff_bytes(lv_.w[l], lv_.h[l], B, sweep_lock_waves());
    return (b + 255) & ~(size_t)255;
  };
  // + per level one word per (flow, row): all-ones = no pixel of the row is updated (written by the record kernel)
  auto rowflag_bytes = [&](int l) { return ((size_t)B * lv_.h[l] * sizeof(unsigned) + 255) & ~(size_t)255; };
  std::vector<size_t> hoff(L + 1, 0);
  for (int l = 0; l < L; ++l) hoff[l + 1] = hoff[l] + 2 * handoff_bytes(l) + rowflag_bytes(l);
  M.handoff.ensure(hoff[L]);
  S360_HIP(hipMemsetAsync(M.handoff.p, 0xFF, hoff[L], st));
  if (!err_.p) {
    err_.ensure(sizeof(unsigned));
    S360_HIP(hipMemsetAsync(err_.p, 0, sizeof(unsigned), st));
  }
  float* pyrI = M.pyrI.as<float>();
  auto LI = [&](int l) { return pyrI + (size_t)2 * N * lv_.off[l]; };
  auto LA = [&](int l) { return pyrI + (size_t)2 * N * lv_.off[l] + (size_t)N * lv_.w[l] * lv_.h[l]; };

  const BlurTaps tPre = gaussian_taps(5, 0.25f), tGrad = gaussian_taps(3, 0.5f), tFlow = gaussian_taps(15, 8.0f),
                 tFinal = gaussian_taps(3, 1.0f);
  {
    ProfScope ps(P, "flow_entry");
    launch_resize_cubic_u8c4(st, nullptr, w, h, 0, M.down.as<uchar4>(), dw_, dh_, n0, N, imageTab);
    launch_gray_alpha(st, M.down.as<uchar4>(), n0, n0, M.gray.as<float>(), LA(0), n0, N);
    launch_sepblur(st, M.gray.as<float>(), LI(0), dw_, dh_, 1, n0, N, tPre);
  }
  {
    ProfScope ps(P, "flow_pyramid");
    for (int l = 1; l < L; ++l) {
      const size_t ns = (size_t)lv_.w[l - 1] * lv_.h[l - 1], nd = (size_t)lv_.w[l] * lv_.h[l];
      // grey and alpha planes of a level are adjacent: one launch resizes all 2N planes
      launch_resize_linear_f32(st, LI(l - 1), lv_.w[l - 1], lv_.h[l - 1], ns, LI(l), lv_.w[l], lv_.h[l], nd, 1, 2 * N, 1.f,
                               0);
    }
  }
  float2* prevPyr = nullptr;
  float* motionPyr = nullptr;
  if (usePrev) {
    ProfScope ps(P, "flow_prev");
    M.prevdown.ensure(N * n0 * sizeof(uchar4));
    M.prevPyr.ensure(B * lv_.total * sizeof(float2));
    M.motionPyr.ensure(N * lv_.total * sizeof(float));
    prevPyr = M.prevPyr.as<float2>();
    motionPyr = M.motionPyr.as<float>();
    launch_resize_cubic_u8c4(st, nullptr, w, h, 0, M.prevdown.as<uchar4>(), dw_, dh_, n0, N, prevImageTab);
    launch_motion(st, M.down.as<uchar4>(), M.prevdown.as<uchar4>(), n0, n0, motionPyr, n0, N);
    // prevFlowDownscaled = resize(prevFlow) * (rows_down / rows_full)  (PixFlow.h:103-104)
    launch_resize_cubic_f32c2(st, nullptr, w, h, 0, prevPyr, dw_, dh_, n0, B, float(dh_) / float(h), prevFlowTab);
    for (int l = 1; l < L; ++l) {
      const size_t ns = (size_t)lv_.w[l - 1] * lv_.h[l - 1], nd = (size_t)lv_.w[l] * lv_.h[l];
      launch_resize_linear_f32(st, (const float*)(prevPyr + (size_t)B * lv_.off[l - 1]), lv_.w[l - 1], lv_.h[l - 1], ns,
                               (float*)(prevPyr + (size_t)B * lv_.off[l]), lv_.w[l], lv_.h[l], nd, 2, B, 1.f, 0);
      launch_resize_linear_f32(st, motionPyr + (size_t)N * lv_.off[l - 1], lv_.w[l - 1], lv_.h[l - 1], ns,
                               motionPyr + (size_t)N * lv_.off[l], lv_.w[l], lv_.h[l], nd, 1, N, 1.f, 0);
    }
    // (the rescale of the previous flow at each level, PixFlow.h:147-153 — level 0's factor is exactly 1 —, is applied where the
    // level is read: launch_diffusion_adjust below)
  }

  float2* cur = M.flowA.as<float2>();
  float2* oth = M.flowB.as<float2>();
  const float invPyr = 1.0f / pc.pyrScaleFactor;
  if (capture_levels) capture_levels->clear();
  for (int l = L - 1; l >= 0; --l) {
    const int wl = lv_.w[l], hl = lv_.h[l];
    const size_t nl = (size_t)wl * hl;
    {
      ProfScope ps(P, "flow_gradients");
      launch_gradients(st, LI(l), M.G.as<float2>(), wl, hl, nl, N, tGrad);
    }
    if (l == L - 1) {
      S360_HIP(hipMemsetAsync(cur, 0, B * nl * sizeof(float2), st));
      if (pc.maxPercentage > 0 && hint != 0) {
        ProfScope ps(P, "flow_search_init");
        M.I1eq.ensure(B * nl * sizeof(float));
        const int dist = (24 * pc.maxPercentage + 50) / 100;
        launch_search_init(st, LI(l), LA(l), wl, hl, nl, B, idx, cur, hint, dist, M.I1eq.as<float>());
      }
    }
    {
      ProfScope ps(P, "flow_blur15");  // the blurred flow goes straight into the sweeps' half-records
      launch_blur_to_records(st, cur, M.rec.p, wl, hl, nl, B, tFlow, sweep_mode_ == 3 ? nullptr : M.G.as<float2>(), LA(l), idx,
                             reinterpret_cast<unsigned*>((char*)M.handoff.p + hoff[l] + 2 * handoff_bytes(l)));
    }
    auto sweep = [&](float2* fl, int dir) {
      ProfScope ps(P, "flow_sweep");
      void* ho = (char*)M.handoff.p + hoff[l] + (dir > 0 ? 0 : handoff_bytes(l));
      if (sweep_mode_ == 3)
        launch_sweep_quad(st, M.rec.as<float2>(), M.G.as<float2>(), fl, ho, err_.as<unsigned>(), wl, hl, nl, B, idx, dir, pc,
                          fastOk, reinterpret_cast<const unsigned*>((char*)M.handoff.p + hoff[l] + 2 * handoff_bytes(l)));
      else
        launch_sweep_lock(st, M.rec.as<float4>(), M.G.as<float2>(), fl, ho, err_.as<unsigned>(), wl, hl, nl, B, idx, dir, pc,
                          fastOk);
    };
    sweep(cur, +1);
    {
      ProfScope ps(P, "flow_median");
      launch_median5_c2(st, cur, oth, wl, hl, nl, B);
    }
    sweep(oth, -1);
    {
      ProfScope ps(P, "flow_median");
      launch_median5_c2(st, oth, cur, wl, hl, nl, B);
    }
    {
      ProfScope ps(P, "flow_diffusion");
      if (usePrev)  // ... and adjustFlowTowardPrevious in the same pass (the previous flow's level rescaled as it is read)
        launch_diffusion_adjust(st, cur, oth, wl, hl, nl, B, tFlow, LA(l), idx, prevPyr + (size_t)B * lv_.off[l],
                                motionPyr + (size_t)N * lv_.off[l], l == 0 ? 1.0f : float(lv_.h[l]) / float(lv_.h[0]));
      else
        launch_diffusion(st, cur, oth, wl, hl, nl, B, tFlow, LA(l), idx);
    }
    if (capture_levels) {
      std::vector<float> hbuf(B * nl * 2);
      S360_HIP(hipMemcpyAsync(hbuf.data(), oth, hbuf.size() * sizeof(float), hipMemcpyDeviceToHost, st));
      S360_HIP(hipStreamSynchronize(st));
      capture_levels->push_back(std::move(hbuf));
    }
    if (l > 0) {
      ProfScope ps(P, "flow_upscale");
      launch_resize_cubic_f32c2(st, oth, wl, hl, nl, cur, lv_.w[l - 1], lv_.h[l - 1],
                                (size_t)lv_.w[l - 1] * lv_.h[l - 1], B, invPyr);
    } else {
      ProfScope ps(P, "flow_final");
      // final upscale + scalar + 3x3 blur fused: the upscaled flow is evaluated while the blur's tile is loaded
      launch_upscale_blur(st, oth, wl, hl, nl, nullptr, w, h, (size_t)w * h, B, 1.0f / pc.downscaleFactor, tFinal, outTab);
    }
  }
}

unsigned FlowEngine::take_error(hipStream_t st) {
  if (!err_.p) return 0;
  unsigned v = 0;
  S360_HIP(hipMemcpyAsync(&v, err_.p, sizeof(v), hipMemcpyDeviceToHost, st));
  S360_HIP(hipStreamSynchronize(st));
  if (v) S360_HIP(hipMemsetAsync(err_.p, 0, sizeof(unsigned), st));
  return v;
}

}  // namespace s360
