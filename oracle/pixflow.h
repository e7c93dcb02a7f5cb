// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
// The logic restated here is PINNED to the reference's source: optical_flow/PixFlow.h compiled from /root/reference over
// oracle/ref_shim (oracle/ref_pixflow.cpp) gives the same bits (tests/test_cpu_refpin.py). The OpenCV primitives it calls
// (cvlite.h) stay unpinned (see that header).
//
// pixflow.h: CPU restatement of PixFlow<false, MaxPercentage>::computeOpticalFlow
// following surround360_render/source/optical_flow/PixFlow.h:81-534 and the
// constants of OpticalFlowFactory.h:25-61. Raster-order Gauss-Seidel sweeps
// exactly as written (PixFlow.h:388-410).
#pragma once
#include <string>

#include "cvlite.h"

namespace orc {

enum DirectionHint { HINT_UNKNOWN = 0, HINT_RIGHT = 1, HINT_DOWN = 2, HINT_LEFT = 3, HINT_UP = 4 };

struct PixFlowParams {
  // OpticalFlowFactory.h:26-41 ("pixflow_low") / :45-60 ("pixflow_search_20")
  float pyrScaleFactor = 0.9f;
  float smoothnessCoef = 0.001f;
  float verticalRegularizationCoef = 0.01f;
  float horizontalRegularizationCoef = 0.01f;
  float gradientStepSize = 0.5f;
  float downscaleFactor = 0.5f;
  int maxPercentage = 0;  // 0: pixflow_low, 20: pixflow_search_20
};

// makeOpticalFlowByName (OpticalFlowFactory.h:23-64); returns false for unknown names.
static inline bool pixflowParamsByName(const std::string& name, PixFlowParams* p) {
  *p = PixFlowParams();
  if (name == "pixflow_low") return true;
  if (name == "pixflow_search_20") { p->maxPercentage = 20; return true; }
  return false;
}

struct PixFlow {
  // PixFlow.h:37-49
  static constexpr int kPyrMinImageSize = 24;
  static constexpr int kPyrMaxLevels = 1000;
  static constexpr float kGradEpsilon = 0.001f;
  static constexpr float kUpdateAlphaThreshold = 0.9f;
  static constexpr int kPreBlurKernelWidth = 5;
  static constexpr float kPreBlurSigma = 0.25f;
  static constexpr int kFinalFlowBlurKernelWidth = 3;
  static constexpr float kFinalFlowBlurSigma = 1.0f;
  static constexpr int kGradientBlurKernelWidth = 3;
  static constexpr float kGradientBlurSigma = 0.5f;
  static constexpr int kBlurredFlowKernelWidth = 15;
  static constexpr float kBlurredFlowSigma = 8.0f;

  PixFlowParams P;
  explicit PixFlow(const PixFlowParams& p) : P(p) {}

  // Optional taps for per-stage parity tests: if non-null, receives copies of
  // intermediate results.
  struct Debug {
    ImgU8 down0, down1;
    ImgF I0, I1, alpha0, alpha1;  // level 0 (after pre-blur)
    std::vector<ImgF> flowPerLevel;  // flow after each level (coarsest first)
  };
  Debug* dbg = nullptr;

  // PixFlow.h:477-491
  std::vector<ImgF> buildPyramid(const ImgF& src) const {
    std::vector<ImgF> pyr;
    pyr.push_back(src);
    if (src.empty()) return pyr;
    while ((int)pyr.size() < kPyrMaxLevels) {
      const int nw = int(pyr.back().w * P.pyrScaleFactor + 0.5f);
      const int nh = int(pyr.back().h * P.pyrScaleFactor + 0.5f);
      if (nh <= kPyrMinImageSize || nw <= kPyrMinImageSize) break;
      pyr.push_back(resizeLinearF32(pyr.back(), nw, nh));
    }
    return pyr;
  }

  // PixFlow.h:457-475
  static inline float getPixBilinear32FExtend(const ImgF& img, float x, float y) {
    x = std::min(img.w - 2.0f, std::max(0.0f, x));
    y = std::min(img.h - 2.0f, std::max(0.0f, y));
    const int x0 = int(x);
    const int y0 = int(y);
    const float xR = x - float(x0);
    const float yR = y - float(y0);
    const float* p = img.row(y0);
    const float f00 = p[x0];
    const float f01 = p[x0 + img.w];
    const float f10 = p[x0 + 1];
    const float f11 = p[x0 + img.w + 1];
    const float a1 = f00;
    const float a2 = f10 - f00;
    const float a3 = f01 - f00;
    const float a4 = f00 + f11 - f10 - f01;
    return a1 + a2 * xR + a3 * yR + a4 * xR * yR;
  }

  struct Level {
    const ImgF *I0, *I1, *alpha0, *alpha1;
    ImgF I0x, I0y, I1x, I1y, blurredFlow;
  };

  // PixFlow.h:493-534 (UseDirectionalRegularization == false for both registered algorithms)
  inline float errorFunction(const Level& L, int x, int y, float fdx, float fdy) const {
    const float matchX = x + fdx;
    const float matchY = y + fdy;
    const float i0x = L.I0x.at(y, x);
    const float i0y = L.I0y.at(y, x);
    const float i1x = getPixBilinear32FExtend(L.I1x, matchX, matchY);
    const float i1y = getPixBilinear32FExtend(L.I1y, matchX, matchY);
    const float dfx = L.blurredFlow.at(y, x, 0) - fdx;
    const float dfy = L.blurredFlow.at(y, x, 1) - fdy;
    const float smoothness = sqrtf(dfx * dfx + dfy * dfy);
    float err = sqrtf((i0x - i1x) * (i0x - i1x) + (i0y - i1y) * (i0y - i1y)) +
                smoothness * P.smoothnessCoef +
                P.verticalRegularizationCoef * fabsf(fdy) / float(L.I0->w) +
                P.horizontalRegularizationCoef * fabsf(fdx) / float(L.I0->h);
    return err;
  }

  // PixFlow.h:415-435
  inline void proposeFlowUpdate(const Level& L, ImgF& flow, float& currErr, int x, int y, float px,
                                float py) const {
    const float proposalErr = errorFunction(L, x, y, px, py);
    if (proposalErr < currErr) {
      flow.at(y, x, 0) = px;
      flow.at(y, x, 1) = py;
      currErr = proposalErr;
    }
  }

  // one pixel of a sweep (PixFlow.h:390-395 / 403-408); dir=+1 forward, -1 backward
  inline void sweepPixel(const Level& L, ImgF& flow, int x, int y, int dir) const {
    const int w = flow.w, h = flow.h;
    if (L.alpha0->at(y, x) > kUpdateAlphaThreshold && L.alpha1->at(y, x) > kUpdateAlphaThreshold) {
      float currErr = errorFunction(L, x, y, flow.at(y, x, 0), flow.at(y, x, 1));
      if (dir > 0) {
        if (x > 0) proposeFlowUpdate(L, flow, currErr, x, y, flow.at(y, x - 1, 0), flow.at(y, x - 1, 1));
        if (y > 0) proposeFlowUpdate(L, flow, currErr, x, y, flow.at(y - 1, x, 0), flow.at(y - 1, x, 1));
      } else {
        if (x < w - 1) proposeFlowUpdate(L, flow, currErr, x, y, flow.at(y, x + 1, 0), flow.at(y, x + 1, 1));
        if (y < h - 1) proposeFlowUpdate(L, flow, currErr, x, y, flow.at(y + 1, x, 0), flow.at(y + 1, x, 1));
      }
      // errorGradient, PixFlow.h:195-217
      const float fx0 = flow.at(y, x, 0), fy0 = flow.at(y, x, 1);
      const float ex = errorFunction(L, x, y, fx0 + kGradEpsilon, fy0 + 0.0f);
      const float ey = errorFunction(L, x, y, fx0 + 0.0f, fy0 + kGradEpsilon);
      const float gx = (ex - currErr) / kGradEpsilon;
      const float gy = (ey - currErr) / kGradEpsilon;
      flow.at(y, x, 0) = fx0 - P.gradientStepSize * gx;
      flow.at(y, x, 1) = fy0 - P.gradientStepSize * gy;
    }
  }

  // PixFlow.h:437-454
  void lowAlphaFlowDiffusion(const ImgF& alpha0, const ImgF& alpha1, ImgF& flow) const {
    ImgF blurred = gaussianBlurF32(flow, kBlurredFlowKernelWidth, kBlurredFlowSigma);
    for (int y = 0; y < flow.h; ++y)
      for (int x = 0; x < flow.w; ++x) {
        const float a0 = alpha0.at(y, x), a1 = alpha1.at(y, x);
        const float diffusionCoef = 1.0f - a0 * a1;
        for (int k = 0; k < 2; ++k)
          flow.at(y, x, k) = diffusionCoef * blurred.at(y, x, k) + (1.0f - diffusionCoef) * flow.at(y, x, k);
      }
  }

  // ---- pixflow_search_20 only (PixFlow.h:219-342) ----
  int computeSearchDistance() const { return (kPyrMinImageSize * P.maxPercentage + 50) / 100; }

  float computePatchError(const ImgF& i0, const ImgF& alpha0, int i0x, int i0y, const ImgF& i1,
                          const ImgF& alpha1, int i1x, int i1y) const {
    const int kPatchRadius = 2;
    float sad = 0, alpha = 0;
    for (int dy = -kPatchRadius; dy <= kPatchRadius; ++dy) {
      const int d0y = i0y + dy;
      if (0 <= d0y && d0y < i0.h) {
        const int d1y = std::min(std::max(i1y + dy, 0), i1.h - 1);
        for (int dx = -kPatchRadius; dx <= kPatchRadius; ++dx) {
          const int d0x = i0x + dx;
          if (0 <= d0x && d0x < i0.w) {
            const int d1x = std::min(std::max(i1x + dx, 0), i1.w - 1);
            const float difference = i0.at(d0y, d0x) - i1.at(d1y, d1x);
            sad += std::abs(difference);
            alpha += alpha0.at(d0y, d0x) * alpha1.at(d1y, d1x);
          }
        }
      }
    }
    sad /= alpha;
    const float ddx = float(i1x - i0x), ddy = float(i1y - i0y);
    const float length = (float)std::sqrt((double)ddx * ddx + (double)ddy * ddy);
    sad *= 1 + length / computeSearchDistance();
    return sad;
  }

  void adjustInitialFlow(const ImgF& I0, const ImgF& I1, const ImgF& alpha0, const ImgF& alpha1, ImgF& flow,
                         int hint) const {
    float sumLhs = 0, sumRhs = 0;  // computeIntensityRatio, PixFlow.h:261-277
    for (int y = 0; y < I0.h; ++y)
      for (int x = 0; x < I0.w; ++x) {
        const float a = alpha0.at(y, x) * alpha1.at(y, x);
        sumLhs += a * I0.at(y, x);
        sumRhs += a * I1.at(y, x);
      }
    const float ratio = sumLhs / sumRhs;
    ImgF I1eq(I1.w, I1.h, 1);
    for (size_t i = 0; i < I1.d.size(); ++i) I1eq.d[i] = I1.d[i] * ratio;
    // computeSearchBox, PixFlow.h:279-296
    const int dist = computeSearchDistance();
    const int kRatio = 8;
    const int ortho = (dist + kRatio / 2) / kRatio;
    const int thickness = 2 * ortho + 1;
    int bx, by, bw, bh;
    switch (hint) {
      case HINT_RIGHT: bx = 0; by = -ortho; bw = dist + 1; bh = thickness; break;
      case HINT_DOWN: bx = -ortho; by = 0; bw = thickness; bh = dist + 1; break;
      case HINT_LEFT: bx = -dist; by = -ortho; bw = dist + 1; bh = thickness; break;
      default: bx = -ortho; by = -dist; bw = thickness; bh = dist + 1; break;  // UP
    }
    for (int i0y = 0; i0y < I0.h; ++i0y)
      for (int i0x = 0; i0x < I0.w; ++i0x)
        if (alpha0.at(i0y, i0x) > kUpdateAlphaThreshold) {
          const float kFraction = 0.8f;
          float errorBest = kFraction * computePatchError(I0, alpha0, i0x, i0y, I1eq, alpha1, i0x, i0y);
          int i1xBest = i0x, i1yBest = i0y;
          for (int dy = by; dy < by + bh; ++dy)
            for (int dx = bx; dx < bx + bw; ++dx) {
              const int i1x = i0x + dx, i1y = i0y + dy;
              if (0 <= i1x && i1x < I1.w && 0 <= i1y && i1y < I1.h) {
                const float error = computePatchError(I0, alpha0, i0x, i0y, I1eq, alpha1, i1x, i1y);
                if (errorBest > error) { errorBest = error; i1xBest = i1x; i1yBest = i1y; }
              }
            }
          flow.at(i0y, i0x, 0) = float(i1xBest - i0x);
          flow.at(i0y, i0x, 1) = float(i1yBest - i0y);
        }
  }

  // PixFlow.h:344-413
  void patchMatchPropagationAndSearch(const ImgF& I0, const ImgF& I1, const ImgF& alpha0, const ImgF& alpha1,
                                      ImgF& flow, int hint) const {
    Level L;
    L.I0 = &I0; L.I1 = &I1; L.alpha0 = &alpha0; L.alpha1 = &alpha1;
    L.I0x = gaussianBlurF32(sobelX(I0), kGradientBlurKernelWidth, kGradientBlurSigma);
    L.I0y = gaussianBlurF32(sobelY(I0), kGradientBlurKernelWidth, kGradientBlurSigma);
    L.I1x = gaussianBlurF32(sobelX(I1), kGradientBlurKernelWidth, kGradientBlurSigma);
    L.I1y = gaussianBlurF32(sobelY(I1), kGradientBlurKernelWidth, kGradientBlurSigma);
    if (flow.empty()) {
      flow = ImgF(I0.w, I0.h, 2, 0.f);
      if (P.maxPercentage > 0 && hint != HINT_UNKNOWN) adjustInitialFlow(I0, I1, alpha0, alpha1, flow, hint);
    }
    L.blurredFlow = gaussianBlurF32(flow, kBlurredFlowKernelWidth, kBlurredFlowSigma);
    const int w = I0.w, h = I0.h;
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) sweepPixel(L, flow, x, y, +1);
    flow = medianBlur5(flow);
    for (int y = h - 1; y >= 0; --y)
      for (int x = w - 1; x >= 0; --x) sweepPixel(L, flow, x, y, -1);
    flow = medianBlur5(flow);
    lowAlphaFlowDiffusion(alpha0, alpha1, flow);
  }

  // PixFlow.h:81-183. prevFlow/prevI0/prevI1 may be empty (first frame / photo).
  void computeOpticalFlow(const ImgU8& rgba0, const ImgU8& rgba1, const ImgF& prevFlow, const ImgU8& prevI0,
                          const ImgU8& prevI1, ImgF& flow, int hint) const {
    assert(rgba0.c == 4 && rgba1.c == 4);
    const int ow = rgba0.w, oh = rgba0.h;
    const int dw = int(rgba0.w * P.downscaleFactor), dh = int(rgba0.h * P.downscaleFactor);
    ImgU8 d0 = resizeCubicU8(rgba0, dw, dh), d1 = resizeCubicU8(rgba1, dw, dh);
    const bool usePrev = !prevFlow.empty();
    ImgF prevFlowDown, motion;
    if (usePrev) {
      prevFlowDown = resizeCubicF32(prevFlow, dw, dh);
      const float s = float(prevFlowDown.h) / float(prevFlow.h);
      for (float& v : prevFlowDown.d) v *= s;
      ImgU8 p1 = resizeCubicU8(prevI1, dw, dh);  // prevI0 is resized by the reference but never used (:106)
      motion = ImgF(dw, dh, 1);
      for (int y = 0; y < dh; ++y)
        for (int x = 0; x < dw; ++x) {
          // fabs(int) -> double; / (255.0f * 3.0f) (PixFlow.h:112-115)
          const double s3 = std::fabs(double(int(d1.at(y, x, 0)) - int(p1.at(y, x, 0)))) +
                            std::fabs(double(int(d1.at(y, x, 1)) - int(p1.at(y, x, 1)))) +
                            std::fabs(double(int(d1.at(y, x, 2)) - int(p1.at(y, x, 2))));
          motion.at(y, x) = float(s3 / (255.0f * 3.0f));
        }
    }
    // grey + alpha as float in [0,1] (PixFlow.h:121-135); "/= 255.0f" is a float multiply by float(1/255.0)
    const float inv255 = (float)(1.0 / 255.0);
    ImgF I0(dw, dh, 1), I1(dw, dh, 1), alpha0(dw, dh, 1), alpha1(dw, dh, 1);
    for (int y = 0; y < dh; ++y)
      for (int x = 0; x < dw; ++x) {
        const uint8_t* p = d0.px(y, x);
        const uint8_t* q = d1.px(y, x);
        I0.at(y, x) = float(bgr2gray(p[0], p[1], p[2])) * inv255;
        I1.at(y, x) = float(bgr2gray(q[0], q[1], q[2])) * inv255;
        alpha0.at(y, x) = float(p[3]) * inv255;
        alpha1.at(y, x) = float(q[3]) * inv255;
      }
    I0 = gaussianBlurF32(I0, kPreBlurKernelWidth, kPreBlurSigma);
    I1 = gaussianBlurF32(I1, kPreBlurKernelWidth, kPreBlurSigma);
    if (dbg) { dbg->down0 = d0; dbg->down1 = d1; dbg->I0 = I0; dbg->I1 = I1; dbg->alpha0 = alpha0; dbg->alpha1 = alpha1; }

    std::vector<ImgF> pyrI0 = buildPyramid(I0), pyrI1 = buildPyramid(I1);
    std::vector<ImgF> pyrA0 = buildPyramid(alpha0), pyrA1 = buildPyramid(alpha1);
    std::vector<ImgF> pyrPrev, pyrMotion;
    if (usePrev) {
      pyrPrev = buildPyramid(prevFlowDown);
      pyrMotion = buildPyramid(motion);
      for (size_t l = 0; l < pyrPrev.size(); ++l) {
        const float s = float(pyrPrev[l].h) / float(pyrPrev[0].h);
        for (float& v : pyrPrev[l].d) v *= s;
      }
    }
    flow = ImgF();
    const float invPyr = 1.0f / P.pyrScaleFactor;
    for (int level = (int)pyrI0.size() - 1; level >= 0; --level) {
      patchMatchPropagationAndSearch(pyrI0[level], pyrI1[level], pyrA0[level], pyrA1[level], flow, hint);
      if (usePrev) {  // adjustFlowTowardPrevious, PixFlow.h:185-193
        const ImgF& pf = pyrPrev[level];
        const ImgF& mo = pyrMotion[level];
        for (int y = 0; y < flow.h; ++y)
          for (int x = 0; x < flow.w; ++x) {
            const float w = 1.0f - mo.at(y, x);
            for (int k = 0; k < 2; ++k) flow.at(y, x, k) = flow.at(y, x, k) * (1.0f - w) + pf.at(y, x, k) * w;
          }
      }
      if (dbg) dbg->flowPerLevel.push_back(flow);
      if (level > 0) {
        flow = resizeCubicF32(flow, pyrI0[level - 1].w, pyrI0[level - 1].h);
        for (float& v : flow.d) v *= invPyr;
      }
    }
    flow = resizeLinearF32(flow, ow, oh);
    const float invDown = 1.0f / P.downscaleFactor;
    for (float& v : flow.d) v *= invDown;
    flow = gaussianBlurF32(flow, kFinalFlowBlurKernelWidth, kFinalFlowBlurSigma);
  }
};

}  // namespace orc
