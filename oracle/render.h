// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
// The logic restated here is PINNED to the reference's source: the reference's whole TestRenderStereoPanorama program,
// compiled from /root/reference over oracle/ref_shim (make -C oracle ref), writes the same bytes as this file's pipeline
// for two chained frames, sharpening + cubemap + pixflow_search_20, and pole removal — equirects, cubemap, every flow and
// state image (tests/test_cpu_refprogram.py). The OpenCV / Eigen primitives under it stay unpinned (cvlite.h header).
//
// render.h: CPU restatement of the per-frame stereo panorama pipeline:
//   surround360_render/source/test/TestRenderStereoPanorama.cpp (TRSP) :75-972
//   surround360_render/source/render/ImageWarper.cpp:143-174
//   surround360_render/source/optical_flow/NovelView.cpp:101-299
//   surround360_render/source/util/CvUtil.cpp:69-115,140-157,224-260
//   surround360_render/source/util/Filter.h:40-127
// Single-threaded; the reference's std::thread fan-out (one thread per camera /
// pair / pole-eye) is applied by the caller (oracle_capi.cpp) where timing needs it.
#pragma once
#include <chrono>
#include <string>
#include <thread>
#include <utility>

#include "camera.h"
#include "cvlite.h"
#include "pixflow.h"

namespace orc {

// MathUtil.h:29-59
static inline float rampf(float x, float a, float b) { return std::max(0.0f, std::min(1.0f, (x - a) / (b - a))); }
static inline float lerpf(float x0, float x1, float alpha) { return x0 * (1.0f - alpha) + x1 * alpha; }

// The gflags of TRSP:44-70 that influence pixels.
struct RenderParams {
  double interpupilary_dist = 6.4;
  int side_alpha_feather_size = 100;
  int std_alpha_feather_size = 31;
  double sharpening = 0.0;
  int enable_top = 0, enable_bottom = 0;
  std::string side_flow_alg = "pixflow_low", polar_flow_alg = "pixflow_low";
  double zero_parallax_dist = 10000.0;
  int eqr_width = 256, eqr_height = 128;
  int final_eqr_width = 3480, final_eqr_height = 960;
  int enable_pole_removal = 0;
  std::string poleremoval_flow_alg = "pixflow_low";
};

// ---------------------------------------------------------------------------
// ImageWarper.cpp:143-174. Returns the float warp map (2ch) for a dst of
// dstW x dstH. cos/sin of the (float) angles resolve to the float overloads
// (`using namespace std`), the products are float, Camera::pixel is double.
static inline ImgF sphericalWarpMap(int dstW, int dstH, const Camera& cam, float leftAngle, float rightAngle,
                                    float topAngle, float bottomAngle) {
  ImgF warp(dstW, dstH, 2);
  for (int x = 0; x < dstW; ++x) {
    const float xFrac = (x + 0.5f) / dstW;
    const float xAngle = (1 - xFrac) * leftAngle + xFrac * rightAngle;
    for (int y = 0; y < dstH; ++y) {
      const float yFrac = (y + 0.5f) / dstH;
      const float yAngle = (1 - yFrac) * topAngle + yFrac * bottomAngle;
      const float ux = cosf(yAngle) * cosf(xAngle);
      const float uy = cosf(yAngle) * sinf(xAngle);
      const float uz = sinf(yAngle);
      const int kNear = int(Camera::kNearInfinity);
      const V2 p = cam.pixel(V3(double(ux) * kNear, double(uy) * kNear, double(uz) * kNear));
      warp.at(y, x, 0) = float(p.x - 0.5);
      warp.at(y, x, 1) = float(p.y - 0.5);
    }
  }
  return warp;
}
static inline ImgU8 bicubicRemapToSpherical(int dstW, int dstH, int dstC, const ImgU8& src, const Camera& cam,
                                            float l, float r, float t, float b) {
  ImgF warp = sphericalWarpMap(dstW, dstH, cam, l, r, t, b);
  if (src.c == 3 && dstC == 4) return remapCubicU8(bgr2bgra(src), warp);
  return remapCubicU8(src, warp);
}

// TRSP:99-135
static inline ImgU8 projectSideToSpherical(int dstW, int dstH, const ImgU8& src, const Camera& cam, float l,
                                           float r, float t, float b, int featherSize) {
  ImgU8 tmp = (src.c == 3) ? bgr2bgra(src) : src;
  if (featherSize) {
    for (int y = 0; y < featherSize; ++y) {
      const uint8_t alpha = (uint8_t)(255.0f * float(y + 0.5f) / float(featherSize));
      for (int x = 0; x < tmp.w; ++x) {
        tmp.at(y, x, 3) = alpha;
        tmp.at(tmp.h - 1 - y, x, 3) = alpha;
      }
    }
  }
  return bicubicRemapToSpherical(dstW, dstH, 4, tmp, cam, l, r, t, b);
}

// Derived sizes/angles of TRSP:153-173 and :309-348.
struct SideGeometry {
  float hRadians, vRadians;
  int camImageWidth, camImageHeight;  // side projection size
  int overlapImageWidth, numNovelViews;
  float fovHorizontalRadians, vergeAtInfinitySlabDisplacement, zeroParallaxNovelViewShiftPixels;
};
static inline SideGeometry sideGeometry(const RigDescription& rig, const RenderParams& P) {
  SideGeometry g;
  g.hRadians = 2 * approximateFov(rig.rigSideOnly, false);
  g.vRadians = 2 * approximateFov(rig.rigSideOnly, true);
  g.camImageHeight = int(P.eqr_height * g.vRadians / M_PI);
  g.camImageWidth = int(P.eqr_width * g.hRadians / (2 * M_PI));
  const int numCams = (int)rig.rigSideOnly.size();
  const double fovHorizontal = 2 * approximateFov(rig.rigSideOnly, false) * (180 / M_PI);  // TRSP:781-782
  const float camFovHorizontalDegrees = (float)fovHorizontal;
  g.fovHorizontalRadians = (float)(camFovHorizontalDegrees * M_PI / 180.0f);  // toRadians(float)
  const float overlapAngleDegrees = (float)((camFovHorizontalDegrees * float(numCams) - 360.0) / float(numCams));
  g.overlapImageWidth = int(float(g.camImageWidth) * (overlapAngleDegrees / camFovHorizontalDegrees));
  g.numNovelViews = g.camImageWidth - g.overlapImageWidth;
  const float cameraRingRadius = rig.getRingRadius();
  const float v = atanf((float)(P.zero_parallax_dist / (P.interpupilary_dist / 2.0f)));
  const float psi = asinf((float)(sinf(v) * (P.interpupilary_dist / 2.0f) / cameraRingRadius));
  g.vergeAtInfinitySlabDisplacement = psi * (float(g.camImageWidth) / g.fovHorizontalRadians);
  const float theta = (float)(-M_PI / 2.0f + v + psi);
  g.zeroParallaxNovelViewShiftPixels = (float)(float(P.eqr_width) * (theta / (2.0f * M_PI)));
  return g;
}
// angles for side camera camIdx (TRSP:163-173)
static inline void sideCameraAngles(const SideGeometry& g, int camIdx, int numCams, float* l, float* r, float* t,
                                    float* b) {
  const float direction = (float)(-float(camIdx) / float(numCams) * 2.0f * M_PI);
  *l = direction + g.hRadians / 2;
  *r = direction - g.hRadians / 2;
  *t = g.vRadians / 2;
  *b = -g.vRadians / 2;
}

// ---------------------------------------------------------------------------
// NovelView.cpp:174-224 with the analytic LazyNovelViewBuffer of TRSP:271-285:
// column u: x_src = slabShift(u) +/- disp, y_src = v, t = u / numNovelViews.
struct LazyView { ImgU8 img; ImgF flowMag; };
static inline LazyView renderLazyNovelView(int width, int height, int camImageWidth, int numNovelViews, float disp,
                                           const ImgU8& srcImage, const ImgF& opticalFlow, bool invertT) {
  ImgF warpFlow(width, height, 2);
  std::vector<float> tcol(width);
  for (int x = 0; x < width; ++x) {
    const float shift = float(x) / float(numNovelViews);
    const float slabShift = float(camImageWidth) * 0.5f - float(numNovelViews - x);
    tcol[x] = shift;
    for (int y = 0; y < height; ++y) {
      warpFlow.at(y, x, 0) = slabShift + disp;
      warpFlow.at(y, x, 1) = float(y);
    }
  }
  ImgF remappedFlow = remapCubicF32(opticalFlow, warpFlow);
  ImgF warpComp(width, height, 2);
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x) {
      const float t = invertT ? (1.0f - tcol[x]) : tcol[x];
      warpComp.at(y, x, 0) = warpFlow.at(y, x, 0) + remappedFlow.at(y, x, 0) * t;
      warpComp.at(y, x, 1) = warpFlow.at(y, x, 1) + remappedFlow.at(y, x, 1) * t;
    }
  LazyView out;
  out.img = remapCubicU8(srcImage, warpComp);
  out.flowMag = ImgF(width, height, 1);
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x) {
      const float t = invertT ? (1.0f - tcol[x]) : tcol[x];
      out.img.at(y, x, 3) = (uint8_t)int((1.0f - t) * out.img.at(y, x, 3));
      const float fx = remappedFlow.at(y, x, 0), fy = remappedFlow.at(y, x, 1);
      out.flowMag.at(y, x) = sqrtf(fx * fx + fy * fy);
    }
  return out;
}

// NovelView.cpp:101-154
static inline ImgU8 combineLazyViews(const ImgU8& imageL, const ImgU8& imageR, const ImgF& flowMagL,
                                     const ImgF& flowMagR) {
  ImgU8 blend(imageL.w, imageL.h, 4);
  for (int y = 0; y < imageL.h; ++y)
    for (int x = 0; x < imageL.w; ++x) {
      const uint8_t* cL = imageL.px(y, x);
      const uint8_t* cR = imageR.px(y, x);
      uint8_t* o = blend.px(y, x);
      const uint8_t outAlpha = (double)(std::max(cL[3], cR[3]) / 255.0f) > 0.1 ? 255 : 0;
      if (cL[3] == 0 && cR[3] == 0) {
        o[0] = o[1] = o[2] = 0; o[3] = outAlpha;
      } else if (cL[3] == 0) {
        o[0] = cR[0]; o[1] = cR[1]; o[2] = cR[2]; o[3] = outAlpha;
      } else if (cR[3] == 0) {
        o[0] = cL[0]; o[1] = cL[1]; o[2] = cL[2]; o[3] = outAlpha;
      } else {
        const float magL = flowMagL.at(y, x) / float(imageL.w);
        const float magR = flowMagR.at(y, x) / float(imageL.w);
        float blendL = float(cL[3]), blendR = float(cR[3]);
        const float norm = blendL + blendR;
        blendL /= norm;
        blendR /= norm;
        const float colorDiff =
            (std::abs(cL[0] - cR[0]) + std::abs(cL[1] - cR[1]) + std::abs(cL[2] - cR[2])) / 255.0f;
        const float kColorDiffCoef = 10.0f, kSoftmaxSharpness = 10.0f, kFlowMagCoef = 20.0f;
        const float deghostCoef = tanhf(colorDiff * kColorDiffCoef);
        const double expL = exp(kSoftmaxSharpness * blendL * (1.0 + kFlowMagCoef * magL));
        const double expR = exp(kSoftmaxSharpness * blendR * (1.0 + kFlowMagCoef * magR));
        const double sumExp = expL + expR + 0.00001;
        const float softmaxL = float(expL / sumExp);
        const float softmaxR = float(expR / sumExp);
        const float wL = lerpf(blendL, softmaxL, deghostCoef), wR = lerpf(blendR, softmaxR, deghostCoef);
        o[0] = truncU8(float(cL[0]) * wL + float(cR[0]) * wR);
        o[1] = truncU8(float(cL[1]) * wL + float(cR[1]) * wR);
        o[2] = truncU8(float(cL[2]) * wL + float(cR[2]) * wR);
        o[3] = 255;
      }
    }
  return blend;
}

// NovelView.cpp:226-268 + TRSP:259-292: the two eye chunks of one camera pair.
static inline std::pair<ImgU8, ImgU8> combineLazyNovelViews(const SideGeometry& g, int eqrWidth, int numCams,
                                                            const ImgU8& imageL, const ImgU8& imageR,
                                                            const ImgF& flowLtoR, const ImgF& flowRtoL) {
  const int width = eqrWidth / numCams, height = g.camImageHeight;  // LazyNovelViewBuffer(eqr_width/numCams, h)
  const float d = g.vergeAtInfinitySlabDisplacement;
  LazyView lfl = renderLazyNovelView(width, height, g.camImageWidth, g.numNovelViews, +d, imageL, flowRtoL, false);
  LazyView lfr = renderLazyNovelView(width, height, g.camImageWidth, g.numNovelViews, +d, imageR, flowLtoR, true);
  LazyView rfl = renderLazyNovelView(width, height, g.camImageWidth, g.numNovelViews, -d, imageL, flowRtoL, false);
  LazyView rfr = renderLazyNovelView(width, height, g.camImageWidth, g.numNovelViews, -d, imageR, flowLtoR, true);
  return std::make_pair(combineLazyViews(lfl.img, lfr.img, lfl.flowMag, lfr.flowMag),
                        combineLazyViews(rfl.img, rfr.img, rfl.flowMag, rfr.flowMag));
}

// CvUtil.cpp:93-115 (remap INTER_NEAREST, BORDER_WRAP)
static inline ImgU8 offsetHorizontalWrap(const ImgU8& src, float offset) {
  ImgU8 dst(src.w, src.h, src.c);
  for (int x = 0; x < src.w; ++x) {
    float srcX = float(x) - offset;
    if (srcX < 0) srcX += src.w;
    if (srcX >= src.w) srcX -= src.w;
    int sx = (int)satS16(cvRoundF(srcX));
    if (sx < 0) sx -= ((sx - src.w + 1) / src.w) * src.w;
    if (sx >= src.w) sx %= src.w;
    for (int y = 0; y < src.h; ++y) std::memcpy(dst.px(y, x), src.px(y, sx), src.c);
  }
  return dst;
}

static inline ImgU8 stackHorizontal(const std::vector<ImgU8>& v) {
  int w = 0;
  for (const ImgU8& i : v) w += i.w;
  ImgU8 d(w, v[0].h, v[0].c);
  int x0 = 0;
  for (const ImgU8& i : v) {
    for (int y = 0; y < i.h; ++y) std::memcpy(d.px(y, x0), i.row(y), size_t(i.w) * i.c);
    x0 += i.w;
  }
  return d;
}
static inline ImgU8 stackVertical2(const ImgU8& a, const ImgU8& b) {
  ImgU8 d(a.w, a.h + b.h, a.c);
  std::memcpy(d.d.data(), a.d.data(), a.bytes());
  std::memcpy(d.row(a.h), b.d.data(), b.bytes());
  return d;
}
// TRSP:701-713
static inline ImgU8 padToHeight(const ImgU8& s, int targetHeight) {
  const int above = (targetHeight - s.h) / 2;
  return padRows(s, above, targetHeight - s.h - above);
}

// CvUtil.cpp:140-157
static inline ImgU8 featherAlphaChannel(const ImgU8& src, int erodeSize) {
  ImgU8 a(src.w, src.h, 1);
  for (size_t i = 0, n = size_t(src.w) * src.h; i < n; ++i) a.d[i] = src.d[i * 4 + 3];
  a = erodeCrossU8C1(a, erodeSize);
  a = gaussianBlurU8C1(a, erodeSize, erodeSize / 2.0f);
  ImgU8 d = src;
  for (size_t i = 0, n = size_t(src.w) * src.h; i < n; ++i) d.d[i * 4 + 3] = a.d[i];
  return d;
}

// CvUtil.cpp:224-260
static inline ImgU8 flattenLayersDeghostPreferBase(const ImgU8& bottomLayer, const ImgU8& topLayer) {
  ImgU8 m(bottomLayer.w, bottomLayer.h, 4);
  for (int y = 0; y < m.h; ++y)
    for (int x = 0; x < m.w; ++x) {
      const uint8_t* base = bottomLayer.px(y, x);
      const uint8_t* top = topLayer.px(y, x);
      const float colorDiff =
          (std::abs(base[0] - top[0]) + std::abs(base[1] - top[1]) + std::abs(base[2] - top[2])) / 255.0f;
      const float kColorDiffCoef = 5.0f, kSoftmaxSharpness = 5.0f, kBaseLayerBias = 2.0f;
      const float deghostCoef = tanhf(colorDiff * kColorDiffCoef);
      const float alphaR = top[3] / 255.0f;
      const float alphaL = 1.0f - alphaR;
      // CvUtil.cpp has `using namespace std`, the arguments are float: overload resolution picks
      // std::exp(float), i.e. expf, and the result is widened to double.
      const double expL = (double)expf(kSoftmaxSharpness * alphaL * kBaseLayerBias);
      const double expR = (double)expf(kSoftmaxSharpness * alphaR);
      const double sumExp = expL + expR + 0.00001;
      const float softmaxL = float(expL / sumExp);
      const float softmaxR = 1.0f - softmaxL;
      const float wL = lerpf(alphaL, softmaxL, deghostCoef), wR = lerpf(alphaR, softmaxR, deghostCoef);
      uint8_t* o = m.px(y, x);
      o[0] = truncU8(float(base[0]) * wL + float(top[0]) * wR);
      o[1] = truncU8(float(base[1]) * wL + float(top[1]) * wR);
      o[2] = truncU8(float(base[2]) * wL + float(top[2]) * wR);
      o[3] = std::max(top[3], base[3]);
    }
  return m;
}

// ---------------------------------------------------------------------------
// flow_bottom_secondary.bin + flow_images/bottomImage{,2}.png (PoleRemoval.cpp:95-126)
struct PoleRemovalState {
  ImgF flow;
  ImgU8 bottomImage, bottomImage2;
  bool valid = false;
};
// the secondary bottom camera's image and the two red pole masks (PoleRemoval.cpp:48-66)
struct PoleRemovalInput {
  ImgU8 bottom2, mask, mask2;  // BGR
};

// CvUtil.cpp:201-211
static inline void circleAlphaCut(ImgU8& imageBGRA, float radius) {
  for (int y = 0; y < imageBGRA.h; ++y)
    for (int x = 0; x < imageBGRA.w; ++x) {
      const float dx = float(x) - float(imageBGRA.w) / 2.0f;
      const float dy = float(y) - float(imageBGRA.h) / 2.0f;
      const float r = sqrtf(dx * dx + dy * dy);
      const float alpha = r < radius ? 1.0f : 0.0f;
      imageBGRA.at(y, x, 3) = (unsigned char)(alpha * 255.0f);
    }
}
// CvUtil.cpp:213-222
static inline void cutRedMaskOutOfAlphaChannel(ImgU8& destBGRA, const ImgU8& redMaskBGR) {
  assert(destBGRA.w == redMaskBGR.w && destBGRA.h == redMaskBGR.h);
  for (int y = 0; y < redMaskBGR.h; ++y)
    for (int x = 0; x < redMaskBGR.w; ++x)
      if (redMaskBGR.at(y, x, 0) == 0 && redMaskBGR.at(y, x, 1) == 0 && redMaskBGR.at(y, x, 2) == 255)
        destBGRA.at(y, x, 3) = 0;
}
// combineBottomImagesWithPoleRemoval (PoleRemoval.cpp:32-188) on decoded images; returns the merged BGRA bottom image.
static inline ImgU8 combineBottomImagesWithPoleRemoval(const RigDescription& rig, const RenderParams& P,
                                                       const ImgU8& bottomBGR, const PoleRemovalInput& in,
                                                       const PoleRemovalState* prev, PoleRemovalState* state) {
  const Camera& cam = rig.findCameraByDirection(V3(0, 0, -1));  // TRSP:576-580
  const Camera& cam2 = rig.findLargestDistCamAxisToRigCenter();
  const float radius = approximateUsablePixelsRadius(cam), radius2 = approximateUsablePixelsRadius(cam2);
  const bool flip180 = cam.up().dot(cam2.up()) < 0;
  ImgU8 bottomImage = bgr2bgra(bottomBGR), bottomImage2 = bgr2bgra(in.bottom2);
  circleAlphaCut(bottomImage, radius);
  circleAlphaCut(bottomImage2, radius2);
  cutRedMaskOutOfAlphaChannel(bottomImage, in.mask);
  cutRedMaskOutOfAlphaChannel(bottomImage2, in.mask2);
  bottomImage = featherAlphaChannel(bottomImage, P.std_alpha_feather_size);
  bottomImage2 = featherAlphaChannel(bottomImage2, P.std_alpha_feather_size);
  if (flip180) bottomImage2 = flipBoth(bottomImage2);
  PixFlowParams fp;
  pixflowParamsByName(P.poleremoval_flow_alg, &fp);
  PixFlow pf(fp);
  ImgF flow;
  if (prev && prev->valid)
    pf.computeOpticalFlow(bottomImage, bottomImage2, prev->flow, prev->bottomImage, prev->bottomImage2, flow, HINT_DOWN);
  else
    pf.computeOpticalFlow(bottomImage, bottomImage2, ImgF(), ImgU8(), ImgU8(), flow, HINT_DOWN);
  if (state) { state->flow = flow; state->bottomImage = bottomImage; state->bottomImage2 = bottomImage2; state->valid = true; }
  ImgF warp(bottomImage.w, bottomImage.h, 2);
  for (int y = 0; y < warp.h; ++y)
    for (int x = 0; x < warp.w; ++x) {
      warp.at(y, x, 0) = float(x) + flow.at(y, x, 0);
      warp.at(y, x, 1) = float(y) + flow.at(y, x, 1);
    }
  const ImgU8 warped2 = remapCubicU8(bottomImage2, warp);
  for (int y = 0; y < bottomImage.h; ++y)
    for (int x = 0; x < bottomImage.w; ++x) {
      uint8_t* p1 = bottomImage.px(y, x);
      const uint8_t* p2 = warped2.px(y, x);
      const float alpha = p1[3] / 255.0f, alpha2 = p2[3] / 255.0f;
      if (alpha < 1.0f && alpha2 > 0.0f) {
        const float a1 = alpha, a2 = 1.0f - alpha;
        const float r1 = p1[2], g1 = p1[1], b1 = p1[0], r2 = p2[2], g2 = p2[1], b2 = p2[0];
        p1[0] = (unsigned char)(a1 * b1 + a2 * b2);
        p1[1] = (unsigned char)(a1 * g1 + a2 * g2);
        p1[2] = (unsigned char)(a1 * r1 + a2 * r2);
        p1[3] = 255;
      }
    }
  circleAlphaCut(bottomImage, radius);
  return featherAlphaChannel(bottomImage, P.std_alpha_feather_size);
}

// TRSP:647-685 / 564-644 (no pole removal): pole camera -> spherical BGRA with
// the bottom-rows alpha feather.
static inline ImgU8 preparePoleImage(const ImgU8& img, const Camera& cam, const RenderParams& P, bool isTop) {
  const int rows = int(P.eqr_height * cam.getFov() / M_PI);
  ImgU8 sph;
  if (isTop)
    sph = bicubicRemapToSpherical(P.eqr_width, rows, 3, img, cam, (float)(2.0f * M_PI), 0.f, (float)(M_PI / 2.0f),
                                  (float)(M_PI / 2.0f - cam.getFov()));
  else
    sph = bicubicRemapToSpherical(P.eqr_width, rows, img.c, img, cam, 0.f, (float)(2.0f * M_PI), (float)(-(M_PI / 2.0f)),
                                  (float)(-(M_PI / 2.0f - cam.getFov())));
  if (sph.c != 4) sph = bgr2bgra(sph);  // TRSP:621-623: the pole-removal result already has an alpha channel
  const int yFeatherStart = sph.h - 1 - P.std_alpha_feather_size;
  for (int y = yFeatherStart; y < sph.h; ++y)
    for (int x = 0; x < sph.w; ++x) {
      const float alpha = 1.0f - float(y - yFeatherStart) / float(P.std_alpha_feather_size);
      const uint8_t a = (uint8_t)(255.0f * alpha);
      sph.at(y, x, 3) = isTop ? a : std::min(sph.at(y, x, 3), a);
    }
  return sph;
}

// Ramp constants of TRSP:454-481.
struct PoleRamp { float poleCameraRadius, phiRampStart, phiMid, phiRampEnd; };
static inline PoleRamp poleRamp(const RigDescription& rig) {
  float poleCameraRadius = (float)rig.findCameraByDirection(V3(0, 0, -1)).getFov();
  float sideCameraRadius = approximateFov(rig.rigSideOnly, true);
  float poleCameraCropRadius =
      (float)(0.5f * (M_PI / 2 - sideCameraRadius) + 0.5f * (std::min(float(M_PI / 2), poleCameraRadius)));
  poleCameraCropRadius = (float)(poleCameraCropRadius * (180 / M_PI));
  poleCameraRadius = (float)(poleCameraRadius * (180 / M_PI));
  sideCameraRadius = (float)(sideCameraRadius * (180 / M_PI));
  const float kRampFrac = 1.0f;
  const float phiFromPole = poleCameraCropRadius;
  const float phiFromSide = 90.0f - sideCameraRadius;
  PoleRamp r;
  r.poleCameraRadius = poleCameraRadius;
  r.phiMid = (phiFromPole + phiFromSide) / 2.0f;
  const float phiDiff = fabsf(phiFromPole - phiFromSide);
  r.phiRampStart = r.phiMid - kRampFrac * phiDiff / 2.0f;
  r.phiRampEnd = r.phiMid + kRampFrac * phiDiff / 2.0f;
  return r;
}

struct PoleFlowState {  // what the reference persists for the next frame (TRSP:413-452)
  ImgU8 extendedSide, extendedFisheye;
  ImgF flow;
};

// TRSP:388-561. sideSpherical is the full-height (eqr_h) eye panorama (already
// flipped for the bottom pole); fisheyeSpherical the pole image. prev may be null.
static inline ImgU8 poleToSideFlow(const RigDescription& rig, const RenderParams& P, const ImgU8& sideSpherical,
                                   const ImgU8& fisheyeSpherical, const PoleFlowState* prev, PoleFlowState* state) {
  const int cols = fisheyeSpherical.w, rows = fisheyeSpherical.h;
  ImgU8 cropped = cropRows(sideSpherical, 0, rows);
  cropped = featherAlphaChannel(cropped, P.std_alpha_feather_size);
  const float kExtendFrac = 1.2f;
  const int extendedWidth = int(float(cols) * kExtendFrac);
  ImgU8 extSide(extendedWidth, rows, 4), extFish(extendedWidth, rows, 4);
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < extendedWidth; ++x) {
      std::memcpy(extSide.px(y, x), cropped.px(y, x % cols), 4);
      std::memcpy(extFish.px(y, x), fisheyeSpherical.px(y, x % cols), 4);
    }
  PixFlowParams fp;
  pixflowParamsByName(P.polar_flow_alg, &fp);
  PixFlow pf(fp);
  ImgF flow;
  if (prev) pf.computeOpticalFlow(extSide, extFish, prev->flow, prev->extendedSide, prev->extendedFisheye, flow, HINT_DOWN);
  else pf.computeOpticalFlow(extSide, extFish, ImgF(), ImgU8(), ImgU8(), flow, HINT_DOWN);

  const PoleRamp R = poleRamp(rig);
  ImgF warp(extendedWidth, rows, 2);
  for (int y = 0; y < rows; ++y) {
    const float phi = R.poleCameraRadius * float(y + 0.5f) / float(rows);
    const float alpha = 1.0f - rampf(phi, R.phiRampStart, R.phiMid);
    for (int x = 0; x < extendedWidth; ++x) {
      warp.at(y, x, 0) = float(x) + (1.0f - alpha) * flow.at(y, x, 0);
      warp.at(y, x, 1) = float(y) + (1.0f - alpha) * flow.at(y, x, 1);
    }
  }
  ImgU8 warpedExt = remapCubicU8(extFish, warp);
  ImgU8 warped = cropCols(warpedExt, 0, cols);
  const int maxBlendX = int(float(cols) * (kExtendFrac - 1.0f));
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < maxBlendX; ++x) {
      const uint8_t* s = warped.px(y, x);
      const uint8_t* wr = warpedExt.px(y, x + cols);
      const float srcB = s[0], srcG = s[1], srcR = s[2], srcA = s[3];
      const float wrapB = wr[0], wrapG = wr[1], wrapR = wr[2];
      const float alpha = 1.0f - rampf(float(x), float(maxBlendX) * 0.333f, float(maxBlendX) * 0.667f);
      uint8_t* o = warped.px(y, x);
      o[0] = truncU8(wrapB * alpha + srcB * (1.0f - alpha));
      o[1] = truncU8(wrapG * alpha + srcG * (1.0f - alpha));
      o[2] = truncU8(wrapR * alpha + srcR * (1.0f - alpha));
      o[3] = truncU8(srcA);
    }
  for (int y = 0; y < rows; ++y) {
    const float phi = R.poleCameraRadius * float(y + 0.5f) / float(rows);
    const float alpha = 1.0f - rampf(phi, R.phiMid, R.phiRampEnd);
    for (int x = 0; x < cols; ++x) warped.at(y, x, 3) = truncU8(float(warped.at(y, x, 3)) * alpha);
  }
  if (state) {
    state->extendedSide = std::move(extSide);
    state->extendedFisheye = std::move(extFish);
    state->flow = std::move(flow);
  }
  return padRows(warped, 0, sideSpherical.h - rows);
}

// ---------------------------------------------------------------------------
// Filter.h:40-127 + TRSP:688-696 (sharpen; "next" row in SURVEY §8f).
static inline int wrapI(int x, int r) { return x < 0 ? r + x : x >= r ? x - r : x; }
static inline int reflectI(int x, int r) { return x < 0 ? -x : x >= r ? 2 * r - x - 2 : x; }
static inline void sharpen(ImgU8& img /*BGR*/, float sharpening) {
  const float alpha = powf(0.25f, 1.0f / 4.0f);
  const int width = img.w, height = img.h;
  ImgU8 lp(width, height, 3);
  std::vector<float> buf(size_t(std::max(width, height)) * 3);
  auto clampf = [](float v) { return v < 0.0f ? 0.0f : v > 255.0f ? 255.0f : v; };
  for (int i = 0; i < height; ++i) {
    float v[3] = {float(img.at(i, 0, 0)), float(img.at(i, 0, 1)), float(img.at(i, 0, 2))};
    for (int j = 1; j <= width; ++j) {
      const uint8_t* ip = img.px(i, wrapI(j, width));
      for (int k = 0; k < 3; ++k) v[k] = float(ip[k]) * (1.0f - alpha) + v[k] * alpha;
      float* b = &buf[size_t(wrapI(j - 1, width)) * 3];
      b[0] = v[0]; b[1] = v[1]; b[2] = v[2];
    }
    for (int j = width - 2; j >= -1; --j) {
      const float* ip = &buf[size_t(wrapI(j, width)) * 3];
      for (int k = 0; k < 3; ++k) v[k] = ip[k] * (1.0f - alpha) + v[k] * alpha;
      for (int k = 0; k < 3; ++k) lp.at(i, j + 1, k) = (uint8_t)clampf(v[k]);
    }
  }
  for (int j = 0; j < width; ++j) {
    float v[3] = {float(lp.at(0, j, 0)), float(lp.at(0, j, 1)), float(lp.at(0, j, 2))};
    for (int i = 1; i <= height; ++i) {
      const uint8_t* ip = lp.px(reflectI(i, height), j);
      for (int k = 0; k < 3; ++k) v[k] = float(ip[k]) * (1.0f - alpha) + v[k] * alpha;
      float* b = &buf[size_t(reflectI(i - 1, height)) * 3];
      b[0] = v[0]; b[1] = v[1]; b[2] = v[2];
    }
    for (int i = height - 2; i >= -1; --i) {
      const float* ip = &buf[size_t(reflectI(i, height)) * 3];
      for (int k = 0; k < 3; ++k) v[k] = ip[k] * (1.0f - alpha) + v[k] * alpha;
      for (int k = 0; k < 3; ++k) lp.at(i + 1, j, k) = (uint8_t)clampf(v[k]);
    }
  }
  const float amount = 1.0f + sharpening, noiseCore = 100.0f;
  for (int i = 0; i < height; ++i)
    for (int j = 0; j < width; ++j)
      for (int k = 0; k < 3; ++k) {
        const float l = float(lp.at(i, j, k));
        const float hp = float(img.at(i, j, k)) - l;
        const float ng = 1.0f - expf(-((hp * hp) * noiseCore));
        img.at(i, j, k) = (uint8_t)clampf(l + hp * ng * amount);
      }
}

// ---------------------------------------------------------------------------
// Temporal state the reference writes under output_data_dir for the next frame
// (TRSP:201-255, 413-452).
struct FrameState {
  PoleRemovalState poleRemoval;
  std::vector<ImgU8> overlapL, overlapR;
  std::vector<ImgF> flowLtoR, flowRtoL;
  PoleFlowState pole[4];  // top_left, top_right, bottom_left, bottom_right
  bool valid = false;
};

struct FrameDebug {  // intermediates exposed for stage-by-stage parity tests
  std::vector<ImgU8> projections;
  ImgU8 sidePanoL, sidePanoR;  // after offsetHorizontalWrap + padToheight
  ImgU8 topSpherical, bottomSpherical;
  ImgU8 poleWarped[4];
  ImgU8 eyeL, eyeR;  // BGR before final resize
};

// TRSP:716-972 renderStereoPanorama (no pole removal, no cubemap).
// sideImages: numCams BGR (or BGRA) images; top/bottom BGR images (may be empty
// if disabled). nthreads>1 uses the reference's thread fan-out.
static inline ImgU8 renderStereoPanorama(const RigDescription& rig, const RenderParams& P,
                                         const std::vector<ImgU8>& sideImages, const ImgU8& topImage,
                                         const ImgU8& bottomImage, const FrameState* prev, FrameState* state,
                                         FrameDebug* dbg, bool threaded, double* stageSec /*[5] or null*/,
                                         const PoleRemovalInput* poleRemoval = nullptr) {
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const int numCams = (int)rig.rigSideOnly.size();
  const SideGeometry g = sideGeometry(rig, P);
  const double t0 = now();
  // pole projections (threads joined as late as possible in the reference)
  ImgU8 topSph, botSph;
  std::thread topThread, botThread;
  if (P.enable_bottom) {
    auto f = [&] {
      if (P.enable_pole_removal && poleRemoval) {  // TRSP:569-597
        const ImgU8 merged = combineBottomImagesWithPoleRemoval(rig, P, bottomImage, *poleRemoval,
                                                                prev ? &prev->poleRemoval : nullptr,
                                                                state ? &state->poleRemoval : nullptr);
        botSph = preparePoleImage(merged, rig.findCameraByDirection(V3(0, 0, -1)), P, false);
      } else {
        botSph = preparePoleImage(bottomImage, rig.findCameraByDirection(V3(0, 0, -1)), P, false);
      }
    };
    if (threaded) botThread = std::thread(f); else f();
  }
  if (P.enable_top) {
    auto f = [&] { topSph = preparePoleImage(topImage, rig.findCameraByDirection(V3(0, 0, 1)), P, true); };
    if (threaded) topThread = std::thread(f); else f();
  }
  // side projections (TRSP:138-186)
  std::vector<ImgU8> proj(numCams);
  {
    std::vector<std::thread> th;
    for (int i = 0; i < numCams; ++i) {
      auto f = [&, i] {
        float l, r, t, b;
        sideCameraAngles(g, i, numCams, &l, &r, &t, &b);
        proj[i] = projectSideToSpherical(g.camImageWidth, g.camImageHeight, sideImages[i], rig.rigSideOnly[i], l, r,
                                         t, b, P.side_alpha_feather_size);
      };
      if (threaded) th.emplace_back(f); else f();
    }
    for (auto& t : th) t.join();
  }
  const double t1 = now();
  // flows (TRSP:189-256, 320-335)
  PixFlowParams fp;
  pixflowParamsByName(P.side_flow_alg, &fp);
  std::vector<ImgU8> ovL(numCams), ovR(numCams);
  std::vector<ImgF> fLR(numCams), fRL(numCams);
  {
    std::vector<std::thread> th;
    for (int i = 0; i < numCams; ++i) {
      auto f = [&, i] {
        const int r = (i + 1) % numCams;
        ovL[i] = cropCols(proj[i], proj[i].w - g.overlapImageWidth, g.overlapImageWidth);
        ovR[i] = cropCols(proj[r], 0, g.overlapImageWidth);
        PixFlow pf(fp);
        if (prev && prev->valid) {
          // NovelView.cpp:282-297
          pf.computeOpticalFlow(ovL[i], ovR[i], prev->flowLtoR[i], prev->overlapL[i], prev->overlapR[i], fLR[i], HINT_LEFT);
          pf.computeOpticalFlow(ovR[i], ovL[i], prev->flowRtoL[i], prev->overlapR[i], prev->overlapL[i], fRL[i], HINT_RIGHT);
        } else {
          pf.computeOpticalFlow(ovL[i], ovR[i], ImgF(), ImgU8(), ImgU8(), fLR[i], HINT_LEFT);
          pf.computeOpticalFlow(ovR[i], ovL[i], ImgF(), ImgU8(), ImgU8(), fRL[i], HINT_RIGHT);
        }
      };
      if (threaded) th.emplace_back(f); else f();
    }
    for (auto& t : th) t.join();
  }
  const double t2 = now();
  // novel views (TRSP:354-384)
  std::vector<ImgU8> chunksL(numCams), chunksR(numCams);
  {
    std::vector<std::thread> th;
    for (int i = 0; i < numCams; ++i) {
      auto f = [&, i] {
        auto lr = combineLazyNovelViews(g, P.eqr_width, numCams, ovL[i], ovR[i], fLR[i], fRL[i]);
        chunksL[i] = std::move(lr.first);
        chunksR[i] = std::move(lr.second);
      };
      if (threaded) th.emplace_back(f); else f();
    }
    for (auto& t : th) t.join();
  }
  ImgU8 panoL = offsetHorizontalWrap(stackHorizontal(chunksL), g.zeroParallaxNovelViewShiftPixels);
  ImgU8 panoR = offsetHorizontalWrap(stackHorizontal(chunksR), -g.zeroParallaxNovelViewShiftPixels);
  const double t3 = now();
  panoL = padToHeight(panoL, P.eqr_height);
  panoR = padToHeight(panoR, P.eqr_height);
  if (dbg) { dbg->projections = proj; dbg->sidePanoL = panoL; dbg->sidePanoR = panoR; }
  if (state) { state->overlapL = ovL; state->overlapR = ovR; state->flowLtoR = fLR; state->flowRtoL = fRL; }

  // poles (TRSP:811-885)
  ImgU8 warped[4];
  std::thread pth[4];
  ImgU8 flipL, flipR;
  if (P.enable_top) {
    if (topThread.joinable()) topThread.join();
    auto fl = [&] { warped[0] = poleToSideFlow(rig, P, panoL, topSph, (prev && prev->valid) ? &prev->pole[0] : nullptr, state ? &state->pole[0] : nullptr); };
    auto fr = [&] { warped[1] = poleToSideFlow(rig, P, panoR, topSph, (prev && prev->valid) ? &prev->pole[1] : nullptr, state ? &state->pole[1] : nullptr); };
    if (threaded) { pth[0] = std::thread(fl); pth[1] = std::thread(fr); } else { fl(); fr(); }
  }
  if (P.enable_bottom) {
    if (botThread.joinable()) botThread.join();
    flipL = flipBoth(panoL);
    flipR = flipBoth(panoR);
    auto fl = [&] { warped[2] = poleToSideFlow(rig, P, flipL, botSph, (prev && prev->valid) ? &prev->pole[2] : nullptr, state ? &state->pole[2] : nullptr); };
    auto fr = [&] { warped[3] = poleToSideFlow(rig, P, flipR, botSph, (prev && prev->valid) ? &prev->pole[3] : nullptr, state ? &state->pole[3] : nullptr); };
    if (threaded) { pth[2] = std::thread(fl); pth[3] = std::thread(fr); } else { fl(); fr(); }
  }
  if (P.enable_top) {
    if (pth[0].joinable()) pth[0].join();
    if (pth[1].joinable()) pth[1].join();
    panoL = flattenLayersDeghostPreferBase(panoL, warped[0]);
    panoR = flattenLayersDeghostPreferBase(panoR, warped[1]);
  }
  if (P.enable_bottom) {
    if (pth[2].joinable()) pth[2].join();
    if (pth[3].joinable()) pth[3].join();
    panoL = flipBoth(flattenLayersDeghostPreferBase(flipBoth(panoL), warped[2]));
    panoR = flipBoth(flattenLayersDeghostPreferBase(flipBoth(panoR), warped[3]));
  }
  const double t4 = now();
  if (dbg) {
    dbg->topSpherical = topSph; dbg->bottomSpherical = botSph;
    for (int i = 0; i < 4; ++i) dbg->poleWarped[i] = warped[i];
  }
  ImgU8 eyeL = bgra2bgr(panoL), eyeR = bgra2bgr(panoR);
  if (P.sharpening > 0.0) {
    if (threaded) {
      std::thread a([&] { sharpen(eyeL, (float)P.sharpening); }), b([&] { sharpen(eyeR, (float)P.sharpening); });
      a.join(); b.join();
    } else { sharpen(eyeL, (float)P.sharpening); sharpen(eyeR, (float)P.sharpening); }
  }
  if (dbg) { dbg->eyeL = eyeL; dbg->eyeR = eyeR; }
  if (P.final_eqr_width != 0 && P.final_eqr_height != 0 && P.final_eqr_width != P.eqr_width &&
      P.final_eqr_height != P.eqr_height / 2) {  // TRSP:938-957
    eyeL = resizeCubicU8(eyeL, P.final_eqr_width, P.final_eqr_height / 2);
    eyeR = resizeCubicU8(eyeR, P.final_eqr_width, P.final_eqr_height / 2);
  }
  ImgU8 out = stackVertical2(eyeL, eyeR);
  if (state) state->valid = true;
  const double t5 = now();
  if (stageSec) {  // same buckets as TRSP:964-971
    stageSec[0] = t1 - t0; stageSec[1] = t2 - t1; stageSec[2] = t3 - t2; stageSec[3] = t4 - t3; stageSec[4] = t5 - t0;
  }
  return out;
}

// ---- cubemap output (ImageWarper.cpp:26-141, CvUtil.cpp:117-138, TRSP:917-935) --------------------------------
// Faces in the reference's order: RIGHT, LEFT, TOP, BOTTOM, BACK, FRONT (enum CubemapFace values 0..5 are
// BACK, LEFT, TOP, BOTTOM, FRONT, RIGHT in ImageWarper.h; only the switch below depends on them).
enum CubeFace { CUBE_BACK = 0, CUBE_LEFT, CUBE_TOP, CUBE_BOTTOM, CUBE_FRONT, CUBE_RIGHT };
static inline void cubemapIndexToVec3(float x, float y, int face, float out[3]) {  // ImageWarper.cpp:26-61
  const float dir[3] = {x, y, 0.5f};
  out[0] = dir[0]; out[1] = dir[1]; out[2] = dir[2];
  switch (face) {
    case CUBE_BACK: out[0] = dir[0]; out[1] = dir[2]; out[2] = -dir[1]; break;
    case CUBE_LEFT: out[0] = -dir[2]; out[1] = dir[0]; out[2] = -dir[1]; break;
    case CUBE_TOP: break;
    case CUBE_BOTTOM: out[0] = dir[0]; out[1] = -dir[1]; out[2] = -dir[2]; break;
    case CUBE_FRONT: out[0] = -dir[0]; out[1] = -dir[2]; out[2] = -dir[1]; break;
    case CUBE_RIGHT: out[0] = dir[2]; out[1] = -dir[0]; out[2] = -dir[1]; break;
  }
}
static inline void mapEquirectToCubemapCoordinate(float x, float y, int face, int srcCols, int srcRows,
                                                  float fisheyeFovRadians, float* srcX, float* srcY) {  // :63-93
  float dir[3];
  cubemapIndexToVec3(x, y, face, dir);
  const float r = sqrtf(dir[0] * dir[0] + dir[1] * dir[1]);
  // cv::norm(Vec3f): float accumulation of the squares, sqrt, returned as double
  float s2 = 0.f;
  for (int i = 0; i < 3; ++i) s2 += dir[i] * dir[i];
  const double nrm = (double)std::sqrt(s2);
  const float phi = acosf((float)((double)dir[2] / nrm));
  float theta = r > 0.0f ? acosf(std::fabs(dir[0] / r)) : 0.0f;
  if (dir[0] > 0 && dir[1] > 0) {
  } else if (dir[0] <= 0 && dir[1] > 0) {
    theta = (float)(M_PI - theta);
  } else if (dir[0] <= 0 && dir[1] <= 0) {
    theta = (float)(M_PI + theta);
  } else {
    theta = (float)(2 * M_PI - theta);
  }
  const float phiPrime = std::min(std::max(phi, 0.0f), fisheyeFovRadians);
  const float thetaPrime = std::min(std::max(theta, 0.0f), float(2.0f * M_PI));
  *srcX = (float)(float(srcCols) * thetaPrime / (2.0f * M_PI));
  *srcY = float(srcRows) * phiPrime / fisheyeFovRadians;
}
static inline ImgF cubemapWarpMap(int face, int srcCols, int srcRows, float fov, int faceW, int faceH) {  // :111-128
  const float dy = 1.0f / float(faceW), dx = 1.0f / float(faceH);
  ImgF m(faceW, faceH, 2);
  for (int j = 0; j < faceH; ++j)
    for (int i = 0; i < faceW; ++i)
      mapEquirectToCubemapCoordinate(float(i) * dy - 0.5f, float(j) * dx - 0.5f, face, srcCols, srcRows, fov,
                                     &m.at(j, i, 0), &m.at(j, i, 1));
  return m;
}
static inline ImgU8 flipHorizontal(const ImgU8& s) {
  ImgU8 d(s.w, s.h, s.c);
  for (int y = 0; y < s.h; ++y)
    for (int x = 0; x < s.w; ++x)
      for (int k = 0; k < s.c; ++k) d.at(y, x, k) = s.at(y, s.w - 1 - x, k);
  return d;
}
// convertSphericalToCubemapBicubicRemap + stackOutputCubemapFaces for one eye (format "video" or "photo")
static inline ImgU8 cubemapOfEye(const ImgU8& eyeBGR, int faceW, int faceH, const std::string& format) {
  static const int faces[6] = {CUBE_RIGHT, CUBE_LEFT, CUBE_TOP, CUBE_BOTTOM, CUBE_BACK, CUBE_FRONT};
  std::vector<ImgU8> img(6);
  for (int f = 0; f < 6; ++f)
    img[f] = remapCubicU8Wrap(eyeBGR, cubemapWarpMap(faces[f], eyeBGR.w, eyeBGR.h, (float)M_PI, faceW, faceH));
  if (format == "video") {
    const ImgU8 a = stackHorizontal({flipHorizontal(img[1]), flipHorizontal(img[0]), flipHorizontal(img[2])});
    const ImgU8 b = stackHorizontal({flipHorizontal(img[3]), flipHorizontal(img[4]), flipHorizontal(img[5])});
    return stackVertical2(a, b);
  }
  ImgU8 d = img[0];
  for (int f = 1; f < 6; ++f) d = stackVertical2(d, img[f]);
  return d;
}
static inline ImgU8 stereoCubemap(const ImgU8& eyeL, const ImgU8& eyeR, int faceW, int faceH, const std::string& format) {
  return stackVertical2(cubemapOfEye(eyeL, faceW, faceH, format), cubemapOfEye(eyeR, faceW, faceH, format));
}

}  // namespace orc
