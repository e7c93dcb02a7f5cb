// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
//
// isp.h: CPU restatement of the reference's soft ISP, the non-accelerated path of Raw2Rgb
// (surround360_render/source/camera_isp/Raw2Rgb.cpp:441-456 -> camera_isp/CameraIsp.h): 16-bit Bayer raw ->
// black level, anti-vignetting, white balance, clamp + stretch, demosaic (bilinear or edge-aware), composite CCM +
// tone-curve LUT, IIR unsharp mask, 8- or 16-bit BGR.
//
// PARITY PINNED: unlike the OpenCV-dependent parts of the oracle, this restatement is checked bit for bit against the
// reference's own source compiled from /root/reference (oracle/_ref/libref_isp.so = CameraIsp.h + JsonUtil.cpp +
// supereasyjson over a container-only OpenCV stand-in, oracle/ref_isp.cpp) by tests/test_cpu_isp.py, and against the
// committed outputs of that library (tests/golden/isp_golden.npz) where /root/reference is absent.
//
// Not restated: the DCT demosaic (FREQUENCY_DM_FILTER needs cv::dct). Stuck-pixel removal with a non-zero radius IS
// (round 3; formerly:
// (a serial in-place pass over an unstable sort, CameraIsp.h:1024-1104; radius 0 in every shipped configuration).
// Build with -ffp-contract=off like the rest of the oracle.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <vector>

namespace orc {

constexpr int kIspToneLutSize = 4096;  // kToneCurveLutSize, CameraIsp.h:42
constexpr int kIspMaxCurvePoints = 16;

// The fields of the "CameraIsp" JSON object that influence pixels, as the constructor stores them (doubles narrowed
// to float, CameraIsp.h:425-607), plus the Raw2Rgb flags.
struct IspConfig {
  float blackLevel[3] = {0, 0, 0};
  float clampMin[3] = {0, 0, 0}, clampMax[3] = {1, 1, 1};
  float whiteBalanceGain[3] = {1, 1, 1};
  float ccm[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  float saturation = 1.0f, contrast = 1.0f;
  float gamma[3] = {1, 1, 1}, lowKeyBoost[3] = {0, 0, 0}, highKeyBoost[3] = {0, 0, 0};
  float sharpening[3] = {0, 0, 0};
  float sharpeningSupport = 10.0f / 2048.0f, noiseCore = 1000.0f;
  int nVignetteH = 1, nVignetteV = 1;
  float vignetteRollOffH[kIspMaxCurvePoints][3] = {{1, 1, 1}}, vignetteRollOffV[kIspMaxCurvePoints][3] = {{1, 1, 1}};
  int stuckPixelRadius = 0;  // 2 x the JSON value (CameraIsp.h:512)
  int bayerPattern = 2;      // 0 RGGB, 1 GRBG, 2 GBRG (default), 3 BGGR
  // Raw2Rgb flags
  int outputBpp = 8, demosaicFilter = 2 /*EDGE_AWARE*/, resize = 1, disableToneCurve = 0, blackLevelOffset = 0;
  // removeStuckPixels (CameraIsp.h:1024-1104), used when stuckPixelRadius > 0
  int stuckPixelThreshold = 0;
  float stuckPixelDarknessThreshold = 0.0f;
};

namespace isp_detail {
inline float clampf(float x, float a, float b) { return x < a ? a : x > b ? b : x; }  // MathUtil.h:38-41
inline int reflecti(int x, int r) { return x < 0 ? -x : x >= r ? 2 * r - x - 2 : x; }  // MathUtil.h:43-46
inline float lerpf(float x0, float x1, float a) { return x0 * (1.0f - a) + x1 * a; }   // MathUtil.h:58-61
inline float bilerpf(float x00, float x01, float x10, float x11, float a, float b) {   // MathUtil.h:63-76
  return lerpf(lerpf(x00, x01, a), lerpf(x10, x11, a), b);
}
struct V3 { float v[3]; };
inline V3 lerp3(const V3& a, const V3& b, float t) {  // lerp<Vec3f, float>: per element
  V3 r;
  for (int k = 0; k < 3; ++k) r.v[k] = a.v[k] * (1.0f - t) + b.v[k] * t;
  return r;
}
// BezierCurve<float, Vec3f>::operator()(i, j, t): De Casteljau by recursion (MathUtil.h:205-213)
inline V3 bezier(const float (*p)[3], int i, int j, float t) {
  if (i == j) return V3{{p[i][0], p[i][1], p[i][2]}};
  return lerp3(bezier(p, i, j - 1, t), bezier(p, i + 1, j, t), t);
}
// bezier / highKey / lowKey of the tone curve (CameraIsp.h:361-388)
inline float bezier4(float a, float b, float c, float d, float t) {
  return lerpf(lerpf(lerpf(a, b, t), lerpf(b, c, t), t), lerpf(lerpf(b, c, t), lerpf(c, d, t), t), t);
}
inline float highKey(float boost, float x) {
  const float a = 0.5f, b = clampf(0.6666f, 0.0f, 1.0f), c = clampf(0.8333f + boost, 0.0f, 1.0f), d = 1.0f;
  return x > 0.5f ? bezier4(a, b, c, d, (x - 0.5f) * 2.0f) : 0;
}
inline float lowKey(float boost, float x) {
  const float a = 0.0f, b = clampf(0.1666f + boost, 0.0f, 1.0f), c = clampf(0.3333f, 0.0f, 1.0f), d = 0.5f;
  return x <= 0.5f ? bezier4(a, b, c, d, x * 2.0f) : 0;
}
// 3x3 float product as the OpenCV stand-in defines it (oracle/ref_shim/opencv2/core.hpp: float products summed left
// to right in float)
inline void mul33(const float* a, const float* b, float* d) {
  float t[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float s = a[i * 3] * b[j];
      s = s + a[i * 3 + 1] * b[3 + j];
      s = s + a[i * 3 + 2] * b[6 + j];
      t[i * 3 + j] = s;
    }
  for (int i = 0; i < 9; ++i) d[i] = t[i];
}
}  // namespace isp_detail

// Everything CameraIsp::setup() / buildToneCurveLut() derive from the configuration (host-side in the product too).
struct IspTables {
  bool red[2][2], green[2][2];  // Bayer pattern tables (CameraIsp.h:613-660)
  float compositeCCM[9];        // transpose(ccm) * (yuv2rgb * sat * rgb2yuv) * 4095 (CameraIsp.h:671-686)
  std::vector<float> toneLut;   // [4096][3] (CameraIsp.h:390-425)
};

inline IspTables ispSetup(const IspConfig& c) {
  using namespace isp_detail;
  IspTables t;
  static const bool R[4][2][2] = {{{1, 0}, {0, 0}}, {{0, 1}, {0, 0}}, {{0, 0}, {1, 0}}, {{0, 0}, {0, 1}}};
  static const bool G[4][2][2] = {{{0, 1}, {1, 0}}, {{1, 0}, {0, 1}}, {{1, 0}, {0, 1}}, {{0, 1}, {1, 0}}};
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2; ++j) {
      t.red[i][j] = R[c.bayerPattern][i][j];
      t.green[i][j] = G[c.bayerPattern][i][j];
    }
  // ColorspaceConversion.h:24-38
  static const float rgb2yuv[9] = {0.299f, 0.587f, 0.114f, -0.14713f, -0.28886f, 0.436f, 0.615f, -0.51499f, -0.10001f};
  static const float yuv2rgb[9] = {1.0f, 0.0f, 1.13983f, 1.0f, -0.39465f, -0.58060f, 1.0f, 2.03211f, 0.0f};
  float sat[9] = {1.0f, 0, 0, 0, c.saturation, 0, 0, 0, c.saturation};
  float tmp[9], satMat[9];
  mul33(yuv2rgb, sat, tmp);   // satMat = yuv2rgb * satMat * rgb2yuv (left to right)
  mul33(tmp, rgb2yuv, satMat);
  float ccmT[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) ccmT[j * 3 + i] = c.ccm[i * 3 + j];  // transpose(ccm, compositeCCM)
  mul33(ccmT, satMat, t.compositeCCM);                                // compositeCCM *= satMat
  for (int i = 0; i < 9; ++i) t.compositeCCM[i] = t.compositeCCM[i] * float(kIspToneLutSize - 1);
  // buildToneCurveLut
  t.toneLut.resize((size_t)kIspToneLutSize * 3);
  const float range = float((1 << c.outputBpp) - 1);
  const float dx = 1.0f / float(kIspToneLutSize - 1);
  const float angle = M_PI * 0.25f * c.contrast;  // double product narrowed to float
  const float slope = tanf(angle);
  const float bias = 0.5f * (1.0f - slope);
  for (int i = 0; i < kIspToneLutSize; ++i) {
    const float x = dx * i;
    float* o = &t.toneLut[(size_t)i * 3];
    if (c.disableToneCurve) {
      const float y = x * range;
      o[0] = o[1] = o[2] = y;
    } else {
      for (int k = 0; k < 3; ++k) {
        float v = powf(x, c.gamma[k]);
        v = lowKey(c.lowKeyBoost[k], v) + highKey(c.highKeyBoost[k], v);
        o[k] = clampf((slope * v + bias) * range, 0.0f, range);
      }
    }
  }
  return t;
}

// RawConverter::convert8Frame / convert12Frame (camera_isp/RawConverter.cpp:15-59): the sensor's packed bytes -> 16-bit
// samples. 8-bit: v * 0x101. 12-bit: three bytes hold two pixels (b0 << 4 | b1 & 15, then b2 << 4 | b1 >> 4), packed
// continuously over the whole frame; the 12-bit value is widened by replicating its top bits (v << 4 | v >> 8).
inline void ispUnpackFrame(int bits, const uint8_t* frame, int w, int h, uint16_t* out) {
  const size_t n = (size_t)w * h;
  if (bits == 8) {
    for (size_t i = 0; i < n; ++i) out[i] = (uint16_t)(frame[i] * 0x101);
    return;
  }
  size_t p = 0;
  for (size_t i = 0; i < n; ++i) {
    const uint16_t lo = frame[p], hi = frame[p + 1];
    uint16_t v;
    if ((i % (size_t)w) & 1) { p += 2; v = (uint16_t)(hi << 4 | lo >> 4); }
    else { p += 1; v = (uint16_t)(lo << 4 | (hi & 0xF)); }
    out[i] = (uint16_t)(v << 4 | v >> 8);
  }
}

// One frame through CameraIsp::loadImage + getImage(swizzle = true). raw: inH x inW uint16. out: (inH/resize) x
// (inW/resize) x 3 in B,G,R order, uint8 (outputBpp 8) or uint16 bit patterns (outputBpp 16).
inline void ispRun(const IspConfig& c, const uint16_t* raw, int inW, int inH, void* out) {
  using namespace isp_detail;
  if (c.demosaicFilter != 0 && c.demosaicFilter != 2) throw std::runtime_error("isp oracle: demosaic filter 1 (DCT) is not restated");
  if (c.resize != 1 && c.resize != 2 && c.resize != 4 && c.resize != 8) throw std::runtime_error("expecting a resize value of 1, 2, 4, or 8");
  const IspTables T = ispSetup(c);
  const int width = inW / c.resize, height = inH / c.resize;
  const int maxDimension = std::max(width, height);
  const int maxPixelValue = 65535;  // loadImage, 16-bit input (CameraIsp.h:845-849)
  const size_t n = (size_t)width * height;
  auto redPixel = [&](int i, int j) { return T.red[i % 2][j % 2]; };
  auto greenPixel = [&](int i, int j) { return T.green[i % 2][j % 2]; };
  std::vector<float> rawImage(n);
  auto RAW = [&](int i, int j) -> float& { return rawImage[(size_t)i * width + j]; };
  {  // resizeInput<uint16_t> (CameraIsp.h:338-358)
    const int resize = c.resize;
    const float areaRecip = 1.0f / (maxPixelValue * float(resize * resize));
    const int r = resize > 1 ? 2 : 1;
    for (int i = 0; i < height; ++i)
      for (int j = 0; j < width; ++j) {
        float sum = 0.0f;
        for (int k = 0; k < resize; ++k) {
          const int ip = i * resize + k * 2;
          const int ipp = reflecti(ip + (i % r), inH);
          for (int l = 0; l < resize; ++l) {
            const int jp = j * resize + l * 2;
            const int jpp = reflecti(jp + (j % r), inW);
            sum += float(raw[(size_t)ipp * inW + jpp]);
          }
        }
        RAW(i, j) = sum * areaRecip;
      }
  }
  {  // blackLevelAdjust (CameraIsp.h:1106-1126); addBlackLevelOffset (:884-888)
    const float bl[3] = {c.blackLevel[0] + float(c.blackLevelOffset), c.blackLevel[1] + float(c.blackLevelOffset),
                         c.blackLevel[2] + float(c.blackLevelOffset)};
    const float br = bl[0] / float(maxPixelValue), bg = bl[1] / float(maxPixelValue), bb = bl[2] / float(maxPixelValue);
    const float sr = 1.0f / (1.0f - br), sg = 1.0f / (1.0f - bg), sb = 1.0f / (1.0f - bb);
    for (int i = 0; i < height; ++i)
      for (int j = 0; j < width; ++j)
        if (RAW(i, j) < 1.0f) {
          if (redPixel(i, j)) RAW(i, j) = (RAW(i, j) - br) * sr;
          else if (greenPixel(i, j)) RAW(i, j) = (RAW(i, j) - bg) * sg;
          else RAW(i, j) = (RAW(i, j) - bb) * sb;
        }
  }
  {  // antiVignette (CameraIsp.h:1145-1154)
    std::vector<V3> curveH(width);
    for (int j = 0; j < width; ++j) curveH[j] = bezier(c.vignetteRollOffH, 0, c.nVignetteH - 1, float(j) / float(maxDimension));
    for (int i = 0; i < height; ++i) {
      const V3 vV = bezier(c.vignetteRollOffV, 0, c.nVignetteV - 1, float(i) / float(maxDimension));
      for (int j = 0; j < width; ++j) {
        const int ch = redPixel(i, j) ? 0 : greenPixel(i, j) ? 1 : 2;
        RAW(i, j) *= curveH[j].v[ch] * vV.v[ch];
      }
    }
  }
  // whiteBalance(clampOutput = true) (CameraIsp.h:1005-1021), clampAndStretch (:1128-1143)
  for (int i = 0; i < height; ++i)
    for (int j = 0; j < width; ++j) {
      const int ch = redPixel(i, j) ? 0 : greenPixel(i, j) ? 1 : 2;
      float v = RAW(i, j) * c.whiteBalanceGain[ch];
      v = clampf(v, 0.0f, 1.0f);
      const float lo = c.clampMin[ch], hi = c.clampMax[ch];
      v = clampf(v, lo, hi);
      RAW(i, j) = (v - lo) / (hi - lo);
    }
  if (c.stuckPixelRadius > 0) {  // removeStuckPixels (CameraIsp.h:1024-1104), in place and in its boustrophedon order
    struct Pval { float val; int i, j; };
    std::vector<Pval> region;
    const int R = c.stuckPixelRadius;
    for (int i = 0; i < height; ++i) {
      const bool even = (i % 2) == 0;
      const int jStart = even ? 0 : width - 1, jEnd = even ? width - 1 : 0, jStep = even ? 1 : -1;
      for (int j = jStart; j != jEnd; j += jStep) {  // (the last pixel of a scan line is never visited: `!=`, :1054)
        const bool tr = redPixel(i, j), tg = greenPixel(i, j), tb = !tr && !tg;
        region.clear();
        float mean = 0.0f;
        for (int y = -R; y <= R; y++) {
          const int ip = reflecti(i + y, height);
          for (int x = -R; x <= R; x++) {
            const int jp = reflecti(j + x, width);
            const bool pr = redPixel(ip, jp), pg = greenPixel(ip, jp), pb = !pr && !pg;
            if ((pr && tr) || (pg && tg) || (pb && tb)) {
              mean += RAW(ip, jp);
              region.push_back(Pval{RAW(ip, jp), ip, jp});
            }
          }
        }
        mean /= float(region.size());
        if (mean < c.stuckPixelDarknessThreshold) {
          // The reference sorts the region here and then walks it from the top while
          //     int k = region.size() - 1;  k <= region.size() - stuckPixelThreshold;  k--          (:1090-1092)
          // holds — a comparison of size_t values: with 2 <= threshold <= region.size() it is false at once and the pixel
          // is left alone; otherwise (threshold 0, 1, negative or above the region's size) it is true for every k down to
          // 0, the centre pixel is always found and takes the region's median value region[size / 2].val.
          const size_t n = region.size(), lim = n - (size_t)c.stuckPixelThreshold;
          bool found = false;
          for (int k = (int)n - 1; (size_t)k <= lim; k--)
            if (region[(size_t)k].i == i && region[(size_t)k].j == j) { found = true; break; }
          if (found) {
            std::vector<float> v(n);
            for (size_t k = 0; k < n; ++k) v[k] = region[k].val;
            std::sort(v.begin(), v.end());  // (only the median VALUE is read: the unstable order of equal values is immaterial)
            RAW(i, j) = v[n / 2];
          }
        }
      }
    }
  }
  // demosaic (CameraIsp.h:1156-1212): planes r, g, b hold the raw value at their own Bayer sites
  std::vector<float> r(n, 0.0f), g(n, 0.0f), b(n, 0.0f);
  auto AT = [&](std::vector<float>& m, int i, int j) -> float& { return m[(size_t)i * width + j]; };
  for (int i = 0; i < height; ++i)
    for (int j = 0; j < width; ++j) {
      if (redPixel(i, j)) AT(r, i, j) = RAW(i, j);
      else if (greenPixel(i, j)) AT(g, i, j) = RAW(i, j);
      else AT(b, i, j) = RAW(i, j);
    }
  if (c.demosaicFilter == 0) {  // demosaicBilinearFilter (CameraIsp.h:89-148)
    for (int i = 0; i < height; ++i) {
      const int i_1 = reflecti(i - 1, height), i1 = reflecti(i + 1, height);
      const bool redGreenRow = (redPixel(i, 0) && greenPixel(i, 1)) || (redPixel(i, 1) && greenPixel(i, 0));
      for (int j = 0; j < width; ++j) {
        const int j_1 = reflecti(j - 1, width), j1 = reflecti(j + 1, width);
        if (redPixel(i, j)) {
          AT(g, i, j) = bilerpf(AT(g, i_1, j), AT(g, i1, j), AT(g, i, j_1), AT(g, i, j1), 0.5f, 0.5f);
          AT(b, i, j) = bilerpf(AT(b, i_1, j_1), AT(b, i1, j_1), AT(b, i_1, j1), AT(b, i1, j1), 0.5f, 0.5f);
        } else if (greenPixel(i, j)) {
          if (redGreenRow) {
            AT(b, i, j) = (AT(b, i_1, j) + AT(b, i1, j)) / 2.0f;
            AT(r, i, j) = (AT(r, i, j_1) + AT(r, i, j1)) / 2.0f;
          } else {
            AT(r, i, j) = (AT(r, i_1, j) + AT(r, i1, j)) / 2.0f;
            AT(b, i, j) = (AT(b, i, j_1) + AT(b, i, j1)) / 2.0f;
          }
        } else {
          AT(g, i, j) = bilerpf(AT(g, i_1, j), AT(g, i1, j), AT(g, i, j_1), AT(g, i, j1), 0.5f, 0.5f);
          AT(r, i, j) = bilerpf(AT(r, i_1, j_1), AT(r, i1, j_1), AT(r, i_1, j1), AT(r, i1, j1), 0.5f, 0.5f);
        }
      }
    }
  } else {  // demosaicEdgeAware (CameraIsp.h:181-335)
    std::vector<float> gV(n), gH(n), dV(n), dH(n);
    for (int i = 0; i < height; ++i) {
      const int i_1 = reflecti(i - 1, height), i1 = reflecti(i + 1, height), i_2 = reflecti(i - 2, height), i2 = reflecti(i + 2, height);
      for (int j = 0; j < width; ++j) {
        const int j_1 = reflecti(j - 1, width), j1 = reflecti(j + 1, width), j_2 = reflecti(j - 2, width), j2 = reflecti(j + 2, width);
        if (greenPixel(i, j)) {
          AT(gV, i, j) = AT(g, i, j);
          AT(gH, i, j) = AT(g, i, j);
          AT(dV, i, j) = (fabsf(AT(g, i2, j) - AT(g, i, j)) + fabsf(AT(g, i, j) - AT(g, i_2, j))) / 2.0f;
          AT(dH, i, j) = (fabsf(AT(g, i, j2) - AT(g, i, j)) + fabsf(AT(g, i, j) - AT(g, i, j_2))) / 2.0f;
        } else {
          AT(gV, i, j) = (AT(g, i_1, j) + AT(g, i1, j)) / 2.0f;
          AT(gH, i, j) = (AT(g, i, j_1) + AT(g, i, j1)) / 2.0f;
          AT(dV, i, j) = (fabsf(AT(g, i_1, j) - AT(g, i1, j))) / 2.0f;
          AT(dH, i, j) = (fabsf(AT(g, i, j_1) - AT(g, i, j1))) / 2.0f;
          std::vector<float>& ch = redPixel(i, j) ? r : b;
          AT(gV, i, j) += (2.0f * AT(ch, i, j) - AT(ch, i_2, j) - AT(ch, i2, j)) / 4.0f;
          AT(gH, i, j) += (2.0f * AT(ch, i, j) - AT(ch, i, j_2) - AT(ch, i, j2)) / 4.0f;
          AT(dV, i, j) += fabsf(-2.0f * AT(ch, i, j) + AT(ch, i_2, j) + AT(ch, i2, j)) / 2.0f;
          AT(dH, i, j) += fabsf(-2.0f * AT(ch, i, j) + AT(ch, i, j_2) + AT(ch, i, j2)) / 2.0f;
        }
      }
    }
    const int w = 4, diameter = 2 * w + 1, diameterSquared = diameter * diameter;
    for (int i = 0; i < height; ++i)
      for (int j = 0; j < width; ++j) {
        int hCount = 0;  // homogeneity test
        for (int l = -w; l <= w; ++l) {
          const int il = reflecti(i + l, height);
          for (int k = -w; k <= w; ++k) {
            const int jk = reflecti(j + k, width);
            hCount += (AT(dH, il, jk) <= AT(dV, il, jk));
          }
        }
        AT(g, i, j) = hCount < diameterSquared / 2 ? AT(gV, i, j) : AT(gH, i, j);
      }
    std::vector<float> rmg(n, 0.0f), bmg(n, 0.0f);  // red - green, blue - green at their own sites
    for (int i = 0; i < height; ++i)
      for (int j = 0; j < width; ++j) {
        if (redPixel(i, j)) AT(rmg, i, j) = AT(r, i, j) - AT(g, i, j);
        else if (!greenPixel(i, j)) AT(bmg, i, j) = AT(b, i, j) - AT(g, i, j);
      }
    const std::vector<float>& pG = g;  // pGreen == green: green is final before this loop
    for (int i = 0; i < height; ++i) {
      const int i_1 = reflecti(i - 1, height), i1 = reflecti(i + 1, height), i_2 = reflecti(i - 2, height), i2 = reflecti(i + 2, height);
      const bool redGreenRow = (redPixel(i, 0) && greenPixel(i, 1)) || (redPixel(i, 1) && greenPixel(i, 0));
      for (int j = 0; j < width; ++j) {
        const int j_1 = reflecti(j - 1, width), j1 = reflecti(j + 1, width), j_2 = reflecti(j - 2, width), j2 = reflecti(j + 2, width);
        const float pg = pG[(size_t)i * width + j];
        if (redPixel(i, j)) {
          AT(b, i, j) = (AT(bmg, i_1, j_1) + AT(bmg, i1, j_1) + AT(bmg, i_1, j1) + AT(bmg, i1, j1)) / 4.0f + pg;
          AT(r, i, j) = (AT(rmg, i, j) + AT(rmg, i_2, j) + AT(rmg, i2, j) + AT(rmg, i, j_2) + AT(rmg, i, j2)) / 5.0f + pg;
        } else if (greenPixel(i, j)) {
          std::vector<float>& d1 = redGreenRow ? bmg : rmg;
          std::vector<float>& d2 = redGreenRow ? rmg : bmg;
          std::vector<float>& c1 = redGreenRow ? b : r;
          std::vector<float>& c2 = redGreenRow ? r : b;
          // (the reference adds (i1, j2) twice and never (i1, j): CameraIsp.h:298-304)
          AT(c1, i, j) = (AT(d1, i_1, j_2) + AT(d1, i_1, j) + AT(d1, i_1, j2) + AT(d1, i1, j_2) + AT(d1, i1, j2) + AT(d1, i1, j2)) / 6.0f + pg;
          AT(c2, i, j) = (AT(d2, i_2, j_1) + AT(d2, i, j_1) + AT(d2, i2, j_1) + AT(d2, i_2, j1) + AT(d2, i, j1) + AT(d2, i2, j1)) / 6.0f + pg;
        } else {
          AT(r, i, j) = (AT(rmg, i_1, j_1) + AT(rmg, i1, j_1) + AT(rmg, i_1, j1) + AT(rmg, i1, j1)) / 4.0f + pg;
          AT(b, i, j) = (AT(bmg, i, j) + AT(bmg, i_2, j) + AT(bmg, i2, j) + AT(bmg, i, j_2) + AT(bmg, i, j2)) / 5.0f + pg;
        }
      }
    }
  }
  // colorCorrect (CameraIsp.h:1214-1242): composite CCM (scaled by 4095), clamp, float -> int index, tone LUT
  std::vector<float> img(n * 3);
  const float lutRange = float(kIspToneLutSize - 1);
  for (size_t p = 0; p < n; ++p) {
    const float pr = r[p], pgv = g[p], pb = b[p];
    for (int k = 0; k < 3; ++k) {
      const float v = T.compositeCCM[k * 3] * pr + T.compositeCCM[k * 3 + 1] * pgv + T.compositeCCM[k * 3 + 2] * pb;
      const int idx = (int)clampf(v, 0.0f, lutRange);  // vector index: float -> size_t by truncation
      img[p * 3 + k] = T.toneLut[(size_t)idx * 3 + k];
    }
  }
  // sharpen (CameraIsp.h:1244-1259; Filter.h:38-126): two-tap IIR low pass with reflected boundaries, unsharp mask
  if (c.sharpening[0] != 0.0 && c.sharpening[1] != 0.0 && c.sharpening[2] != 0.0) {
    const float maxVal = (1 << c.outputBpp) - 1.0f;
    const float alpha = powf(c.sharpeningSupport, 1.0f / 4.0f);
    std::vector<float> lp(n * 3), buffer((size_t)std::max(width, height) * 3);
    for (int i = 0; i < height; ++i) {  // horizontal: causal into the buffer, anticausal into lp
      float v[3] = {img[((size_t)i * width) * 3], img[((size_t)i * width) * 3 + 1], img[((size_t)i * width) * 3 + 2]};
      for (int j = 1; j <= width; ++j) {
        const float* ip = &img[((size_t)i * width + reflecti(j, width)) * 3];
        float* bo = &buffer[(size_t)reflecti(j - 1, width) * 3];
        for (int k = 0; k < 3; ++k) { v[k] = ip[k] * (1.0f - alpha) + v[k] * alpha; bo[k] = v[k]; }
      }
      for (int j = width - 2; j >= -1; --j) {
        const float* ip = &buffer[(size_t)reflecti(j, width) * 3];
        float* o = &lp[((size_t)i * width + j + 1) * 3];
        for (int k = 0; k < 3; ++k) { v[k] = ip[k] * (1.0f - alpha) + v[k] * alpha; o[k] = clampf(v[k], 0.0f, maxVal); }
      }
    }
    for (int j = 0; j < width; ++j) {  // vertical, in place on lp
      float v[3] = {lp[(size_t)j * 3], lp[(size_t)j * 3 + 1], lp[(size_t)j * 3 + 2]};
      for (int i = 1; i <= height; ++i) {
        const float* ip = &lp[((size_t)reflecti(i, height) * width + j) * 3];
        float* bo = &buffer[(size_t)reflecti(i - 1, height) * 3];
        for (int k = 0; k < 3; ++k) { v[k] = ip[k] * (1.0f - alpha) + v[k] * alpha; bo[k] = v[k]; }
      }
      for (int i = height - 2; i >= -1; --i) {
        const float* ip = &buffer[(size_t)reflecti(i, height) * 3];
        float* o = &lp[((size_t)(i + 1) * width + j) * 3];
        for (int k = 0; k < 3; ++k) { v[k] = ip[k] * (1.0f - alpha) + v[k] * alpha; o[k] = clampf(v[k], 0.0f, maxVal); }
      }
    }
    const float amount[3] = {1.0f + c.sharpening[0], 1.0f + c.sharpening[1], 1.0f + c.sharpening[2]};
    for (size_t p = 0; p < n * 3; ++p) {  // sharpenWithIirLowPass (Filter.h:92-126)
      const int k = (int)(p % 3);
      const float hp = img[p] - lp[p];
      const float ng = 1.0f - expf(-((hp * hp) * c.noiseCore));
      img[p] = clampf(lp[p] + hp * ng * amount[k], 0.0f, maxVal);
    }
  }
  // getImage (CameraIsp.h:1275-1299): float -> uchar / short by C++ conversion (truncation), swizzled to B,G,R
  for (size_t p = 0; p < n; ++p)
    for (int k = 0; k < 3; ++k) {
      const float v = img[p * 3 + k];
      if (c.outputBpp == 8) reinterpret_cast<uint8_t*>(out)[p * 3 + (2 - k)] = (uint8_t)(int)v;
      else reinterpret_cast<uint16_t*>(out)[p * 3 + (2 - k)] = (uint16_t)(int)v;
    }
}

}  // namespace orc
