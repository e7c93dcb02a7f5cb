// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
//
// cvlite.h: an OpenCV-free CPU restatement of the handful of OpenCV 3.1-era
// imgproc primitives that Surround360's stereo-panorama hot path calls
// (resize, remap, GaussianBlur, Sobel, medianBlur, cvtColor, erode, ...).
//
// PARITY UNPINNED (these primitives only; everything the reference's authors wrote above them is pinned — the reference's
// whole programs run over them, see render.h / pixflow.h / isp.h and tests/test_cpu_refprogram.py): OpenCV
// (pinned by the reference at git f109c01, WITH_IPP=OFF,
// surround360_render/README.md:144-153) is not available in this environment and
// the reference ships no golden vectors for this path. Each
// primitive below restates the published OpenCV 3.1 algorithm as the reference's
// x86-64 build executes it: SSE2 is part of the x86-64 baseline, so where OpenCV's
// SSE2 code path rounds differently from its scalar tail loop (8-bit cubic resize,
// 8-bit Gaussian column pass) BOTH are restated, each on the elements it covers.
// DESIGN.md §2 lists, per primitive, what is restated and how sure that is; where
// OpenCV's behaviour is ambiguous, THIS FILE IS THE DEFINITION the HIP kernels are
// tested against.
//
// Build with -ffp-contract=off: the reference's x86 build has no FMA
// (surround360_render/CMakeLists.txt:33-35), and float op order matters.
#pragma once
#include <algorithm>
#include <cassert>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace orc {

// ---------------------------------------------------------------------------
// Row-major, channel-interleaved image (== continuous cv::Mat).
template <typename T>
struct Img {
  int w = 0, h = 0, c = 0;
  std::vector<T> d;
  Img() {}
  Img(int w_, int h_, int c_) : w(w_), h(h_), c(c_), d(size_t(w_) * h_ * c_) {}
  Img(int w_, int h_, int c_, T v) : w(w_), h(h_), c(c_), d(size_t(w_) * h_ * c_, v) {}
  bool empty() const { return d.empty(); }
  T* row(int y) { return d.data() + size_t(y) * w * c; }
  const T* row(int y) const { return d.data() + size_t(y) * w * c; }
  T* px(int y, int x) { return row(y) + size_t(x) * c; }
  const T* px(int y, int x) const { return row(y) + size_t(x) * c; }
  T& at(int y, int x, int k = 0) { return d[(size_t(y) * w + x) * c + k]; }
  const T& at(int y, int x, int k = 0) const { return d[(size_t(y) * w + x) * c + k]; }
  size_t bytes() const { return d.size() * sizeof(T); }
};
using ImgU8 = Img<uint8_t>;
using ImgF = Img<float>;

// ---------------------------------------------------------------------------
// Rounding helpers with OpenCV/SSE semantics.
// cvRound(float/double) = cvtss2si / cvtsd2si: round-half-even, "integer
// indefinite" (INT_MIN) on overflow or NaN.
static inline int cvRoundF(float v) {
  if (!(v >= -2147483648.0f && v < 2147483648.0f)) return INT_MIN;
  return (int)lrintf(v);
}
static inline int cvRoundD(double v) {
  if (!(v >= -2147483648.5 && v < 2147483647.5)) return INT_MIN;
  return (int)lrint(v);
}
static inline int cvFloorF(float v) {
  int i = (int)v;
  return i - (v < (float)i);
}
static inline uint8_t satU8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
static inline short satS16(int v) { return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }
// C++ implicit float -> unsigned char as the reference relies on it
// (Vec4b(float,...), NovelView.cpp:144-148): truncation toward zero.
// Values are always within [0,256) on this path; clamp defensively so that the
// GPU and CPU agree on out-of-range behaviour as well.
static inline uint8_t truncU8(float v) {
  int i = (int)v;
  return (uint8_t)(i < 0 ? 0 : i > 255 ? 255 : i);
}
static inline int clipIdx(int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; }

// ---------------------------------------------------------------------------
// Bicubic coefficients, A = -0.75 (OpenCV interpolateCubic).
static inline void interpolateCubic(float x, float* coeffs) {
  const float A = -0.75f;
  coeffs[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
  coeffs[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
  coeffs[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
  coeffs[3] = 1.f - coeffs[0] - coeffs[1] - coeffs[2];
}

// Source coordinate of destination index d:
// f = (float)((d+0.5)*scale - 0.5); s = floor(f); f -= s, scale = 1/(dst/src).
static inline void resizeCoord(int d, double scale, int* s, float* f) {
  float fx = (float)((d + 0.5) * scale - 0.5);
  int sx = cvFloorF(fx);
  *s = sx;
  *f = fx - sx;
}
static inline double resizeScale(int srcN, int dstN) {
  double inv = (double)dstN / (double)srcN;
  return 1.0 / inv;
}

// ---------------------------------------------------------------------------
// resize INTER_CUBIC, 8-bit, any channel count (fixed point):
// short weights = saturate(round(w*2048)); H pass -> int32 (HResizeCubic, no SIMD);
// V pass: float arithmetic on the SSE2-covered elements, (sum + (1<<21)) >> 22 on the
// scalar tail (see below), saturate. Taps clamped (replicate).
// Used by PixFlow entry downscale (PixFlow.h:98-107) and the final equirect
// resize (TestRenderStereoPanorama.cpp:938-957).
static inline ImgU8 resizeCubicU8(const ImgU8& src, int dw, int dh) {
  const int cn = src.c, sw = src.w, sh = src.h;
  const double scx = resizeScale(sw, dw), scy = resizeScale(sh, dh);
  std::vector<int> xofs(dw), yofs(dh);
  std::vector<short> xa(size_t(dw) * 4), ya(size_t(dh) * 4);
  for (int dx = 0; dx < dw; ++dx) {
    float f, cb[4];
    resizeCoord(dx, scx, &xofs[dx], &f);
    interpolateCubic(f, cb);
    for (int k = 0; k < 4; ++k) xa[dx * 4 + k] = satS16(cvRoundF(cb[k] * 2048.f));
  }
  for (int dy = 0; dy < dh; ++dy) {
    float f, cb[4];
    resizeCoord(dy, scy, &yofs[dy], &f);
    interpolateCubic(f, cb);
    for (int k = 0; k < 4; ++k) ya[dy * 4 + k] = satS16(cvRoundF(cb[k] * 2048.f));
  }
  // horizontal pass for every source row
  std::vector<int> hbuf(size_t(sh) * dw * cn);
  for (int y = 0; y < sh; ++y) {
    const uint8_t* S = src.row(y);
    int* D = hbuf.data() + size_t(y) * dw * cn;
    for (int dx = 0; dx < dw; ++dx) {
      const int sx = xofs[dx];
      const short* a = &xa[dx * 4];
      const int x0 = clipIdx(sx - 1, 0, sw), x1 = clipIdx(sx, 0, sw), x2 = clipIdx(sx + 1, 0, sw),
                x3 = clipIdx(sx + 2, 0, sw);
      for (int k = 0; k < cn; ++k)
        D[dx * cn + k] = S[x0 * cn + k] * a[0] + S[x1 * cn + k] * a[1] + S[x2 * cn + k] * a[2] +
                         S[x3 * cn + k] * a[3];
    }
  }
  ImgU8 dst(dw, dh, cn);
  for (int dy = 0; dy < dh; ++dy) {
    const int sy = yofs[dy];
    const short* b = &ya[dy * 4];
    const int* S0 = hbuf.data() + size_t(clipIdx(sy - 1, 0, sh)) * dw * cn;
    const int* S1 = hbuf.data() + size_t(clipIdx(sy, 0, sh)) * dw * cn;
    const int* S2 = hbuf.data() + size_t(clipIdx(sy + 1, 0, sh)) * dw * cn;
    const int* S3 = hbuf.data() + size_t(clipIdx(sy + 2, 0, sh)) * dw * cn;
    uint8_t* D = dst.row(dy);
    // VResizeCubicVec_32s8u (the SSE2 path, taken for whole groups of 8 elements of the row): the taps become
    // floats b*2^-22, the int32 row sums are converted to float, multiplied and added left to right in float,
    // and the sum is converted back with cvtps2dq (round-half-even) and saturated.
    const int wc = dw * cn, vecEnd = (wc / 8) * 8;
    const float scale = 1.f / (2048 * 2048);
    const float fb0 = b[0] * scale, fb1 = b[1] * scale, fb2 = b[2] * scale, fb3 = b[3] * scale;
    for (int x = 0; x < vecEnd; ++x) {
      float s = (float)S0[x] * fb0;
      s = s + (float)S1[x] * fb1;
      s = s + (float)S2[x] * fb2;
      s = s + (float)S3[x] * fb3;
      D[x] = satU8(cvRoundF(s));
    }
    // scalar tail: FixedPtCast<int, uchar, 22>
    for (int x = vecEnd; x < wc; ++x) {
      int v = S0[x] * b[0] + S1[x] * b[1] + S2[x] * b[2] + S3[x] * b[3];
      D[x] = satU8((v + (1 << 21)) >> 22);
    }
  }
  return dst;
}

// resize INTER_CUBIC, float, any channels (flow upscale between pyramid levels,
// PixFlow.h:170; prevFlow downscale :103). Float weights, float accumulate,
// left-associated 4-term sums, H pass then V pass.
static inline ImgF resizeCubicF32(const ImgF& src, int dw, int dh) {
  const int cn = src.c, sw = src.w, sh = src.h;
  const double scx = resizeScale(sw, dw), scy = resizeScale(sh, dh);
  std::vector<int> xofs(dw), yofs(dh);
  std::vector<float> xa(size_t(dw) * 4), ya(size_t(dh) * 4);
  for (int dx = 0; dx < dw; ++dx) {
    float f;
    resizeCoord(dx, scx, &xofs[dx], &f);
    interpolateCubic(f, &xa[dx * 4]);
  }
  for (int dy = 0; dy < dh; ++dy) {
    float f;
    resizeCoord(dy, scy, &yofs[dy], &f);
    interpolateCubic(f, &ya[dy * 4]);
  }
  std::vector<float> hbuf(size_t(sh) * dw * cn);
  for (int y = 0; y < sh; ++y) {
    const float* S = src.row(y);
    float* D = hbuf.data() + size_t(y) * dw * cn;
    for (int dx = 0; dx < dw; ++dx) {
      const int sx = xofs[dx];
      const float* a = &xa[dx * 4];
      const int x0 = clipIdx(sx - 1, 0, sw), x1 = clipIdx(sx, 0, sw), x2 = clipIdx(sx + 1, 0, sw),
                x3 = clipIdx(sx + 2, 0, sw);
      for (int k = 0; k < cn; ++k)
        D[dx * cn + k] = S[x0 * cn + k] * a[0] + S[x1 * cn + k] * a[1] + S[x2 * cn + k] * a[2] +
                         S[x3 * cn + k] * a[3];
    }
  }
  ImgF dst(dw, dh, cn);
  for (int dy = 0; dy < dh; ++dy) {
    const int sy = yofs[dy];
    const float* b = &ya[dy * 4];
    const float* S0 = hbuf.data() + size_t(clipIdx(sy - 1, 0, sh)) * dw * cn;
    const float* S1 = hbuf.data() + size_t(clipIdx(sy, 0, sh)) * dw * cn;
    const float* S2 = hbuf.data() + size_t(clipIdx(sy + 1, 0, sh)) * dw * cn;
    const float* S3 = hbuf.data() + size_t(clipIdx(sy + 2, 0, sh)) * dw * cn;
    float* D = dst.row(dy);
    for (int x = 0; x < dw * cn; ++x) D[x] = S0[x] * b[0] + S1[x] * b[1] + S2[x] * b[2] + S3[x] * b[3];
  }
  return dst;
}

// resize INTER_LINEAR, float, any channels (pyramid ×0.9, PixFlow.h:487; final
// flow upscale :176). SURVEY App. A.1: s<0 => s=0,f=0; s>=src-1 => s=src-1,f=0
// horizontally (tail uses S[s]*1); vertically rows are clipped but f is kept.
static inline ImgF resizeLinearF32(const ImgF& src, int dw, int dh) {
  const int cn = src.c, sw = src.w, sh = src.h;
  const double scx = resizeScale(sw, dw), scy = resizeScale(sh, dh);
  std::vector<int> xofs(dw);
  std::vector<float> xf(dw);
  for (int dx = 0; dx < dw; ++dx) {
    int sx;
    float f;
    resizeCoord(dx, scx, &sx, &f);
    if (sx < 0) { f = 0; sx = 0; }
    if (sx >= sw - 1) { f = 0; sx = sw - 1; }
    xofs[dx] = sx;
    xf[dx] = f;
  }
  std::vector<float> hbuf(size_t(sh) * dw * cn);
  for (int y = 0; y < sh; ++y) {
    const float* S = src.row(y);
    float* D = hbuf.data() + size_t(y) * dw * cn;
    for (int dx = 0; dx < dw; ++dx) {
      const int sx = xofs[dx];
      if (sx >= sw - 1) {
        for (int k = 0; k < cn; ++k) D[dx * cn + k] = S[sx * cn + k] * 1.0f;
      } else {
        const float a0 = 1.f - xf[dx], a1 = xf[dx];
        for (int k = 0; k < cn; ++k) D[dx * cn + k] = S[sx * cn + k] * a0 + S[(sx + 1) * cn + k] * a1;
      }
    }
  }
  ImgF dst(dw, dh, cn);
  for (int dy = 0; dy < dh; ++dy) {
    int sy;
    float f;
    resizeCoord(dy, scy, &sy, &f);
    const float b0 = 1.f - f, b1 = f;
    const float* S0 = hbuf.data() + size_t(clipIdx(sy, 0, sh)) * dw * cn;
    const float* S1 = hbuf.data() + size_t(clipIdx(sy + 1, 0, sh)) * dw * cn;
    float* D = dst.row(dy);
    for (int x = 0; x < dw * cn; ++x) D[x] = S0[x] * b0 + S1[x] * b1;
  }
  return dst;
}

// ---------------------------------------------------------------------------
// remap. Coordinates quantised to 1/32 px; 32x32 table of
// 4x4 weights.
struct BicubicTab {
  float f[1024][16];
  short i[1024][16];
  BicubicTab() {
    float t1[32][4];
    for (int k = 0; k < 32; ++k) interpolateCubic(k * (1.f / 32), t1[k]);
    for (int iy = 0; iy < 32; ++iy)
      for (int ix = 0; ix < 32; ++ix) {
        float* tf = f[iy * 32 + ix];
        short* ti = i[iy * 32 + ix];
        int isum = 0;
        for (int k1 = 0; k1 < 4; ++k1) {
          const float vy = t1[iy][k1];
          for (int k2 = 0; k2 < 4; ++k2) {
            const float v = vy * t1[ix][k2];
            tf[k1 * 4 + k2] = v;
            isum += ti[k1 * 4 + k2] = satS16(cvRoundF(v * 32768.f));
          }
        }
        if (isum != 32768) {  // force the integer weights to sum to exactly 1<<15
          const int diff = isum - 32768;
          const int ks2 = 2;
          int Mk1 = ks2, Mk2 = ks2, mk1 = ks2, mk2 = ks2;
          for (int k1 = ks2; k1 < ks2 + 2; ++k1)
            for (int k2 = ks2; k2 < ks2 + 2; ++k2) {
              if (ti[k1 * 4 + k2] < ti[mk1 * 4 + mk2]) mk1 = k1, mk2 = k2;
              else if (ti[k1 * 4 + k2] > ti[Mk1 * 4 + Mk2]) Mk1 = k1, Mk2 = k2;
            }
          if (diff < 0) ti[Mk1 * 4 + Mk2] = (short)(ti[Mk1 * 4 + Mk2] - diff);
          else ti[mk1 * 4 + mk2] = (short)(ti[mk1 * 4 + mk2] - diff);
        }
      }
  }
};
static inline const BicubicTab& bicubicTab() {
  static const BicubicTab t;
  return t;
}

// Fixed-point coordinate of a float map entry: returns integer pixel (already
// offset by -1 to the first tap) and the 10-bit fraction index.
static inline void remapCoord(float mx, float my, int* sx, int* sy, int* fxy) {
  const int ix = cvRoundF(mx * 32.f), iy = cvRoundF(my * 32.f);
  *fxy = (iy & 31) * 32 + (ix & 31);
  *sx = (int)satS16(ix >> 5) - 1;
  *sy = (int)satS16(iy >> 5) - 1;
}

// remap INTER_CUBIC, BORDER_CONSTANT(0), 8-bit source with `cn` channels.
// map is 2-channel float (x,y). ImageWarper.cpp:173, NovelView.cpp:206,
// TestRenderStereoPanorama.cpp:497-503.
static inline ImgU8 remapCubicU8(const ImgU8& src, const ImgF& map) {
  assert(map.c == 2);
  const BicubicTab& T = bicubicTab();
  const int cn = src.c, sw = src.w, sh = src.h;
  ImgU8 dst(map.w, map.h, cn);
  const unsigned width1 = std::max(sw - 3, 0), height1 = std::max(sh - 3, 0);
  for (int y = 0; y < map.h; ++y) {
    const float* M = map.row(y);
    uint8_t* D = dst.row(y);
    for (int x = 0; x < map.w; ++x, D += cn) {
      int sx, sy, fxy;
      remapCoord(M[2 * x], M[2 * x + 1], &sx, &sy, &fxy);
      const short* w = T.i[fxy];
      if ((unsigned)sx < width1 && (unsigned)sy < height1) {
        const uint8_t* S = src.px(sy, sx);
        const size_t ss = size_t(sw) * cn;
        for (int k = 0; k < cn; ++k) {
          int sum = 0;
          for (int r = 0; r < 4; ++r) {
            const uint8_t* Sr = S + r * ss + k;
            sum += Sr[0] * w[r * 4] + Sr[cn] * w[r * 4 + 1] + Sr[cn * 2] * w[r * 4 + 2] +
                   Sr[cn * 3] * w[r * 4 + 3];
          }
          D[k] = satU8((sum + (1 << 14)) >> 15);
        }
      } else if (sx >= sw || sx + 4 <= 0 || sy >= sh || sy + 4 <= 0) {
        for (int k = 0; k < cn; ++k) D[k] = 0;
      } else {
        for (int k = 0; k < cn; ++k) {
          int sum = 0;
          for (int r = 0; r < 4; ++r) {
            const int yi = sy + r;
            if (yi < 0 || yi >= sh) continue;
            for (int q = 0; q < 4; ++q) {
              const int xi = sx + q;
              if (xi < 0 || xi >= sw) continue;
              sum += src.at(yi, xi, k) * w[r * 4 + q];
            }
          }
          D[k] = satU8((sum + (1 << 14)) >> 15);
        }
      }
    }
  }
  return dst;
}

// remap INTER_CUBIC, BORDER_WRAP, 8-bit source (ImageWarper.cpp:131-138: equirect -> cubemap face). Taps whose
// 4x4 window leaves the image are fetched at borderInterpolate(p, len, BORDER_WRAP) in x and in y.
static inline int borderWrap(int p, int len) {
  if (p < 0) p -= ((p - len + 1) / len) * len;
  if (p >= len) p %= len;
  return p;
}
static inline ImgU8 remapCubicU8Wrap(const ImgU8& src, const ImgF& map) {
  assert(map.c == 2);
  const BicubicTab& T = bicubicTab();
  const int cn = src.c, sw = src.w, sh = src.h;
  ImgU8 dst(map.w, map.h, cn);
  for (int y = 0; y < map.h; ++y) {
    const float* M = map.row(y);
    uint8_t* D = dst.row(y);
    for (int x = 0; x < map.w; ++x, D += cn) {
      int sx, sy, fxy;
      remapCoord(M[2 * x], M[2 * x + 1], &sx, &sy, &fxy);
      const short* w = T.i[fxy];
      int xs[4], ys[4];
      for (int q = 0; q < 4; ++q) { xs[q] = borderWrap(sx + q, sw); ys[q] = borderWrap(sy + q, sh); }
      for (int k = 0; k < cn; ++k) {
        int sum = 0;
        for (int r = 0; r < 4; ++r)
          for (int q = 0; q < 4; ++q) sum += src.at(ys[r], xs[q], k) * w[r * 4 + q];
        D[k] = satU8((sum + (1 << 14)) >> 15);
      }
    }
  }
  return dst;
}

// remap INTER_CUBIC, BORDER_CONSTANT(0), float source (flow field remap in
// renderLazyNovelView, NovelView.cpp:191). Float weights and accumulate;
// interior: sum = row0(4-term) ; sum += row1 ; ... ; border: sequential adds.
static inline ImgF remapCubicF32(const ImgF& src, const ImgF& map) {
  assert(map.c == 2);
  const BicubicTab& T = bicubicTab();
  const int cn = src.c, sw = src.w, sh = src.h;
  ImgF dst(map.w, map.h, cn);
  const unsigned width1 = std::max(sw - 3, 0), height1 = std::max(sh - 3, 0);
  for (int y = 0; y < map.h; ++y) {
    const float* M = map.row(y);
    float* D = dst.row(y);
    for (int x = 0; x < map.w; ++x, D += cn) {
      int sx, sy, fxy;
      remapCoord(M[2 * x], M[2 * x + 1], &sx, &sy, &fxy);
      const float* w = T.f[fxy];
      if ((unsigned)sx < width1 && (unsigned)sy < height1) {
        const float* S = src.px(sy, sx);
        const size_t ss = size_t(sw) * cn;
        for (int k = 0; k < cn; ++k) {
          const float* S0 = S + k;
          float sum = S0[0] * w[0] + S0[cn] * w[1] + S0[cn * 2] * w[2] + S0[cn * 3] * w[3];
          S0 += ss;
          sum += S0[0] * w[4] + S0[cn] * w[5] + S0[cn * 2] * w[6] + S0[cn * 3] * w[7];
          S0 += ss;
          sum += S0[0] * w[8] + S0[cn] * w[9] + S0[cn * 2] * w[10] + S0[cn * 3] * w[11];
          S0 += ss;
          sum += S0[0] * w[12] + S0[cn] * w[13] + S0[cn * 2] * w[14] + S0[cn * 3] * w[15];
          D[k] = sum;
        }
      } else if (sx >= sw || sx + 4 <= 0 || sy >= sh || sy + 4 <= 0) {
        for (int k = 0; k < cn; ++k) D[k] = 0.f;
      } else {
        for (int k = 0; k < cn; ++k) {
          float sum = 0.f;
          for (int r = 0; r < 4; ++r) {
            const int yi = sy + r;
            if (yi < 0 || yi >= sh) continue;
            for (int q = 0; q < 4; ++q) {
              const int xi = sx + q;
              if (xi < 0 || xi >= sw) continue;
              sum += src.at(yi, xi, k) * w[r * 4 + q];
            }
          }
          D[k] = sum;
        }
      }
    }
  }
  return dst;
}

// ---------------------------------------------------------------------------
// GaussianBlur on CV_32F. Kernel: exp in double, stored float, normalised by the (double) sum
// of the float taps (getGaussianKernel). BORDER_REFLECT_101, row pass then column pass (sepFilter2D).
//  * row pass, ksize <= 5: SymmRowSmallFilter — k[c]*x[c] + k[c+1]*(x[c+1] + x[c-1]) (+ k[c+2]*(x[c+2] + x[c-2]));
//  * row pass, ksize  > 5: the generic RowFilter<float,float,RowVec_32f> — taps accumulated LEFT TO RIGHT,
//    s = k[0]*x[0]; s += k[1]*x[1]; ...; the SSE2 loop (whole groups of 8 elements of the row) starts from +0
//    instead (only the sign of an all-minus-zero sum can differ);
//  * column pass: SymmColumnFilter / SymmColumnSmallFilter — k[c]*x[c] + delta (delta = +0), then
//    += k[c+j]*(x[c+j] + x[c-j]).
static inline std::vector<float> gaussianKernel(int n, double sigma) {
  std::vector<float> k(n);
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
  return k;
}
static inline int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) {
    if (p < 0) p = -p;
    else p = 2 * len - 2 - p;
  }
  return p;
}
static inline ImgF gaussianBlurF32(const ImgF& src, int ksize, double sigma) {
  const std::vector<float> kern = gaussianKernel(ksize, sigma);
  const int r = ksize / 2, cn = src.c, w = src.w, h = src.h;
  const float* kc = kern.data() + r;
  ImgF tmp(w, h, cn), dst(w, h, cn);
  std::vector<int> xi(w + 2 * r);
  for (int i = 0; i < w + 2 * r; ++i) xi[i] = reflect101(i - r, w);
  const int vecEnd = ((w * cn) / 8) * 8;
  for (int y = 0; y < h; ++y) {
    const float* S = src.row(y);
    float* D = tmp.row(y);
    for (int x = 0; x < w; ++x)
      for (int k = 0; k < cn; ++k) {
        float s;
        if (ksize <= 5) {
          s = kc[0] * S[xi[x + r] * cn + k];
          for (int j = 1; j <= r; ++j) s += kc[j] * (S[xi[x + r + j] * cn + k] + S[xi[x + r - j] * cn + k]);
        } else {
          s = (x * cn + k < vecEnd) ? 0.0f : -0.0f;  // (-0) + p == p for every p
          for (int j = 0; j < ksize; ++j) s += kern[j] * S[xi[x + j] * cn + k];
        }
        D[x * cn + k] = s;
      }
  }
  for (int y = 0; y < h; ++y) {
    float* D = dst.row(y);
    const float* Sc = tmp.row(y);
    for (int x = 0; x < w * cn; ++x) D[x] = kc[0] * Sc[x] + 0.0f;
    for (int j = 1; j <= r; ++j) {
      const float* Sp = tmp.row(reflect101(y + j, h));
      const float* Sm = tmp.row(reflect101(y - j, h));
      for (int x = 0; x < w * cn; ++x) D[x] += kc[j] * (Sp[x] + Sm[x]);
    }
  }
  return dst;
}

// 8-bit GaussianBlur on a single-channel image (alpha feather, CvUtil.cpp:152). createSeparableLinearFilter's
// fixed-point branch: taps = round(k*256) per pass, int32 row pass (RowVec_8u32s, exact). Column pass:
//  * SymmColumnVec_32s8u (the SSE2 path, whole groups of 4 columns): taps as floats ik*2^-16, the int32 row sums
//    (pairs added as integers first) converted to float, accumulated centre first then outwards in float, converted
//    back with cvtps2dq (round-half-even) and saturated;
//  * scalar tail (the last w % 4 columns): FixedPtCastEx — (sum + (1<<15)) >> 16.
// The two differ only on exact .5 ties (and when a float partial sum needs more than 24 bits).
static inline ImgU8 gaussianBlurU8C1(const ImgU8& src, int ksize, double sigma) {
  assert(src.c == 1);
  const std::vector<float> kern = gaussianKernel(ksize, sigma);
  std::vector<int> ik(ksize);
  for (int i = 0; i < ksize; ++i) ik[i] = cvRoundF(kern[i] * 256.f);
  const int r = ksize / 2, w = src.w, h = src.h;
  std::vector<float> fk(r + 1);
  for (int j = 0; j <= r; ++j) fk[j] = (float)ik[r + j] * (float)(1. / 65536);
  std::vector<int> tmp(size_t(w) * h);
  for (int y = 0; y < h; ++y) {
    const uint8_t* S = src.row(y);
    for (int x = 0; x < w; ++x) {
      int s = ik[r] * S[x];
      for (int j = 1; j <= r; ++j) s += ik[r + j] * (S[reflect101(x + j, w)] + S[reflect101(x - j, w)]);
      tmp[size_t(y) * w + x] = s;
    }
  }
  ImgU8 dst(w, h, 1);
  const int vecEnd = (w / 4) * 4;
  for (int y = 0; y < h; ++y) {
    for (int x = 0; x < vecEnd; ++x) {
      float s = (float)tmp[size_t(y) * w + x] * fk[0] + 0.0f;
      for (int j = 1; j <= r; ++j)
        s = s + (float)(tmp[size_t(reflect101(y + j, h)) * w + x] + tmp[size_t(reflect101(y - j, h)) * w + x]) * fk[j];
      dst.at(y, x) = satU8(cvRoundF(s));
    }
    for (int x = vecEnd; x < w; ++x) {
      int s = ik[r] * tmp[size_t(y) * w + x];
      for (int j = 1; j <= r; ++j)
        s += ik[r + j] * (tmp[size_t(reflect101(y + j, h)) * w + x] + tmp[size_t(reflect101(y - j, h)) * w + x]);
      dst.at(y, x) = satU8((s + (1 << 15)) >> 16);
    }
  }
  return dst;
}

// ---------------------------------------------------------------------------
// Sobel ksize=1: [-1 0 1], BORDER_REPLICATE, no scale.
static inline ImgF sobelX(const ImgF& I) {
  ImgF D(I.w, I.h, 1);
  for (int y = 0; y < I.h; ++y)
    for (int x = 0; x < I.w; ++x)
      D.at(y, x) = I.at(y, std::min(x + 1, I.w - 1)) - I.at(y, std::max(x - 1, 0));
  return D;
}
static inline ImgF sobelY(const ImgF& I) {
  ImgF D(I.w, I.h, 1);
  for (int y = 0; y < I.h; ++y)
    for (int x = 0; x < I.w; ++x)
      D.at(y, x) = I.at(std::min(y + 1, I.h - 1), x) - I.at(std::max(y - 1, 0), x);
  return D;
}

// medianBlur(5) per channel on float images, replicate border (App. A.5).
static inline ImgF medianBlur5(const ImgF& src) {
  ImgF dst(src.w, src.h, src.c);
  float v[25];
  for (int y = 0; y < src.h; ++y)
    for (int x = 0; x < src.w; ++x)
      for (int k = 0; k < src.c; ++k) {
        int n = 0;
        for (int dy = -2; dy <= 2; ++dy)
          for (int dx = -2; dx <= 2; ++dx)
            v[n++] = src.at(clipIdx(y + dy, 0, src.h), clipIdx(x + dx, 0, src.w), k);
        std::nth_element(v, v + 12, v + 25);
        dst.at(y, x, k) = v[12];
      }
  return dst;
}

// ---------------------------------------------------------------------------
// Colour conversion.
static inline uint8_t bgr2gray(uint8_t b, uint8_t g, uint8_t r) {
  return (uint8_t)((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14);
}
static inline ImgU8 bgr2bgra(const ImgU8& src) {
  assert(src.c == 3);
  ImgU8 d(src.w, src.h, 4);
  for (size_t i = 0, n = size_t(src.w) * src.h; i < n; ++i) {
    d.d[i * 4] = src.d[i * 3];
    d.d[i * 4 + 1] = src.d[i * 3 + 1];
    d.d[i * 4 + 2] = src.d[i * 3 + 2];
    d.d[i * 4 + 3] = 255;
  }
  return d;
}
static inline ImgU8 bgra2bgr(const ImgU8& src) {
  assert(src.c == 4);
  ImgU8 d(src.w, src.h, 3);
  for (size_t i = 0, n = size_t(src.w) * src.h; i < n; ++i) {
    d.d[i * 3] = src.d[i * 4];
    d.d[i * 3 + 1] = src.d[i * 4 + 1];
    d.d[i * 3 + 2] = src.d[i * 4 + 2];
  }
  return d;
}

// erode with MORPH_CROSS (2e+1)^2 on one channel; outside = +inf (App. A.7).
static inline ImgU8 erodeCrossU8C1(const ImgU8& src, int e) {
  ImgU8 dst(src.w, src.h, 1);
  for (int y = 0; y < src.h; ++y)
    for (int x = 0; x < src.w; ++x) {
      int m = 255;
      for (int xx = std::max(0, x - e); xx <= std::min(src.w - 1, x + e); ++xx) m = std::min<int>(m, src.at(y, xx));
      for (int yy = std::max(0, y - e); yy <= std::min(src.h - 1, y + e); ++yy) m = std::min<int>(m, src.at(yy, x));
      dst.at(y, x) = (uint8_t)m;
    }
  return dst;
}

// flip(src, -1): both axes.
template <typename T>
static inline Img<T> flipBoth(const Img<T>& s) {
  Img<T> d(s.w, s.h, s.c);
  for (int y = 0; y < s.h; ++y)
    for (int x = 0; x < s.w; ++x)
      for (int k = 0; k < s.c; ++k) d.at(s.h - 1 - y, s.w - 1 - x, k) = s.at(y, x, k);
  return d;
}

// copyMakeBorder(top,bottom,0,0,BORDER_CONSTANT 0)
template <typename T>
static inline Img<T> padRows(const Img<T>& s, int top, int bottom) {
  Img<T> d(s.w, s.h + top + bottom, s.c, T(0));
  std::memcpy(d.row(top), s.d.data(), s.bytes());
  return d;
}

template <typename T>
static inline Img<T> cropCols(const Img<T>& s, int x0, int w) {
  Img<T> d(w, s.h, s.c);
  for (int y = 0; y < s.h; ++y) std::memcpy(d.row(y), s.px(y, x0), size_t(w) * s.c * sizeof(T));
  return d;
}
template <typename T>
static inline Img<T> cropRows(const Img<T>& s, int y0, int h) {
  Img<T> d(s.w, h, s.c);
  std::memcpy(d.d.data(), s.row(y0), d.bytes());
  return d;
}

}  // namespace orc
