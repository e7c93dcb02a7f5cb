// TEST INFRASTRUCTURE. The imgproc ALGORITHMS the reference's optical-flow sources call, routed to the oracle's own
// restatements (oracle/cvlite.h) — see the note in opencv2/core.hpp of this directory. Only the argument combinations
// those sources use are accepted; anything else aborts.
#pragma once
#include <cstdlib>
#include <vector>

#include "../../cvlite.h"
#include "opencv2/core.hpp"

namespace cv {

namespace shim {
inline orc::ImgF to_imgf(const Mat& m) {
  assert(m.depth() == CV_32F);
  orc::ImgF r(m.cols, m.rows, m.channels());
  std::memcpy(r.d.data(), m.data, r.bytes());
  return r;
}
inline orc::ImgU8 to_imgu8(const Mat& m) {
  assert(m.depth() == CV_8U);
  orc::ImgU8 r(m.cols, m.rows, m.channels());
  std::memcpy(r.d.data(), m.data, r.bytes());
  return r;
}
inline Mat from_img(const orc::ImgF& i) {
  Mat m(i.h, i.w, CV_MAKETYPE(CV_32F, i.c));
  std::memcpy(m.data, i.d.data(), i.bytes());
  return m;
}
inline Mat from_img(const orc::ImgU8& i) {
  Mat m(i.h, i.w, CV_MAKETYPE(CV_8U, i.c));
  std::memcpy(m.data, i.d.data(), i.bytes());
  return m;
}
[[noreturn]] inline void unsupported(const char* what) {
  std::fprintf(stderr, "ref_shim: %s is not provided\n", what);
  std::abort();
}
}  // namespace shim

inline Mat imread(const std::string&, int = IMREAD_COLOR) { shim::unsupported("imread"); }
inline bool imwrite(const std::string&, const Mat&, const std::vector<int>& = std::vector<int>()) { shim::unsupported("imwrite"); }

// remap as the stereo path calls it: float (x, y) map in map1, no map2.
//   INTER_CUBIC, BORDER_CONSTANT(0): CV_8UC4 / CV_8UC3 images and CV_32FC2 flows -> cvlite's restatements
//   INTER_NEAREST, BORDER_WRAP (offsetHorizontalWrap): source = (cvRound(x), cvRound(y)) wrapped
inline void remap(const Mat& src, Mat& dst, const Mat& map1, const Mat& map2, int interpolation,
                  int borderMode = BORDER_CONSTANT) {
  if (!map2.empty() || map1.type() != CV_32FC2) shim::unsupported("this remap map format");
  const orc::ImgF mp = shim::to_imgf(map1);
  if (interpolation == INTER_CUBIC && borderMode == BORDER_CONSTANT && src.depth() == CV_8U) {
    dst = shim::from_img(orc::remapCubicU8(shim::to_imgu8(src), mp));
  } else if (interpolation == INTER_CUBIC && borderMode == BORDER_CONSTANT && src.depth() == CV_32F) {
    dst = shim::from_img(orc::remapCubicF32(shim::to_imgf(src), mp));
  } else if (interpolation == INTER_NEAREST && borderMode == BORDER_WRAP && src.depth() == CV_8U) {
    Mat d(map1.rows, map1.cols, src.type());
    const size_t es = elemSize(src.type());
    for (int y = 0; y < d.rows; ++y)
      for (int x = 0; x < d.cols; ++x) {
        const Point2f m = map1.at<Point2f>(y, x);
        int sx = orc::cvRoundF(m.x), sy = orc::cvRoundF(m.y);
        sx = ((sx % src.cols) + src.cols) % src.cols;
        sy = ((sy % src.rows) + src.rows) % src.rows;
        std::memcpy(d.data + ((size_t)y * d.cols + x) * es, src.data + ((size_t)sy * src.cols + sx) * es, es);
      }
    dst = d;
  } else {
    shim::unsupported("this remap variant");
  }
}
inline void hconcat(const Mat& a, const Mat& b, Mat& dst) {
  assert(a.rows == b.rows && a.type() == b.type());
  Mat d(a.rows, a.cols + b.cols, a.type());
  const size_t es = elemSize(a.type());
  for (int y = 0; y < a.rows; ++y) {
    std::memcpy(d.data + (size_t)y * d.cols * es, a.data + (size_t)y * a.cols * es, a.cols * es);
    std::memcpy(d.data + ((size_t)y * d.cols + a.cols) * es, b.data + (size_t)y * b.cols * es, b.cols * es);
  }
  dst = d;
}
inline void vconcat(const Mat& a, const Mat& b, Mat& dst) {
  assert(a.cols == b.cols && a.type() == b.type());
  Mat d(a.rows + b.rows, a.cols, a.type());
  const size_t es = elemSize(a.type());
  std::memcpy(d.data, a.data, a.total() * es);
  std::memcpy(d.data + a.total() * es, b.data, b.total() * es);
  dst = d;
}
inline void flip(const Mat& src, const Mat& dst, int code) {  // in place on shared data, as OpenCV does for dst == src
  if (code != 1 || src.data != dst.data) shim::unsupported("this flip variant");
  const size_t es = elemSize(src.type());
  std::vector<uchar> tmp(es);
  for (int y = 0; y < src.rows; ++y)
    for (int x = 0; x < src.cols / 2; ++x) {
      uchar* a = src.data + ((size_t)y * src.cols + x) * es;
      uchar* b = src.data + ((size_t)y * src.cols + src.cols - 1 - x) * es;
      std::memcpy(tmp.data(), a, es); std::memcpy(a, b, es); std::memcpy(b, tmp.data(), es);
    }
}
inline Mat getStructuringElement(int shape, Size ksize, Point anchor = Point(-1, -1)) {
  if (shape != MORPH_CROSS || ksize.width != ksize.height || !(ksize.width & 1) || anchor.x != ksize.width / 2 || anchor.y != anchor.x)
    shim::unsupported("this structuring element");
  return Mat(ksize.height, ksize.width, CV_8UC1);  // only its size is read (erode below)
}
inline void erode(const Mat& src, Mat& dst, const Mat& kernel) {  // centred cross, 8-bit single channel
  if (src.type() != CV_8UC1) shim::unsupported("this erode variant");
  dst = shim::from_img(orc::erodeCrossU8C1(shim::to_imgu8(src), kernel.cols / 2));
}
inline void merge(const std::vector<Mat>& mv, Mat& dst) {
  const int cn = (int)mv.size();
  Mat d(mv[0].rows, mv[0].cols, CV_MAKETYPE(mv[0].depth(), cn));
  const size_t es = elemSize(mv[0].type());
  for (int k = 0; k < cn; ++k)
    for (size_t i = 0; i < d.total(); ++i) std::memcpy(d.data + (i * cn + k) * es, mv[k].data + i * es, es);
  dst = d;
}

inline void resize(const Mat& src, Mat& dst, Size dsize, double fx = 0, double fy = 0, int interpolation = INTER_LINEAR) {
  if (fx != 0 || fy != 0) shim::unsupported("resize with scale factors");
  if (src.depth() == CV_8U && interpolation == INTER_CUBIC) dst = shim::from_img(orc::resizeCubicU8(shim::to_imgu8(src), dsize.width, dsize.height));
  else if (src.depth() == CV_32F && interpolation == INTER_CUBIC) dst = shim::from_img(orc::resizeCubicF32(shim::to_imgf(src), dsize.width, dsize.height));
  else if (src.depth() == CV_32F && interpolation == INTER_LINEAR) dst = shim::from_img(orc::resizeLinearF32(shim::to_imgf(src), dsize.width, dsize.height));
  else shim::unsupported("this resize variant");
}
inline void GaussianBlur(const Mat& src, Mat& dst, Size ksize, double sigmaX, double sigmaY = 0, int borderType = BORDER_REFLECT_101) {
  if (ksize.width != ksize.height || sigmaY != 0 || borderType != BORDER_REFLECT_101) shim::unsupported("this GaussianBlur variant");
  if (src.depth() == CV_32F) dst = shim::from_img(orc::gaussianBlurF32(shim::to_imgf(src), ksize.width, sigmaX));
  else if (src.type() == CV_8UC1) dst = shim::from_img(orc::gaussianBlurU8C1(shim::to_imgu8(src), ksize.width, sigmaX));
  else shim::unsupported("this GaussianBlur variant");
}
inline void Sobel(const Mat& src, Mat& dst, int ddepth, int dx, int dy, int ksize = 3, double scale = 1, double delta = 0, int borderType = BORDER_REFLECT_101) {
  if (src.type() != CV_32F || ksize != 1 || scale != 1 || delta != 0 || borderType != BORDER_REPLICATE || dx + dy != 1 || (ddepth != -1 && ddepth != CV_32F))
    shim::unsupported("this Sobel variant");
  dst = shim::from_img(dx ? orc::sobelX(shim::to_imgf(src)) : orc::sobelY(shim::to_imgf(src)));
}
inline void medianBlur(const Mat& src, Mat& dst, int ksize) {
  if (ksize != 5 || src.depth() != CV_32F) shim::unsupported("this medianBlur variant");
  dst = shim::from_img(orc::medianBlur5(shim::to_imgf(src)));
}
inline void split(const Mat& src, std::vector<Mat>& mv) {
  const int cn = src.channels();
  mv.assign(cn, Mat());
  const size_t es = elemSize(src.type()) / cn;
  for (int k = 0; k < cn; ++k) {
    Mat m(src.rows, src.cols, CV_MAKETYPE(src.depth(), 1));
    for (size_t i = 0; i < src.total(); ++i) std::memcpy(m.data + i * es, src.data + (i * cn + k) * es, es);
    mv[k] = m;
  }
}
inline void cvtColor(const Mat& src, Mat& dst, int code) {
  if (code != CV_BGRA2GRAY || src.type() != CV_8UC4) shim::unsupported("this cvtColor variant");
  Mat m(src.rows, src.cols, CV_8UC1);
  for (size_t i = 0; i < src.total(); ++i) m.data[i] = (uchar)orc::bgr2gray(src.data[4 * i], src.data[4 * i + 1], src.data[4 * i + 2]);
  dst = m;
}

}  // namespace cv
