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
inline Mat cont(const Mat& m) { return m.isContinuous() ? m : m.clone(); }  // the functions below index packed rows
inline orc::ImgF to_imgf(const Mat& m0) {
  assert(m0.depth() == CV_32F);
  const Mat m = cont(m0);
  orc::ImgF r(m.cols, m.rows, m.channels());
  std::memcpy(r.d.data(), m.data, r.bytes());
  return r;
}
inline orc::ImgU8 to_imgu8(const Mat& m0) {
  assert(m0.depth() == CV_8U);
  const Mat m = cont(m0);
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

// imread / imwrite for 8-bit PNG files through the repository's own PNG codec (host/png_io.hpp; checked against PIL in
// tests/test_cpu_host.py): IMREAD_COLOR -> B,G,R; IMREAD_UNCHANGED keeps an alpha channel; imwrite stores B,G,R(,A) or grey.
Mat imread(const std::string& path, int flags = IMREAD_COLOR);
bool imwrite(const std::string& path, const Mat& img, const std::vector<int>& params = std::vector<int>());

// remap as the stereo path calls it: float (x, y) map in map1, no map2.
//   INTER_CUBIC, BORDER_CONSTANT(0): CV_8UC4 / CV_8UC3 images and CV_32FC2 flows -> cvlite's restatements
//   INTER_NEAREST, BORDER_WRAP (offsetHorizontalWrap): source = (cvRound(x), cvRound(y)) wrapped
inline void remap(const Mat& src0, Mat& dst, const Mat& map10, const Mat& map2, int interpolation,
                  int borderMode = BORDER_CONSTANT) {
  const Mat src = shim::cont(src0), map1 = shim::cont(map10);
  if (!map2.empty() || map1.type() != CV_32FC2) shim::unsupported("this remap map format");
  const orc::ImgF mp = shim::to_imgf(map1);
  if (interpolation == INTER_CUBIC && borderMode == BORDER_CONSTANT && src.depth() == CV_8U) {
    dst = shim::from_img(orc::remapCubicU8(shim::to_imgu8(src), mp));
  } else if (interpolation == INTER_CUBIC && borderMode == BORDER_WRAP && src.depth() == CV_8U) {  // cubemap faces
    dst = shim::from_img(orc::remapCubicU8Wrap(shim::to_imgu8(src), mp));
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
inline void hconcat(const Mat& a0, const Mat& b0, Mat& dst) {
  const Mat a = shim::cont(a0), b = shim::cont(b0);
  assert(a.rows == b.rows && a.type() == b.type());
  Mat d(a.rows, a.cols + b.cols, a.type());
  const size_t es = elemSize(a.type());
  for (int y = 0; y < a.rows; ++y) {
    std::memcpy(d.data + (size_t)y * d.cols * es, a.data + (size_t)y * a.cols * es, a.cols * es);
    std::memcpy(d.data + ((size_t)y * d.cols + a.cols) * es, b.data + (size_t)y * b.cols * es, b.cols * es);
  }
  dst = d;
}
inline void vconcat(const Mat& a0, const Mat& b0, Mat& dst) {
  const Mat a = shim::cont(a0), b = shim::cont(b0);
  assert(a.cols == b.cols && a.type() == b.type());
  Mat d(a.rows + b.rows, a.cols, a.type());
  const size_t es = elemSize(a.type());
  std::memcpy(d.data, a.data, a.total() * es);
  std::memcpy(d.data + a.total() * es, b.data, b.total() * es);
  dst = d;
}
// flip: code 1 = around the vertical axis, -1 = both axes; in place (dst shares src's pixels) or into dst
inline void flip(const Mat& src0, Mat& dst, int code) {
  if (code != 1 && code != -1) shim::unsupported("this flip variant");
  const Mat src = shim::cont(src0);
  const size_t es = elemSize(src.type());
  Mat d(src.rows, src.cols, src.type());
  for (int y = 0; y < src.rows; ++y)
    for (int x = 0; x < src.cols; ++x)
      std::memcpy(d.data + ((size_t)y * d.cols + x) * es,
                  src.data + ((size_t)(code == -1 ? src.rows - 1 - y : y) * src.cols + src.cols - 1 - x) * es, es);
  if (dst.data == src0.data) for (int y = 0; y < d.rows; ++y) std::memcpy(dst.data + (size_t)y * dst.step, d.data + (size_t)y * d.step, d.step);
  else dst = d;
}
inline void flip(const Mat& src, const Mat& dst, int code) {  // (OutputArray accepts a const Mat: in place on shared pixels only)
  if (dst.data != src.data) shim::unsupported("flip into a const destination");
  Mat d;
  flip(src, d, code);
  for (int y = 0; y < d.rows; ++y) std::memcpy(dst.data + (size_t)y * dst.step, d.data + (size_t)y * d.step, d.step);
}
inline void copyMakeBorder(const Mat& src0, Mat& dst, int top, int bottom, int left, int right, int borderType, const Scalar& value = Scalar()) {
  const Mat src = shim::cont(src0);
  if (borderType != BORDER_CONSTANT || src.depth() != CV_8U) shim::unsupported("this copyMakeBorder variant");
  const int cn = src.channels();
  Mat d(src.rows + top + bottom, src.cols + left + right, src.type());
  for (size_t i = 0; i < d.total(); ++i)
    for (int k = 0; k < cn; ++k) d.data[i * cn + k] = (uchar)(value.val[k] < 0 ? 0 : value.val[k] > 255 ? 255 : (int)(value.val[k] + 0.5));
  for (int y = 0; y < src.rows; ++y)
    std::memcpy(d.data + ((size_t)(y + top) * d.cols + left) * cn, src.data + (size_t)y * src.cols * cn, (size_t)src.cols * cn);
  dst = d;
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
inline void merge(const std::vector<Mat>& mv0, Mat& dst) {
  std::vector<Mat> mv;
  for (const Mat& m : mv0) mv.push_back(shim::cont(m));
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
inline void split(const Mat& src0, std::vector<Mat>& mv) {
  const Mat src = shim::cont(src0);
  const int cn = src.channels();
  mv.assign(cn, Mat());
  const size_t es = elemSize(src.type()) / cn;
  for (int k = 0; k < cn; ++k) {
    Mat m(src.rows, src.cols, CV_MAKETYPE(src.depth(), 1));
    for (size_t i = 0; i < src.total(); ++i) std::memcpy(m.data + i * es, src.data + (i * cn + k) * es, es);
    mv[k] = m;
  }
}
inline void cvtColor(const Mat& src0, Mat& dst, int code) {
  const Mat src = shim::cont(src0);
  if (code == CV_BGRA2GRAY && src.type() == CV_8UC4) {
    Mat m(src.rows, src.cols, CV_8UC1);
    for (size_t i = 0; i < src.total(); ++i) m.data[i] = (uchar)orc::bgr2gray(src.data[4 * i], src.data[4 * i + 1], src.data[4 * i + 2]);
    dst = m;
  } else if (code == CV_BGR2BGRA && src.type() == CV_8UC3) {  // alpha = 255
    Mat m(src.rows, src.cols, CV_8UC4);
    for (size_t i = 0; i < src.total(); ++i) {
      m.data[4 * i] = src.data[3 * i]; m.data[4 * i + 1] = src.data[3 * i + 1]; m.data[4 * i + 2] = src.data[3 * i + 2]; m.data[4 * i + 3] = 255;
    }
    dst = m;
  } else if (code == CV_BGRA2BGR && src.type() == CV_8UC4) {
    Mat m(src.rows, src.cols, CV_8UC3);
    for (size_t i = 0; i < src.total(); ++i) {
      m.data[3 * i] = src.data[4 * i]; m.data[3 * i + 1] = src.data[4 * i + 1]; m.data[3 * i + 2] = src.data[4 * i + 2];
    }
    dst = m;
  } else {
    shim::unsupported("this cvtColor variant");
  }
}

}  // namespace cv
