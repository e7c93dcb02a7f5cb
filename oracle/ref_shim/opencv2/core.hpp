// TEST INFRASTRUCTURE. Container-only stand-in for the OpenCV types the reference's soft ISP (camera_isp/CameraIsp.h and
// the util headers it includes) is written against, so that the reference's OWN SOURCE can be compiled from
// /root/reference into oracle/_ref without OpenCV. It provides storage and indexing (Mat, Vec, Point3_, Size), the
// element-wise Vec arithmetic those headers use, and the three matrix operations of CameraIsp::setup on 3x3 float
// matrices. Every ISP arithmetic operation on pixels is the reference's code, not this file's.
//
// From memory of OpenCV 3.x (not verifiable offline), stated here because the composite CCM depends on it:
//   * Mat * Mat on 3x3 CV_32F takes gemm's unrolled small-matrix path: float products summed left to right in float
//   * Mat *= double scales every element in float (convertTo with a float working type)
//   * Vec<float,n> * float and Vec + Vec are plain per-element float operations
//   * Mat *= s, Mat /= s, Mat * s on CV_32F data: every element times (float)s resp. (float)(1.0 / s) (convertTo)
//   * norm(Point2f) = sqrt((double)x * x + (double)y * y)
// opencv2/imgproc.hpp of this directory routes the imgproc ALGORITHMS the optical-flow sources call (resize,
// GaussianBlur, Sobel, medianBlur, cvtColor, split) to the oracle's own restatements in cvlite.h: compiling the
// reference's PixFlow.h over it checks the oracle's restatement of PixFlow's logic against the reference's source,
// with the OpenCV primitives common to both sides (they stay unpinned, see cvlite.h).
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace cv {

using std::string;  // name lookup only: the reference's headers find these through `using namespace cv`
using std::vector;
using std::min;
using std::max;
namespace detail {}

typedef unsigned char uchar;
typedef unsigned short ushort;

enum { CV_8U = 0, CV_8S = 1, CV_16U = 2, CV_16S = 3, CV_32S = 4, CV_32F = 5, CV_64F = 6 };
#define CV_CN_SHIFT 3
#define CV_MAT_DEPTH_MASK 7
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << CV_CN_SHIFT))
#define CV_8UC1 CV_MAKETYPE(cv::CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(cv::CV_8U, 3)
#define CV_8UC4 CV_MAKETYPE(cv::CV_8U, 4)
#define CV_16UC1 CV_MAKETYPE(cv::CV_16U, 1)
#define CV_16UC3 CV_MAKETYPE(cv::CV_16U, 3)
#define CV_32FC1 CV_MAKETYPE(cv::CV_32F, 1)
#define CV_32FC2 CV_MAKETYPE(cv::CV_32F, 2)
#define CV_32FC3 CV_MAKETYPE(cv::CV_32F, 3)
#define CV_32FC4 CV_MAKETYPE(cv::CV_32F, 4)
#define CV_8UC2 CV_MAKETYPE(cv::CV_8U, 2)
#define CV_64FC1 CV_MAKETYPE(cv::CV_64F, 1)
enum { IMREAD_UNCHANGED = -1, IMREAD_GRAYSCALE = 0, IMREAD_COLOR = 1 };
enum { MORPH_RECT = 0, MORPH_CROSS = 1, MORPH_ELLIPSE = 2 };
enum { INTER_NEAREST = 0, INTER_LINEAR = 1, INTER_CUBIC = 2 };
enum { BORDER_CONSTANT = 0, BORDER_REPLICATE = 1, BORDER_REFLECT = 2, BORDER_WRAP = 3, BORDER_REFLECT_101 = 4 };
enum { COLOR_BGRA2GRAY = 10, COLOR_BGR2BGRA = 0, COLOR_BGRA2BGR = 1 };
#define CV_INTER_LINEAR 1
#define CV_INTER_CUBIC 2
#define CV_BGRA2GRAY 10
#define CV_BGR2BGRA 0
#define CV_BGRA2BGR 1
#define CV_LOAD_IMAGE_COLOR 1
#define CV_LOAD_IMAGE_UNCHANGED -1
#define CV_LOAD_IMAGE_GRAYSCALE 0
#define CV_LOAD_IMAGE_ANYDEPTH 2

template <typename T, int N>
struct Vec {
  T val[N];
  Vec() { for (int i = 0; i < N; ++i) val[i] = T(0); }
  Vec(T a, T b) { static_assert(N == 2, ""); val[0] = a; val[1] = b; }
  Vec(T a, T b, T c) { static_assert(N == 3, ""); val[0] = a; val[1] = b; val[2] = c; }
  Vec(T a, T b, T c, T d) { static_assert(N == 4, ""); val[0] = a; val[1] = b; val[2] = c; val[3] = d; }
  template <typename U>
  Vec(const Vec<U, N>& o) { for (int i = 0; i < N; ++i) val[i] = T(o.val[i]); }
  T& operator[](int i) { return val[i]; }
  const T& operator[](int i) const { return val[i]; }
};
template <typename T, int N>
inline Vec<T, N> operator+(const Vec<T, N>& a, const Vec<T, N>& b) { Vec<T, N> r; for (int i = 0; i < N; ++i) r.val[i] = T(a.val[i] + b.val[i]); return r; }
template <typename T, int N>
inline Vec<T, N> operator-(const Vec<T, N>& a, const Vec<T, N>& b) { Vec<T, N> r; for (int i = 0; i < N; ++i) r.val[i] = T(a.val[i] - b.val[i]); return r; }
template <typename T, int N>
inline Vec<T, N> operator*(const Vec<T, N>& a, float s) { Vec<T, N> r; for (int i = 0; i < N; ++i) r.val[i] = T(a.val[i] * s); return r; }
template <typename T, int N>
inline Vec<T, N> operator*(float s, const Vec<T, N>& a) { return a * s; }
template <typename T, int N>
inline Vec<T, N>& operator+=(Vec<T, N>& a, const Vec<T, N>& b) { for (int i = 0; i < N; ++i) a.val[i] = T(a.val[i] + b.val[i]); return a; }
template <typename T, int N>
inline Vec<T, N> operator/(const Vec<T, N>& a, float s) { Vec<T, N> r; for (int i = 0; i < N; ++i) r.val[i] = T(a.val[i] * (1.f / s)); return r; }
template <typename T, int N>
inline bool operator==(const Vec<T, N>& a, const Vec<T, N>& b) { for (int i = 0; i < N; ++i) if (a.val[i] != b.val[i]) return false; return true; }
template <typename T, int N>
inline bool operator!=(const Vec<T, N>& a, const Vec<T, N>& b) { return !(a == b); }
template <typename T, int N>
inline std::ostream& operator<<(std::ostream& o, const Vec<T, N>& v) { o << "["; for (int i = 0; i < N; ++i) o << (i ? ", " : "") << v.val[i]; return o << "]"; }
typedef Vec<uchar, 3> Vec3b;
typedef Vec<uchar, 4> Vec4b;
typedef Vec<short, 3> Vec3s;
typedef Vec<ushort, 3> Vec3w;
typedef Vec<float, 2> Vec2f;
typedef Vec<float, 3> Vec3f;
typedef Vec<float, 4> Vec4f;
typedef Vec<double, 3> Vec3d;
// norm(Matx): L2 norm accumulated in double
template <typename T, int N>
inline double norm(const Vec<T, N>& v) { double s = 0; for (int i = 0; i < N; ++i) s += (double)v.val[i] * (double)v.val[i]; return std::sqrt(s); }

template <typename T>
struct Point_ {
  T x, y;
  Point_() : x(0), y(0) {}
  Point_(T a, T b) : x(a), y(b) {}
  T dot(const Point_& o) const { return T(x * o.x + y * o.y); }
};
typedef Point_<int> Point;
typedef Point_<float> Point2f;
template <typename T> inline Point_<T> operator+(const Point_<T>& a, const Point_<T>& b) { return Point_<T>(T(a.x + b.x), T(a.y + b.y)); }
template <typename T> inline Point_<T> operator-(const Point_<T>& a, const Point_<T>& b) { return Point_<T>(T(a.x - b.x), T(a.y - b.y)); }
template <typename T> inline Point_<T> operator*(const Point_<T>& a, float s) { return Point_<T>(T(a.x * s), T(a.y * s)); }
template <typename T> inline Point_<T> operator*(float s, const Point_<T>& a) { return Point_<T>(T(a.x * s), T(a.y * s)); }
template <typename T> inline Point_<T>& operator-=(Point_<T>& a, const Point_<T>& b) { a.x = T(a.x - b.x); a.y = T(a.y - b.y); return a; }
template <typename T> inline Point_<T>& operator+=(Point_<T>& a, const Point_<T>& b) { a.x = T(a.x + b.x); a.y = T(a.y + b.y); return a; }
template <typename T> inline Point_<T>& operator/=(Point_<T>& a, float s) { a.x = T(a.x / s); a.y = T(a.y / s); return a; }
template <typename T> inline Point_<T> operator/(const Point_<T>& a, float s) { Point_<T> t(a); t /= s; return t; }
template <typename T> inline double norm(const Point_<T>& p) { return std::sqrt((double)p.x * p.x + (double)p.y * p.y); }
template <typename T>
struct Point3_ {
  T x, y, z;
  Point3_() : x(0), y(0), z(0) {}
  Point3_(T a, T b, T c) : x(a), y(b), z(c) {}
  Point3_(const Vec<T, 3>& v) : x(v.val[0]), y(v.val[1]), z(v.val[2]) {}
  operator Vec<T, 3>() const { return Vec<T, 3>(x, y, z); }
};
typedef Point3_<float> Point3f;
template <typename T>
inline std::ostream& operator<<(std::ostream& o, const Point3_<T>& p) { return o << "[" << p.x << ", " << p.y << ", " << p.z << "]"; }
template <typename T>
inline std::ostream& operator<<(std::ostream& o, const std::vector<Point3_<T>>& v) { for (auto& p : v) o << p << ";"; return o; }

struct Size {
  int width, height;
  Size() : width(0), height(0) {}
  Size(int w, int h) : width(w), height(h) {}
  bool operator==(const Size& o) const { return width == o.width && height == o.height; }
  bool operator!=(const Size& o) const { return !(*this == o); }
};
struct Rect {
  int x, y, width, height;
  Rect() : x(0), y(0), width(0), height(0) {}
  Rect(int a, int b, int c, int d) : x(a), y(b), width(c), height(d) {}
};
struct Scalar {
  double val[4];
  Scalar(double a = 0, double b = 0, double c = 0, double d = 0) { val[0] = a; val[1] = b; val[2] = c; val[3] = d; }
  double operator[](int i) const { return val[i]; }
};

inline size_t elemSize(int type) {
  static const int d[7] = {1, 1, 2, 2, 4, 4, 8};
  return (size_t)d[type & CV_MAT_DEPTH_MASK] * (size_t)((type >> CV_CN_SHIFT) + 1);
}

class Mat {
 public:
  int rows, cols, dims;
  uchar* data;
  size_t step;  // bytes from one row to the next (a region of interest keeps its parent's)
  Mat() : rows(0), cols(0), dims(0), data(nullptr), step(0), type_(0) {}
  Mat(int r, int c, int type) { create(r, c, type); }
  Mat(Size s, int type) { create(s.height, s.width, type); }
  Mat(int r, int c, int type, void* ext) : rows(r), cols(c), dims(2), data((uchar*)ext), step((size_t)c * elemSize(type)), type_(type) {}  // wraps, does not copy
  void create(int r, int c, int type) {
    rows = r; cols = c; type_ = type; dims = 2;
    step = (size_t)c * elemSize(type);
    store_.reset(new std::vector<uchar>((size_t)r * c * elemSize(type) + 64));  // uninitialised in OpenCV; zero here
    data = store_->data();
  }
  void create(Size s, int type) { create(s.height, s.width, type); }  // cv::Mat::create(Size, int)
  bool isContinuous() const { return step == (size_t)cols * elemSize(type_); }
  static Mat zeros(int r, int c, int type) { return Mat(r, c, type); }
  static Mat zeros(Size s, int type) { return Mat(s.height, s.width, type); }
  template <typename T> T* ptr(int y) { return reinterpret_cast<T*>(data + (size_t)y * step); }
  template <typename T> const T* ptr(int y) const { return reinterpret_cast<const T*>(data + (size_t)y * step); }
  size_t total() const { return (size_t)rows * cols; }
  // convertTo(dst, CV_32F) from 8-bit data (PixFlow.h:128-133): exact
  void convertTo(Mat& dst, int rtype) const {
    assert(depth() == CV_8U && (rtype & CV_MAT_DEPTH_MASK) == CV_32F);
    const Mat s = isContinuous() ? *this : clone();
    Mat d(rows, cols, CV_MAKETYPE(CV_32F, channels()));
    const size_t n = total() * channels();
    for (size_t i = 0; i < n; ++i) reinterpret_cast<float*>(d.data)[i] = (float)s.data[i];
    dst = d;
  }
  static Mat eye(int r, int c, int type) {
    Mat m(r, c, type);
    assert(type == CV_32F);
    for (int i = 0; i < std::min(r, c); ++i) m.at<float>(i, i) = 1.0f;
    return m;
  }
  int type() const { return type_; }
  int depth() const { return type_ & CV_MAT_DEPTH_MASK; }
  int channels() const { return (type_ >> CV_CN_SHIFT) + 1; }
  Size size() const { return Size(cols, rows); }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  Mat clone() const {  // always continuous
    Mat m(rows, cols, type_);
    for (int y = 0; y < rows; ++y) std::memcpy(m.data + (size_t)y * m.step, data + (size_t)y * step, m.step);
    return m;
  }
  // A region of interest: a view that shares the parent's pixels and row step, as in OpenCV. (at(y, x) with x beyond the
  // view's width then lands in the parent's row, not in the view's next row — TestRenderStereoPanorama.cpp:530-535
  // walks a view with the parent's width.)
  Mat operator()(const Rect& r) const {
    assert(r.x >= 0 && r.y >= 0 && r.x + r.width <= cols && r.y + r.height <= rows);
    Mat m;
    m.rows = r.height; m.cols = r.width; m.dims = 2; m.type_ = type_; m.step = step; m.store_ = store_;
    m.data = data + (size_t)r.y * step + (size_t)r.x * elemSize(type_);
    return m;
  }
  void copyTo(Mat& dst) const { dst = clone(); }
  Mat& operator=(const Scalar& s) {  // every element set to the scalar (integer depths: saturate_cast of a finite value)
    const int cn = channels(), d = depth();
    for (int y = 0; y < rows; ++y)
      for (int x = 0; x < cols; ++x)
        for (int k = 0; k < cn; ++k) {
          uchar* e = data + (size_t)y * step + ((size_t)x * cn + k) * (elemSize(type_) / cn);
          const double v = s.val[k < 4 ? k : 3];
          if (d == CV_8U) *e = (uchar)(v < 0 ? 0 : v > 255 ? 255 : std::lrint(v));
          else if (d == CV_16U) *reinterpret_cast<ushort*>(e) = (ushort)(v < 0 ? 0 : v > 65535 ? 65535 : std::lrint(v));
          else if (d == CV_32F) *reinterpret_cast<float*>(e) = (float)v;
          else assert(!"Mat = Scalar: depth not provided");
        }
    return *this;
  }
  template <typename T> T& at(int i, int j) { return *reinterpret_cast<T*>(data + (size_t)i * step + (size_t)j * sizeof(T)); }
  template <typename T> const T& at(int i, int j) const { return *reinterpret_cast<const T*>(data + (size_t)i * step + (size_t)j * sizeof(T)); }
  template <typename T> T& at(int i) { return reinterpret_cast<T*>(data)[i]; }
  template <typename T> const T& at(int i) const { return reinterpret_cast<const T*>(data)[i]; }

 private:
  int type_;
  std::shared_ptr<std::vector<uchar>> store_;
};
inline std::ostream& operator<<(std::ostream& o, const Mat& m) {
  if (m.type() == CV_32F) for (int i = 0; i < m.rows; ++i) for (int j = 0; j < m.cols; ++j) o << m.at<float>(i, j) << (j + 1 == m.cols ? ";" : ",");
  return o;
}
// gemm's small-matrix path for CV_32F (see the header note)
inline Mat operator*(const Mat& a, const Mat& b) {
  assert(a.type() == CV_32F && b.type() == CV_32F && a.cols == b.rows);
  Mat d(a.rows, b.cols, CV_32F);
  for (int i = 0; i < a.rows; ++i)
    for (int j = 0; j < b.cols; ++j) {
      float t = a.at<float>(i, 0) * b.at<float>(0, j);
      for (int k = 1; k < a.cols; ++k) t = t + a.at<float>(i, k) * b.at<float>(k, j);
      d.at<float>(i, j) = t;
    }
  return d;
}
inline Mat& operator*=(Mat& a, const Mat& b) { a = a * b; return a; }
inline Mat& operator*=(Mat& a, double s) {
  assert(a.depth() == CV_32F && a.isContinuous());
  const size_t n = a.total() * a.channels();
  for (size_t i = 0; i < n; ++i) reinterpret_cast<float*>(a.data)[i] = reinterpret_cast<float*>(a.data)[i] * (float)s;
  return a;
}
inline Mat& operator/=(Mat& a, double s) { return a *= (double)(float)(1.0 / s); }
inline Mat operator*(const Mat& a, double s) {
  Mat d = a.clone();
  d *= s;
  return d;
}
inline void transpose(const Mat& src, Mat& dst) {
  assert(src.type() == CV_32F);
  Mat d(src.cols, src.rows, CV_32F);
  for (int i = 0; i < src.rows; ++i) for (int j = 0; j < src.cols; ++j) d.at<float>(j, i) = src.at<float>(i, j);
  dst = d;
}
inline double invert(const Mat&, Mat&) { throw std::runtime_error("shim: invert (the DNG writer's colour matrices) is not available"); }
inline void dct(const Mat&, Mat&) { throw std::runtime_error("shim: dct (FREQUENCY_DM_FILTER) is not available"); }
inline void idct(const Mat&, Mat&) { throw std::runtime_error("shim: idct (FREQUENCY_DM_FILTER) is not available"); }

}  // namespace cv
