// TEST INFRASTRUCTURE: imread / imwrite of the OpenCV stand-in (ref_shim/opencv2/imgproc.hpp) over host/png_io.hpp.
#include "../../host/png_io.hpp"
#include "opencv2/imgproc.hpp"

namespace cv {
Mat imread(const std::string& path, int flags) {
  pngio::Image im;
  try {
    im = pngio::read(path, flags == IMREAD_UNCHANGED);
  } catch (const std::exception&) {
    return Mat();  // imread returns an empty Mat; imreadExceptionOnFail turns that into the reference's exception
  }
  Mat m(im.h, im.w, CV_MAKETYPE(CV_8U, im.c));
  std::memcpy(m.data, im.px.data(), (size_t)im.w * im.h * im.c);
  return m;
}
bool imwrite(const std::string& path, const Mat& img0, const std::vector<int>&) {
  const Mat img = shim::cont(img0);
  if (img.depth() != CV_8U) shim::unsupported("imwrite of non-8-bit images");
  try {
    if (img.channels() == 1) {
      Mat bgr(img.rows, img.cols, CV_8UC3);
      for (size_t i = 0; i < img.total(); ++i) bgr.data[3 * i] = bgr.data[3 * i + 1] = bgr.data[3 * i + 2] = img.data[i];
      pngio::write(path, bgr.data, img.cols, img.rows, 3);
    } else {
      pngio::write(path, img.data, img.cols, img.rows, img.channels());
    }
  } catch (const std::exception&) {
    return false;
  }
  return true;
}
}  // namespace cv

// gflags' own help flags, which util/SystemUtil.cpp:21-24 declares and touches
namespace fLB {
bool FLAGS_help = false;
bool FLAGS_helpshort = false;
}  // namespace fLB
