// TEST INFRASTRUCTURE: imread / imwrite of the OpenCV stand-in (ref_shim/opencv2/imgproc.hpp) over host/png_io.hpp.
#include "../../host/png_io.hpp"
#include "opencv2/imgproc.hpp"

namespace cv {
Mat imread(const std::string& path, int flags) {
  if (flags == (CV_LOAD_IMAGE_GRAYSCALE | CV_LOAD_IMAGE_ANYDEPTH)) {  // raw Bayer frames: 8- or 16-bit greyscale, depth kept
    try {
      int w = 0, h = 0, depth = 0;
      const std::vector<uint16_t> px = pngio::read_gray(path, &w, &h, &depth);
      Mat m(h, w, depth == 16 ? CV_16UC1 : CV_8UC1);
      if (depth == 16) std::memcpy(m.data, px.data(), (size_t)w * h * 2);
      else for (size_t i = 0; i < (size_t)w * h; ++i) m.data[i] = (uchar)px[i];
      return m;
    } catch (const std::exception&) {
      return Mat();
    }
  }
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
  if (img.type() == CV_16UC3) {
    try {
      pngio::write16(path, reinterpret_cast<const uint16_t*>(img.data), img.cols, img.rows);
    } catch (const std::exception&) {
      return false;
    }
    return true;
  }
  if (img.type() == CV_16UC1 && path.size() > 4 && path.find(".tif", path.size() - 5) != std::string::npos) {
    // the raw frames Unpacker keeps (Unpacker.cpp:145-152): a baseline little-endian TIFF, one strip, no compression
    try {
      pngio::OutFile f(path);
      const uint32_t w = (uint32_t)img.cols, h = (uint32_t)img.rows, nbytes = w * h * 2, ifd = 8 + nbytes;
      const uint8_t hdr[8] = {'I', 'I', 42, 0, (uint8_t)ifd, (uint8_t)(ifd >> 8), (uint8_t)(ifd >> 16), (uint8_t)(ifd >> 24)};
      f.put(hdr, 8);
      f.put(img.data, nbytes);
      const uint32_t tags[9][4] = {{256, 4, 1, w}, {257, 4, 1, h}, {258, 3, 1, 16}, {259, 3, 1, 1}, {262, 3, 1, 1},
                                   {273, 4, 1, 8}, {277, 3, 1, 1}, {278, 4, 1, h}, {279, 4, 1, nbytes}};
      std::vector<uint8_t> d(2 + 12 * 9 + 4, 0);
      d[0] = 9;
      for (int i = 0; i < 9; ++i) {
        uint8_t* e = &d[2 + 12 * i];
        const uint16_t id = (uint16_t)tags[i][0], ty = (uint16_t)tags[i][1];
        std::memcpy(e, &id, 2); std::memcpy(e + 2, &ty, 2); std::memcpy(e + 4, &tags[i][2], 4);
        if (ty == 3) { const uint16_t v = (uint16_t)tags[i][3]; std::memcpy(e + 8, &v, 2); } else std::memcpy(e + 8, &tags[i][3], 4);
      }
      f.put(d.data(), d.size());
      f.close();
    } catch (const std::exception&) {
      return false;
    }
    return true;
  }
  if (img.depth() != CV_8U) shim::unsupported("imwrite of this depth");
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

// glog's command-line flags that scripts/batch_process_video.py passes to the program (accepted, without effect here)
#include "gflags/gflags.h"
DEFINE_int32(logbuflevel, 0, "");
DEFINE_string(log_dir, "", "");
DEFINE_int32(stderrthreshold, 2, "");
DEFINE_int32(v, 0, "");

// gflags' own help flags, which util/SystemUtil.cpp:21-24 declares and touches
namespace fLB {
bool FLAGS_help = false;
bool FLAGS_helpshort = false;
}  // namespace fLB
