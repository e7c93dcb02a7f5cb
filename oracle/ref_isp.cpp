// TEST INFRASTRUCTURE (oracle/_ref): the reference's own soft ISP — camera_isp/CameraIsp.h with util/JsonUtil.cpp and the
// vendored supereasyjson, compiled from /root/reference where it lies — behind a C entry point. OpenCV is replaced by
// oracle/ref_shim (containers only, see its header). Built by `make -C oracle ref` when /root/reference exists; the
// resulting oracle/_ref/libref_isp.so travels to the GPU box, the reference sources do not.
//
// What runs is Raw2Rgb's non-accelerated path (Raw2Rgb.cpp:441-456): CameraIsp(json, outputBpp), setBitsPerPixel(16),
// setDemosaicFilter, setResize, enable/disableToneMap, addBlackLevelOffset, loadImage(16-bit raw), getImage.
#include <cstdint>
#include <cstring>
#include <string>

#include "CameraIsp.h"
#include "RawConverter.h"

using namespace surround360;

extern "C" {
// RawConverter::convert8Frame / convert12Frame (Unpacker.cpp:141-143): packed sensor bytes -> h x w uint16
int ref_convert_frame(int bits, const void* frame, int w, int h, uint16_t* out) {
  auto v = bits == 8 ? RawConverter::convert8Frame(frame, w, h) : RawConverter::convert12Frame(frame, w, h);
  std::memcpy(out, v->data(), (size_t)w * h * sizeof(uint16_t));
  return 0;
}
// raw16: h x w uint16 (row-major). out: (h / resize) x (w / resize) x 3, uint8 or uint16 by output_bpp, BGR order
// (swizzle = true like Raw2Rgb's runPipeline). Returns 0, or -1 with the message in err.
int ref_isp_run(const char* json_text, const uint16_t* raw16, int w, int h, int output_bpp, int demosaic_filter,
                int resize, int disable_tone_curve, int black_level_offset, void* out, char* err, int err_cap) {
  try {
    cv::Mat input(h, w, CV_16UC1, const_cast<uint16_t*>(raw16));
    const int ow = w / resize, oh = h / resize;
    cv::Mat output(oh, ow, output_bpp == 8 ? CV_8UC3 : CV_16UC3);
    CameraIsp isp(std::string(json_text), output_bpp);
    isp.setBitsPerPixel(16);  // kIspInputBitsPerPixel (Raw2Rgb.cpp)
    isp.setDemosaicFilter(demosaic_filter);
    isp.setResize(resize);
    if (disable_tone_curve) isp.disableToneMap(); else isp.enableToneMap();
    isp.addBlackLevelOffset(black_level_offset);
    isp.loadImage(input);
    isp.getImage(output, true);
    std::memcpy(out, output.data, (size_t)oh * ow * 3 * (output_bpp == 8 ? 1 : 2));
    return 0;
  } catch (const std::exception& e) {
    if (err && err_cap > 0) { std::strncpy(err, e.what(), err_cap - 1); err[err_cap - 1] = 0; }
    return -1;
  }
}
}
