// TEST INFRASTRUCTURE (oracle/_ref): the reference's ACCELERATED ISP as its own programs run it — camera_isp/CameraIspPipe.h (on
// CameraIsp.h, util/JsonUtil.cpp, supereasyjson) compiled from /root/reference where it lies, calling the four functions its
// Halide generator emits; those are ref_ispgen.cpp's: the generator's own source, executed. OpenCV is oracle/ref_shim (containers
// only). Together: oracle/_ref/libref_isppipe.so — what pins oracle/isp_pipe.h and through it the library's pipe kernels.
//
// mode 0 = Raw2Rgb --accelerate (Raw2Rgb.cpp:427-440): CameraIspPipe(json, fast, bpp), setBitsPerPixel(16), enable / disable tone map,
//          addBlackLevelOffset, loadImage(Mat), initPipe, getImage(Mat) (swizzle = true)
// mode 1 = Unpacker (Unpacker.cpp:176-183): CameraIspPipe(json, false, 16), setBitsPerPixel(bits), enableToneMap,
//          loadImage(ptr, w, h), setup, initPipe, getImage(ptr)
#include <cstdint>
#include <cstring>
#include <string>

#include "CameraIspPipe.h"

using namespace surround360;

extern "C" int ref_isp_pipe_run(int mode, const char* json_text, const uint16_t* raw16, int w, int h, int output_bpp, int fast,
                                int disable_tone_curve, int black_level_offset, int bits_per_pixel, void* out, char* err,
                                int err_cap) {
  try {
    const size_t bytes = (size_t)w * h * 3 * (output_bpp == 8 ? 1 : 2);
    std::vector<uint16_t> in(raw16, raw16 + (size_t)w * h);
    if (mode == 0) {
      cv::Mat input(h, w, CV_16UC1, in.data());
      cv::Mat output(h, w, output_bpp == 8 ? CV_8UC3 : CV_16UC3);
      CameraIspPipe isp(std::string(json_text), fast != 0, output_bpp);
      isp.setBitsPerPixel(16);
      if (disable_tone_curve) isp.disableToneMap(); else isp.enableToneMap();
      isp.addBlackLevelOffset(black_level_offset);
      isp.loadImage(input);
      isp.initPipe();
      CameraIsp* base = &isp;  // runPipeline takes a CameraIsp* (Raw2Rgb.cpp:363-375)
      base->getImage(output);
      std::memcpy(out, output.data, bytes);
    } else {
      std::vector<uint8_t> colored(bytes);
      CameraIspPipe isp(std::string(json_text), false, 16);
      isp.setBitsPerPixel(bits_per_pixel);
      isp.enableToneMap();
      isp.loadImage(reinterpret_cast<uint8_t*>(in.data()), w, h);
      isp.setup();
      isp.initPipe();
      isp.getImage(colored.data());
      std::memcpy(out, colored.data(), bytes);
    }
    return 0;
  } catch (const std::exception& e) {
    if (err && err_cap > 0) { std::strncpy(err, e.what(), err_cap - 1); err[err_cap - 1] = 0; }
    return -1;
  }
}
