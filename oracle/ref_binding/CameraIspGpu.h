// oracle/ref_binding/CameraIspGpu.h — TEST INFRASTRUCTURE. INTEGRATION.md section 3 as a file that compiles: the subclass of
// the reference's CameraIsp (SR/camera_isp/CameraIsp.h) that runs loadImage / getImage on the GPU through include/s360.h
// — what CameraIspPipe does with Halide. `make -C oracle ref_binding_isp` force-includes this header into the reference's
// own Raw2Rgb.cpp (the class name used there redirected to this subclass from outside: nothing of the reference is
// modified or copied), so that the reference's Raw2Rgb program, with its own flags, JSON reader and file I/O, develops
// its images with libs360 — and must write the files the unmodified program writes.
#pragma once
#include <string>
#include <vector>

#include "CameraIsp.h"  // the reference's
#include "VrCamException.h"
#include "s360.h"

namespace surround360 {

class CameraIspGpu : public CameraIsp {              // next to CameraIspPipe, which does the same with Halide
  s360_isp* isp_ = nullptr;
  s360_isp_config cfg_;
  std::vector<uint16_t> raw_; int w_ = 0, h_ = 0;
 public:
  CameraIspGpu(const std::string& json, int outputBpp, int device = 0) : CameraIsp(json, outputBpp) {
    s360_isp_config_defaults(&cfg_);
    cfg_.output_bpp = outputBpp;
    if (s360_isp_config_from_json(json.c_str(), &cfg_) < 0) throw VrCamException(s360_last_error(nullptr));
  }
  void loadImage(const Mat& in) override {           // 16-bit Bayer (Raw2Rgb converts 8-bit inputs first)
    w_ = in.cols; h_ = in.rows;
    raw_.assign((const uint16_t*)in.data, (const uint16_t*)in.data + (size_t)w_ * h_);
  }
  void getImage(Mat& out, const bool swizzle = true) override {
    // what the reference's setters left in the base class (setDemosaicFilter, setResize, disable / enableToneMap,
    // addBlackLevelOffset: CameraIsp.h:884-983) goes into the configuration when the first image is developed
    cfg_.demosaic_filter = (int)demosaicFilter; cfg_.resize = resize; cfg_.disable_tone_curve = disableToneCurve;
    cfg_.black_level[0] = blackLevel.x; cfg_.black_level[1] = blackLevel.y; cfg_.black_level[2] = blackLevel.z;
    if (!isp_ && s360_isp_create(&isp_, 0, &cfg_) < 0) throw VrCamException(s360_last_error(nullptr));
    if (s360_isp_process(isp_, raw_.data(), w_, h_, out.data) < 0) throw VrCamException(s360_last_error(nullptr));
  }                                                  // out: CV_8UC3 / CV_16UC3, B,G,R like swizzle = true
  ~CameraIspGpu() { s360_isp_destroy(isp_); }
};

}  // namespace surround360

// from here on the translation unit's `CameraIsp` is the subclass (Raw2Rgb.cpp:364-456 construct and pass it by that name)
#define CameraIsp CameraIspGpu
