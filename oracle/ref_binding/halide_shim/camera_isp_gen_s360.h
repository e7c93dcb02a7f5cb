// oracle/ref_binding/halide_shim/camera_isp_gen_s360.h — TEST INFRASTRUCTURE. The body shared by the four functions Halide would
// generate (CameraIspGen8 / 16 / Fast8 / Fast16): the generated signature (CameraIspGen.cpp:704-712 + the output buffer; called at
// CameraIspPipe.h:143-175) marshalled into s360_isp_pipe_generated (include/s360.h). One ISP object per calling thread lends
// its stream and buffers (the reference's Unpacker runs one std::async task per camera).
#pragma once
#include <stdexcept>
#include <string>

#include "Halide.h"
#include "s360.h"

namespace s360_halide_shim {
struct Lender {
  s360_isp* isp = nullptr;
  s360_isp* get() {
    if (!isp) {
      s360_isp_config cfg;
      s360_isp_config_defaults(&cfg);
      cfg.pipe = 1;
      cfg.output_bpp = 16;
      if (s360_isp_create(&isp, 0, &cfg) < 0) throw std::runtime_error(s360_last_error(nullptr));
    }
    return isp;
  }
  ~Lender() { if (isp) s360_isp_destroy(isp); }
};
inline int run(int fast, int bpp, buffer_t* in, int width, int height, buffer_t* vigH, buffer_t* vigV, float blR, float blG,
               float blB, float wbR, float wbG, float wbB, float cminR, float cminG, float cminB, float cmaxR, float cmaxG,
               float cmaxB, float shR, float shG, float shB, float support, float noiseCore, buffer_t* ccm, buffer_t* tone,
               bool bgr, int pattern, buffer_t* out) {
  static thread_local Lender lender;
  s360_camera_isp_gen_args a;
  a.input = reinterpret_cast<const uint16_t*>(in->host);
  a.input_stride = in->stride[1];
  a.width = width;
  a.height = height;
  a.vignette_h = reinterpret_cast<const float*>(vigH->host);
  a.vignette_v = reinterpret_cast<const float*>(vigV->host);
  const float bl[3] = {blR, blG, blB}, wb[3] = {wbR, wbG, wbB}, cmin[3] = {cminR, cminG, cminB}, cmax[3] = {cmaxR, cmaxG, cmaxB},
              sh[3] = {shR, shG, shB};
  for (int k = 0; k < 3; ++k) {
    a.black_level[k] = bl[k]; a.white_balance_gain[k] = wb[k]; a.clamp_min[k] = cmin[k]; a.clamp_max[k] = cmax[k]; a.sharpening[k] = sh[k];
  }
  a.sharpening_support = support;
  a.noise_core = noiseCore;
  a.ccm = reinterpret_cast<const float*>(ccm->host);
  a.tone_table = tone->host;
  a.bgr = bgr ? 1 : 0;
  a.bayer_pattern = pattern;
  a.fast = fast;
  a.output_bpp = bpp;
  a.output = out->host;
  if (s360_isp_pipe_generated(lender.get(), &a) < 0) throw std::runtime_error(std::string("s360_isp_pipe_generated: ") + s360_last_error(nullptr));
  return 0;  // (a generated pipeline returns 0 on success)
}
}  // namespace s360_halide_shim

#define S360_HALIDE_GENERATED(NAME, FAST, BPP)                                                                              \
  inline int NAME(buffer_t* in, int width, int height, buffer_t* vigH, buffer_t* vigV, float blR, float blG, float blB,     \
                  float wbR, float wbG, float wbB, float cminR, float cminG, float cminB, float cmaxR, float cmaxG,         \
                  float cmaxB, float shR, float shG, float shB, float support, float noiseCore, buffer_t* ccm,             \
                  buffer_t* tone, bool bgr, int pattern, buffer_t* out) {                                                   \
    return s360_halide_shim::run(FAST, BPP, in, width, height, vigH, vigV, blR, blG, blB, wbR, wbG, wbB, cminR, cminG,      \
                                 cminB, cmaxR, cmaxG, cmaxB, shR, shG, shB, support, noiseCore, ccm, tone, bgr, pattern, out); \
  }
