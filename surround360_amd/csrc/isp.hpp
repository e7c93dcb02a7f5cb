// isp.hpp — internal interface of the soft ISP (isp.cpp: configuration, host-built tables, per-object buffers;
// isp_kernels.hip: the kernels). Reference: surround360_render/source/camera_isp/CameraIsp.h.
#pragma once
#include <vector>

#include "../../include/s360.h"
#include "core.hpp"

namespace s360 {

// kernel parameters derived from the configuration on the host (passed by value)
struct IspDev {
  int resize, demosaic, outputBpp, sharpen;
  unsigned redMask, greenMask;  // bit (i & 1) * 2 + (j & 1) of the Bayer tables (CameraIsp.h:613-660)
  float areaRecip;              // 1 / (65535 * resize^2) (resizeInput)
  float black[3], blackScale[3], wb[3], clampMin[3], clampMax[3];
  float ccm[9];                 // composite CCM x 4095
  float noiseCore, amount[3], maxVal, alpha;
  int stuckR, stuckThr;  // removeStuckPixels: window radius (0: the pass is off or the reference's no-op), threshold
  float stuckDark;
};
struct IspFrameBufs {
  float *plane, *gV, *gH, *green, *img, *lp, *scratch, *state;  // state: 3 * max(w, h) floats (IIR hand-over between passes)
  unsigned char *flag, *stuckAct, *stuckAct0;  // k_isp_stuck: stuckAct / stuckCand one row, stuckAct0 / stuckCand0 the image
  float *stuckCand, *stuckCand0;
  int* stuckDirty;  // one row
  unsigned* stuckCount;  // [0] pixels the pre-pass wants to change, [1] set by k_isp_stuck when that is over `stuckBudget` (pass not run)
  unsigned stuckBudget;  // 0: unbounded
  const float *curveH, *curveV, *lut;
  const unsigned long long* exptab;
};
// the accelerated pipeline's arithmetic (CameraIspPipe / CameraIspGen.cpp; isp_kernels.hip, second half)
struct IspPipeDev {
  int pattern;  // 0 GBRG, 1 RGGB (CameraIspPipe::runPipe knows no other, CameraIspPipe.h:133-141)
  int fast, outputBpp, swizzle;
  float bias[3], invRange[3];  // A (x - B) of black level, white balance and clamp (CameraIspGen.cpp:318-337)
  float ccm[9];                // composite CCM x 4095
  float alpha, noiseCore, amount[3], maxVal;
};
struct IspPipeBufs {
  float *site, *green, *tone, *low, *scratch, *state;  // site: (w + 16) x (h + 16); green: (w + 4) x (h + 4); tone / low / scratch: h x w x 3
  unsigned char* flag;                                  // (w + 12) x (h + 12)
  const float *vigH, *vigV;                             // [w][3] (curve columns 0, 2, 1: CameraIspPipe.h:88-89), [h][3]
  const unsigned short* toneTab;                        // [4096][3]: the tone curve truncated to the output type
  const unsigned long long* exptab;
};
void isp_pipe_launch(hipStream_t st, const IspPipeDev& d, const unsigned short* raw, int w, int h, const IspPipeBufs& B,
                     void* out);
void isp_launch_unpack(hipStream_t st, const unsigned char* frame, int bits, int w, int h, unsigned short* out);
void isp_launch(hipStream_t st, const IspDev& d, const unsigned short* raw, int inW, int inH, const IspFrameBufs& B,
                void* out);

}  // namespace s360

// The object behind the C ABI (s360_isp_*).
struct s360_isp {
  int device = 0;
  hipStream_t st = nullptr;
  s360_isp_config cfg;
  s360::IspDev dev;
  s360::IspPipeDev pipe;  // used instead of `dev` when cfg.pipe != 0
  std::vector<float> ccm, lut;  // host copies of the derived tables (s360_isp_get_tables)
  s360::DevBuf dToneTab;  // pipe: [4096][3] uint16
  s360::DevBuf dStuck;  // k_isp_stuck: n + w floats, w ints, n + w bytes
  s360::DevBuf dStuckCount;        // two words (IspFrameBufs::stuckCount)
  unsigned* hStuckCount = nullptr;  // ... and their page-locked landing place
  s360::DevBuf dGenH, dGenV, dGenTone;  // s360_isp_pipe_generated: the caller's tables
  s360::DevBuf dLut, dExp, dRaw, dPlane, dGV, dGH, dGreen, dFlag, dImg, dLp, dScratch, dState, dOut, dPacked;
  // vignette curves per output size (curveHAtPixel / curveVAtPixel): a rig's side and pole cameras may differ in
  // resolution, so a few sizes are kept instead of rebuilding (and synchronising the upload stream) at every switch
  struct Curves { int w = -1, h = -1; s360::DevBuf h_, v_; };
  Curves curves[4];
  int curveNext = 0;
  // s360_frame_upload_raw runs this object's kernels on a context's upload stream over the buffers above: the object
  // belongs to the first context it is used with (another context's stream would race on dRaw / dPlane / ...)
  unsigned long long boundCtx = 0;  // uid of the context this object feeds (s360_frame_upload_raw); 0 = none yet
  std::string err;
};

namespace s360 {
void isp_config_defaults(s360_isp_config* c);
void isp_config_from_json(const char* text, s360_isp_config* c);
void isp_derive(const s360_isp_config& cfg, IspDev& d, std::vector<float>& lut);
void isp_derive_pipe(const s360_isp_config& cfg, const IspDev& d, IspPipeDev& p);
void isp_vignette_curves(const s360_isp_config& cfg, int w, int h, std::vector<float>& ch, std::vector<float>& cv);
void isp_init(s360_isp* o, int device, const s360_isp_config& cfg);
void isp_process(s360_isp* o, const uint16_t* raw16, int w, int h, void* out);
void isp_process_packed(s360_isp* o, const uint8_t* frame, int bits, int w, int h, void* out);
void isp_pipe_generated(s360_isp* o, const s360_camera_isp_gen_args& a);
void isp_release(s360_isp* o);
void* isp_raw_buffer(s360_isp* o, int inW, int inH);
void* isp_packed_buffer(s360_isp* o, int bits, int inW, int inH);
size_t isp_packed_bytes(int bits, int inW, int inH);
void isp_unpack_on(s360_isp* o, hipStream_t st, int bits, int inW, int inH);
const void* isp_enqueue_on(s360_isp* o, hipStream_t st, unsigned long long ctxUid, int inW, int inH);
}  // namespace s360
