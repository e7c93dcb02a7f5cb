// isp.cpp — host side of the soft ISP: the CameraIsp constructor's reading of a configuration, CameraIsp::setup /
// buildToneCurveLut (composite CCM, tone curve) and the vignette curves on the CPU like the reference, per-object device
// buffers, one frame = upload, 5-8 kernels, download. Reference: surround360_render/source/camera_isp/CameraIsp.h.
#include "isp.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>

#include "json_mini.hpp"

namespace s360 {

namespace {
inline float clampf(float x, float a, float b) { return x < a ? a : x > b ? b : x; }  // MathUtil.h:38-41
inline float lerpf(float x0, float x1, float a) { return x0 * (1.0f - a) + x1 * a; }   // MathUtil.h:58-61
// BezierCurve<float, Vec3f>::operator()(i, j, t) on one channel (MathUtil.h:205-213): De Casteljau by recursion
float bezier(const float (*p)[3], int ch, int i, int j, float t) {
  if (i == j) return p[i][ch];
  return lerpf(bezier(p, ch, i, j - 1, t), bezier(p, ch, i + 1, j, t), t);
}
// tone curve pieces (CameraIsp.h:361-388)
float bezier4(float a, float b, float c, float d, float t) {
  return lerpf(lerpf(lerpf(a, b, t), lerpf(b, c, t), t), lerpf(lerpf(b, c, t), lerpf(c, d, t), t), t);
}
float high_key(float boost, float x) {
  const float a = 0.5f, b = clampf(0.6666f, 0.0f, 1.0f), c = clampf(0.8333f + boost, 0.0f, 1.0f), d = 1.0f;
  return x > 0.5f ? bezier4(a, b, c, d, (x - 0.5f) * 2.0f) : 0;
}
float low_key(float boost, float x) {
  const float a = 0.0f, b = clampf(0.1666f + boost, 0.0f, 1.0f), c = clampf(0.3333f, 0.0f, 1.0f), d = 0.5f;
  return x <= 0.5f ? bezier4(a, b, c, d, x * 2.0f) : 0;
}
// 3x3 CV_32F product: float products summed left to right (what cv::gemm's small-matrix path does)
void mul33(const float* a, const float* b, float* d) {
  float t[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float s = a[i * 3] * b[j];
      s = s + a[i * 3 + 1] * b[3 + j];
      s = s + a[i * 3 + 2] * b[6 + j];
      t[i * 3 + j] = s;
    }
  std::memcpy(d, t, sizeof t);
}
// a JSON number, or an error like the reference's ToDouble() / ToInt() on anything else (CameraIsp.h:464-607)
double number_of(const JV& v, const char* key) {
  if (v.t != JV::NUM) throw Error(S360_ERR_IO, std::string("isp json: '") + key + "' holds a value that is not a number");
  return v.num;
}
// ... as an int (the reference's ToInt()); values no int holds are an error here, not undefined behaviour
int int_of(const JV& v, const char* key) {
  const double d = number_of(v, key);
  if (!(d > -1e9 && d < 1e9)) throw Error(S360_ERR_IO, std::string("isp json: '") + key + "' is out of range");
  return (int)d;
}
void vec3(const JV& o, const char* key, float* dst) {
  const JV* a = o.get(key);
  if (!a) return;
  if (a->t != JV::ARR || a->arr.size() != 3) throw Error(S360_ERR_IO, std::string("isp json: '") + key + "' is not a 3-vector");
  for (int k = 0; k < 3; ++k) dst[k] = (float)number_of(a->arr[k], key);
}
void num(const JV& o, const char* key, float* dst) {
  if (const JV* a = o.get(key)) *dst = (float)number_of(*a, key);
}
void coord_list(const JV& o, const char* key, float (*dst)[3], int32_t* count) {
  const JV* a = o.get(key);
  if (!a) return;
  if (a->t != JV::ARR || a->arr.empty() || a->arr.size() > S360_ISP_MAX_CURVE_POINTS)
    throw Error(S360_ERR_IO, std::string("isp json: '") + key + "' needs 1.." + std::to_string(S360_ISP_MAX_CURVE_POINTS) + " points");
  *count = (int32_t)a->arr.size();
  for (size_t i = 0; i < a->arr.size(); ++i) {
    const JV& p = a->arr[i];
    if (p.t != JV::ARR || p.arr.size() != 3) throw Error(S360_ERR_IO, std::string("isp json: '") + key + "' holds a non-3-vector");
    for (int k = 0; k < 3; ++k) dst[i][k] = (float)number_of(p.arr[k], key);
  }
}
}  // namespace

void isp_config_defaults(s360_isp_config* c) {  // CameraIsp.h:440-462
  std::memset(c, 0, sizeof *c);
  for (int k = 0; k < 3; ++k) c->clamp_max[k] = c->white_balance_gain[k] = c->gamma[k] = 1.0f;
  c->ccm[0] = c->ccm[4] = c->ccm[8] = 1.0f;
  c->saturation = c->contrast = 1.0f;
  c->sharpening_support = 10.0f / 2048.0f;
  c->noise_core = 1000.0f;
  c->n_vignette_h = c->n_vignette_v = 1;
  for (int k = 0; k < 3; ++k) c->vignette_roll_off_h[0][k] = c->vignette_roll_off_v[0][k] = 1.0f;
  c->bayer_pattern = 2;  // "GBRG"
  c->output_bpp = 8;     // Raw2Rgb.cpp flag defaults
  c->demosaic_filter = 2;
  c->resize = 1;
}

void isp_config_from_json(const char* text, s360_isp_config* c) {  // CameraIsp.h:464-607
  const s360_isp_config flags = *c;
  isp_config_defaults(c);
  c->output_bpp = flags.output_bpp;
  c->demosaic_filter = flags.demosaic_filter;
  c->resize = flags.resize;
  c->disable_tone_curve = flags.disable_tone_curve;
  c->black_level_offset = flags.black_level_offset;
  c->pipe = flags.pipe;
  const std::string s(text);
  JP p{s.data(), s.data() + s.size()};
  const JV root = p.document();
  const JV* isp = root.get("CameraIsp");
  if (!isp || isp->t != JV::OBJ) return;  // "Missing CameraIsp: using defaults"
  vec3(*isp, "blackLevel", c->black_level);
  vec3(*isp, "clampMin", c->clamp_min);
  vec3(*isp, "clampMax", c->clamp_max);
  vec3(*isp, "whiteBalanceGain", c->white_balance_gain);
  vec3(*isp, "gamma", c->gamma);
  vec3(*isp, "lowKeyBoost", c->low_key_boost);
  vec3(*isp, "highKeyBoost", c->high_key_boost);
  vec3(*isp, "sharpening", c->sharpening);
  num(*isp, "saturation", &c->saturation);
  num(*isp, "contrast", &c->contrast);
  num(*isp, "sharpeningSupport", &c->sharpening_support);
  num(*isp, "noiseCore", &c->noise_core);
  coord_list(*isp, "vignetteRollOffH", c->vignette_roll_off_h, &c->n_vignette_h);
  coord_list(*isp, "vignetteRollOffV", c->vignette_roll_off_v, &c->n_vignette_v);
  if (const JV* m = isp->get("ccm")) {
    if (m->t != JV::ARR || m->arr.size() != 3) throw Error(S360_ERR_IO, "isp json: 'ccm' is not 3x3");
    for (int i = 0; i < 3; ++i) {
      if (m->arr[i].t != JV::ARR || m->arr[i].arr.size() != 3) throw Error(S360_ERR_IO, "isp json: 'ccm' is not 3x3");
      for (int j = 0; j < 3; ++j) c->ccm[i * 3 + j] = (float)number_of(m->arr[i].arr[j], "ccm");
    }
  }
  if (const JV* r = isp->get("stuckPixelRadius")) c->stuck_pixel_radius = 2 * int_of(*r, "stuckPixelRadius");
  if (const JV* r = isp->get("stuckPixelThreshold")) c->stuck_pixel_threshold = int_of(*r, "stuckPixelThreshold");
  if (const JV* r = isp->get("stuckPixelDarknessThreshold"))
    c->stuck_pixel_darkness_threshold = (float)number_of(*r, "stuckPixelDarknessThreshold");
  if (const JV* b = isp->get("bayerPattern")) {  // setup(): the first of these names the string contains
    if (b->t != JV::STR) throw Error(S360_ERR_IO, "isp json: 'bayerPattern' is not a string");
    static const char* names[4] = {"RGGB", "GRBG", "GBRG", "BGGR"};
    int found = -1;
    for (int i = 0; i < 4 && found < 0; ++i)
      if (b->str.find(names[i]) != std::string::npos) found = i;
    if (found < 0) throw Error(S360_ERR_IO, "isp json: unknown bayerPattern '" + b->str + "'");
    c->bayer_pattern = found;
  }
}

// Everything the kernels need that depends on the configuration only — host arithmetic like the reference's
// (CameraIsp::setup, buildToneCurveLut, addBlackLevelOffset). No device involved.
void isp_derive(const s360_isp_config& cfg, IspDev& d, std::vector<float>& lut) {
  if (cfg.output_bpp != 8 && cfg.output_bpp != 16) throw Error(S360_ERR_INVALID_ARG, "output_bpp must be 8 or 16");
  if (cfg.pipe < 0 || cfg.pipe > 2) throw Error(S360_ERR_INVALID_ARG, "pipe must be 0 (CameraIsp), 1 (CameraIspPipe) or 2 (CameraIspPipe, fast)");
  // CameraIspPipe has neither a demosaic filter choice nor stuck-pixel removal, and no resize: Raw2Rgb --accelerate sizes its
  // output by --resize and lets the pipeline write the full frame into it (Raw2Rgb.cpp:421-437) — refused here
  if (cfg.pipe && cfg.resize != 1) throw Error(S360_ERR_INVALID_ARG, "the accelerated pipeline (CameraIspPipe) has no resize");
  if (!cfg.pipe && cfg.demosaic_filter == 1) throw Error(S360_ERR_INVALID_ARG, "demosaic_filter 1 (DCT) is not supported");
  if (cfg.demosaic_filter < 0 || cfg.demosaic_filter > 2) throw Error(S360_ERR_INVALID_ARG, "expecting Demosaic filter in [0,2]");
  if (cfg.resize != 1 && cfg.resize != 2 && cfg.resize != 4 && cfg.resize != 8)
    throw Error(S360_ERR_INVALID_ARG, "expecting a resize value of 1, 2, 4, or 8. got " + std::to_string(cfg.resize));
  if (cfg.stuck_pixel_radius > 7 && !cfg.pipe)
    throw Error(S360_ERR_INVALID_ARG, "stuckPixelRadius above 3 (a window wider than 15 x 15) is not supported");
  if (cfg.bayer_pattern < 0 || cfg.bayer_pattern > 3) throw Error(S360_ERR_INVALID_ARG, "bayer_pattern must be 0..3");
  if (cfg.n_vignette_h < 1 || cfg.n_vignette_h > S360_ISP_MAX_CURVE_POINTS || cfg.n_vignette_v < 1 ||
      cfg.n_vignette_v > S360_ISP_MAX_CURVE_POINTS)
    throw Error(S360_ERR_INVALID_ARG, "vignette curves need 1..16 control points");
  std::memset(&d, 0, sizeof d);
  d.resize = cfg.resize;
  d.demosaic = cfg.demosaic_filter;
  d.outputBpp = cfg.output_bpp;
  static const unsigned R[4] = {1u << 0, 1u << 1, 1u << 2, 1u << 3};               // red site: bit i * 2 + j
  static const unsigned G[4] = {(1u << 1) | (1u << 2), (1u << 0) | (1u << 3), (1u << 0) | (1u << 3), (1u << 1) | (1u << 2)};
  d.redMask = R[cfg.bayer_pattern];
  d.greenMask = G[cfg.bayer_pattern];
  const int maxPixelValue = 65535;  // 16-bit input (loadImage)
  d.areaRecip = 1.0f / (maxPixelValue * float(cfg.resize * cfg.resize));
  for (int k = 0; k < 3; ++k) {
    const float bl = cfg.black_level[k] + float(cfg.black_level_offset);  // addBlackLevelOffset
    d.black[k] = bl / float(maxPixelValue);
    d.blackScale[k] = 1.0f / (1.0f - d.black[k]);
    d.wb[k] = cfg.white_balance_gain[k];
    d.clampMin[k] = cfg.clamp_min[k];
    d.clampMax[k] = cfg.clamp_max[k];
    d.amount[k] = 1.0f + cfg.sharpening[k];
  }
  d.sharpen = cfg.sharpening[0] != 0.0 && cfg.sharpening[1] != 0.0 && cfg.sharpening[2] != 0.0;
  // removeStuckPixels' loop over the sorted region, `k <= region.size() - stuckPixelThreshold` in size_t arithmetic
  // (CameraIsp.h:1090-1092), is false from the start for 2 <= threshold <= region.size(): the pass then changes nothing.
  // The smallest region is a red / blue pixel's — the same-colour sites of a (2 R + 1)^2 window, R = stuck_pixel_radius =
  // 2 x the JSON value: (R + 1)^2 — so thresholds of 2 .. (R + 1)^2 (the shipped 5 included) need no kernel at all; any
  // other threshold makes the pass a serial in-place median filter of the dark regions, which k_isp_stuck runs.
  d.stuckR = 0;
  if (cfg.stuck_pixel_radius > 0) {
    const long long nmin = (long long)(cfg.stuck_pixel_radius + 1) * (cfg.stuck_pixel_radius + 1);
    if (cfg.stuck_pixel_threshold < 2 || cfg.stuck_pixel_threshold > nmin) d.stuckR = cfg.stuck_pixel_radius;
  }
  d.stuckThr = cfg.stuck_pixel_threshold;
  d.stuckDark = cfg.stuck_pixel_darkness_threshold;
  d.noiseCore = cfg.noise_core;
  d.maxVal = (1 << cfg.output_bpp) - 1.0f;
  d.alpha = powf(cfg.sharpening_support, 1.0f / 4.0f);
  // CameraIsp::setup (:662-686): satMat = yuv2rgb * diag(1, sat, sat) * rgb2yuv; compositeCCM = ccm^T * satMat * 4095
  static const float rgb2yuv[9] = {0.299f, 0.587f, 0.114f, -0.14713f, -0.28886f, 0.436f, 0.615f, -0.51499f, -0.10001f};
  static const float yuv2rgb[9] = {1.0f, 0.0f, 1.13983f, 1.0f, -0.39465f, -0.58060f, 1.0f, 2.03211f, 0.0f};
  const float sat[9] = {1.0f, 0, 0, 0, cfg.saturation, 0, 0, 0, cfg.saturation};
  float tmp[9], satMat[9], ccmT[9], comp[9];
  mul33(yuv2rgb, sat, tmp);
  mul33(tmp, rgb2yuv, satMat);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) ccmT[j * 3 + i] = cfg.ccm[i * 3 + j];
  mul33(ccmT, satMat, comp);
  for (int i = 0; i < 9; ++i) comp[i] = comp[i] * 4095.0f;
  std::memcpy(d.ccm, comp, sizeof comp);
  // buildToneCurveLut (:390-425)
  lut.resize(4096 * 3);
  const float range = float((1 << cfg.output_bpp) - 1);
  const float dx = 1.0f / 4095.0f;
  const float angle = M_PI * 0.25f * cfg.contrast;
  const float slope = tanf(angle);
  const float bias = 0.5f * (1.0f - slope);
  for (int i = 0; i < 4096; ++i) {
    const float x = dx * i;
    float* e = &lut[(size_t)i * 3];
    if (cfg.disable_tone_curve) {
      e[0] = e[1] = e[2] = x * range;
    } else {
      for (int k = 0; k < 3; ++k) {
        float v = powf(x, cfg.gamma[k]);
        v = low_key(cfg.low_key_boost[k], v) + high_key(cfg.high_key_boost[k], v);
        e[k] = clampf((slope * v + bias) * range, 0.0f, range);
      }
    }
  }
}

static void pipe_enqueue(s360_isp* o, hipStream_t st, const IspPipeDev& d, int w, int h, const float* vigH, const float* vigV,
                         const unsigned short* toneTab);
// The scalar preamble of the generated pipeline (CameraIspGen.cpp:318-337, 569-571, 605) from the parameters it is called with
static void pipe_preamble(IspPipeDev& p, const float* black, const float* wb, const float* cmin, const float* cmax,
                          const float* sharpening, float support, float noiseCore, const float* ccm9) {
  const float maxRaw = float((1 << 16) - 1);
  for (int k = 0; k < 3; ++k) {
    const float minRaw = black[k];
    const float rangeRaw = maxRaw - minRaw;
    p.bias[k] = minRaw + cmin[k] * rangeRaw / wb[k];
    p.invRange[k] = wb[k] / (rangeRaw * (cmax[k] - cmin[k]));
    p.amount[k] = 1.0f + sharpening[k];
  }
  std::memcpy(p.ccm, ccm9, sizeof p.ccm);
  p.alpha = powf(support, 1.0f / 4.0f);
  p.noiseCore = noiseCore;
  p.maxVal = float((1 << p.outputBpp) - 1);
}
// What CameraIspPipe hands its generated pipeline (CameraIspPipe.h:131-176) and what the generator's preamble derives from
// it (CameraIspGen.cpp:318-337, 569-571, 605): float arithmetic in the written order.
void isp_derive_pipe(const s360_isp_config& cfg, const IspDev& d, IspPipeDev& p) {
  std::memset(&p, 0, sizeof p);
  p.pattern = cfg.bayer_pattern == 0 ? 1 : 0;  // "GBRG" -> 0, else "RGGB" -> 1, else 0 (runPipe)
  p.fast = cfg.pipe == 2;
  p.outputBpp = cfg.output_bpp;
  p.swizzle = 1;  // getImage(.., swizzle = true)
  float black[3];
  for (int k = 0; k < 3; ++k) black[k] = cfg.black_level[k] + float(cfg.black_level_offset);  // addBlackLevelOffset
  pipe_preamble(p, black, cfg.white_balance_gain, cfg.clamp_min, cfg.clamp_max, cfg.sharpening, cfg.sharpening_support,
                cfg.noise_core, d.ccm);
}

// An object that feeds a live context (s360_frame_upload_raw / _packed: its kernels run on THAT context's upload stream over this
// object's buffers) cannot develop images on its own stream at the same time: the two streams would race on dRaw / dOut.
static void refuse_while_feeding(const s360_isp* o, const char* what) {
  if (o->boundCtx && context_alive(o->boundCtx))
    throw Error(S360_ERR_STATE, std::string(what) + ": this ISP object feeds a context (s360_frame_upload_raw); use another object, or destroy that context first");
}

// The generated functions themselves (s360_isp_pipe_generated): every parameter comes from the caller, the object lends its
// stream and buffers.
void isp_pipe_generated(s360_isp* o, const s360_camera_isp_gen_args& a) {
  if (!o->cfg.pipe) throw Error(S360_ERR_STATE, "s360_isp_pipe_generated needs an ISP object created with pipe = 1 or 2");
  if (!a.input || !a.output || !a.vignette_h || !a.vignette_v || !a.ccm || !a.tone_table) throw Error(S360_ERR_INVALID_ARG, "null argument");
  if (a.output_bpp != 8 && a.output_bpp != 16) throw Error(S360_ERR_INVALID_ARG, "output_bpp must be 8 or 16");
  if (a.width < 16 || a.height < 16) throw Error(S360_ERR_INVALID_ARG, "image too small for the accelerated pipeline (needs at least 16x16)");
  if (a.input_stride < a.width) throw Error(S360_ERR_INVALID_ARG, "input_stride is smaller than width");
  if (a.bayer_pattern != 0 && a.bayer_pattern != 1) throw Error(S360_ERR_INVALID_ARG, "bayer_pattern is 0 (GBRG) or 1 (RGGB) at this level");
  refuse_while_feeding(o, "s360_isp_pipe_generated");
  S360_HIP(hipSetDevice(o->device));
  const int w = a.width, h = a.height;
  IspPipeDev d;
  std::memset(&d, 0, sizeof d);
  d.pattern = a.bayer_pattern;
  d.fast = a.fast != 0;
  d.outputBpp = a.output_bpp;
  d.swizzle = a.bgr != 0;
  pipe_preamble(d, a.black_level, a.white_balance_gain, a.clamp_min, a.clamp_max, a.sharpening, a.sharpening_support, a.noise_core, a.ccm);
  std::vector<unsigned short> tt((size_t)4096 * 3);
  for (size_t i = 0; i < tt.size(); ++i)
    tt[i] = a.output_bpp == 8 ? static_cast<const unsigned char*>(a.tone_table)[i] : static_cast<const unsigned short*>(a.tone_table)[i];
  o->dRaw.ensure((size_t)w * h * sizeof(uint16_t));
  o->dGenH.ensure((size_t)w * 3 * sizeof(float));
  o->dGenV.ensure((size_t)h * 3 * sizeof(float));
  o->dGenTone.ensure(tt.size() * sizeof(unsigned short));
  if (a.input_stride == w) {
    S360_HIP(hipMemcpyAsync(o->dRaw.p, a.input, (size_t)w * h * 2, hipMemcpyHostToDevice, o->st));
  } else {  // (a padded buffer_t: row by row)
    for (int y = 0; y < h; ++y)
      S360_HIP(hipMemcpyAsync(o->dRaw.as<uint16_t>() + (size_t)y * w, a.input + (size_t)y * a.input_stride, (size_t)w * 2,
                              hipMemcpyHostToDevice, o->st));
  }
  S360_HIP(hipMemcpyAsync(o->dGenH.p, a.vignette_h, (size_t)w * 3 * sizeof(float), hipMemcpyHostToDevice, o->st));
  S360_HIP(hipMemcpyAsync(o->dGenV.p, a.vignette_v, (size_t)h * 3 * sizeof(float), hipMemcpyHostToDevice, o->st));
  S360_HIP(hipMemcpyAsync(o->dGenTone.p, tt.data(), tt.size() * sizeof(unsigned short), hipMemcpyHostToDevice, o->st));
  pipe_enqueue(o, o->st, d, w, h, o->dGenH.as<float>(), o->dGenV.as<float>(), o->dGenTone.as<unsigned short>());
  S360_HIP(hipMemcpyAsync(a.output, o->dOut.p, (size_t)w * h * 3 * (a.output_bpp == 8 ? 1 : 2), hipMemcpyDeviceToHost, o->st));
  S360_HIP(hipStreamSynchronize(o->st));
}

void isp_vignette_curves(const s360_isp_config& cfg, int w, int h, std::vector<float>& ch, std::vector<float>& cv) {
  const int maxDimension = std::max(w, h);  // curveHAtPixel / curveVAtPixel (CameraIsp.h:709-715)
  ch.resize((size_t)w * 3);
  cv.resize((size_t)h * 3);
  for (int j = 0; j < w; ++j)
    for (int k = 0; k < 3; ++k) ch[(size_t)j * 3 + k] = bezier(cfg.vignette_roll_off_h, k, 0, cfg.n_vignette_h - 1, float(j) / float(maxDimension));
  for (int i = 0; i < h; ++i)
    for (int k = 0; k < 3; ++k) cv[(size_t)i * 3 + k] = bezier(cfg.vignette_roll_off_v, k, 0, cfg.n_vignette_v - 1, float(i) / float(maxDimension));
}

void isp_init(s360_isp* o, int device, const s360_isp_config& cfg) {
  isp_derive(cfg, o->dev, o->lut);
  o->ccm.assign(o->dev.ccm, o->dev.ccm + 9);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw Error(S360_ERR_NO_DEVICE, "no HIP device available (libs360 has no CPU path)");
  if (device < 0 || device >= ndev) throw Error(S360_ERR_INVALID_ARG, "device index out of range");
  o->device = device;
  o->cfg = cfg;
  S360_HIP(hipSetDevice(device));
  S360_HIP(hipStreamCreateWithFlags(&o->st, hipStreamNonBlocking));
  o->dLut.ensure(o->lut.size() * sizeof(float));
  S360_HIP(hipMemcpyAsync(o->dLut.p, o->lut.data(), o->lut.size() * sizeof(float), hipMemcpyHostToDevice, o->st));
  // 2^(i/32) bit patterns minus i << 47 (the table of glibc's expf, see isp_kernels.hip)
  unsigned long long tab[32];
  for (int i = 0; i < 32; ++i) {
    const double v = exp2((double)i / 32);
    unsigned long long u;
    std::memcpy(&u, &v, 8);
    tab[i] = u - ((unsigned long long)i << 47);
  }
  if (cfg.pipe) {
    isp_derive_pipe(cfg, o->dev, o->pipe);
    std::vector<unsigned short> tt(o->lut.size());  // `const int r = toneCurveLut[i][0]` into the 8- / 16-bit table (CameraIspPipe.h:67-81)
    for (size_t i = 0; i < tt.size(); ++i) tt[i] = (unsigned short)(int)o->lut[i];
    o->dToneTab.ensure(tt.size() * sizeof(unsigned short));
    S360_HIP(hipMemcpyAsync(o->dToneTab.p, tt.data(), tt.size() * sizeof(unsigned short), hipMemcpyHostToDevice, o->st));
    S360_HIP(hipStreamSynchronize(o->st));  // (tt goes out of scope)
  }
  o->dExp.ensure(sizeof tab);
  S360_HIP(hipMemcpyAsync(o->dExp.p, tab, sizeof tab, hipMemcpyHostToDevice, o->st));
  S360_HIP(hipStreamSynchronize(o->st));
}

static void isp_run_uploaded(s360_isp* o, int inW, int inH, void* out);
// pixels k_isp_stuck may rewrite per frame (S360_ISP_STUCK_BUDGET; 0 = no bound). Default 262 144: about a second of the serial walk.
static unsigned stuck_pixel_budget() {
  static const unsigned v = [] {
    const char* e = std::getenv("S360_ISP_STUCK_BUDGET");
    if (!e || !e[0]) return 262144u;
    const long long n = std::atoll(e);
    return n <= 0 ? 0u : (unsigned)std::min<long long>(n, 0xffffffffll);
  }();
  return v;
}

static void isp_enqueue(s360_isp* o, hipStream_t st, int inW, int inH);
void isp_process(s360_isp* o, const uint16_t* raw16, int inW, int inH, void* out) {
  refuse_while_feeding(o, "s360_isp_process");
  S360_HIP(hipSetDevice(o->device));
  o->dRaw.ensure((size_t)inW * inH * sizeof(uint16_t));
  S360_HIP(hipMemcpyAsync(o->dRaw.p, raw16, (size_t)inW * inH * sizeof(uint16_t), hipMemcpyHostToDevice, o->st));
  isp_run_uploaded(o, inW, inH, out);
}
// Unpacker's per-frame work (Unpacker.cpp:136-143, 169-183): the sensor's packed bytes are widened on the device
void isp_process_packed(s360_isp* o, const uint8_t* frame, int bits, int inW, int inH, void* out) {
  if (bits != 8 && bits != 12) throw Error(S360_ERR_INVALID_ARG, "packed frames are 8 or 12 bits per pixel");
  if (bits == 12 && (inW & 1)) throw Error(S360_ERR_INVALID_ARG, "12-bit packed frames need an even width");
  refuse_while_feeding(o, "s360_isp_process_packed");
  S360_HIP(hipSetDevice(o->device));
  const size_t bytes = bits == 8 ? (size_t)inW * inH : (size_t)inH * (3 * (size_t)inW / 2);
  o->dPacked.ensure(bytes);
  o->dRaw.ensure((size_t)inW * inH * sizeof(uint16_t));
  S360_HIP(hipMemcpyAsync(o->dPacked.p, frame, bytes, hipMemcpyHostToDevice, o->st));
  isp_launch_unpack(o->st, o->dPacked.as<unsigned char>(), bits, inW, inH, o->dRaw.as<unsigned short>());
  isp_run_uploaded(o, inW, inH, out);
}

// The accelerated pipeline on the frame in dRaw: buffers, launch; the result is left in dOut.
static void pipe_enqueue(s360_isp* o, hipStream_t st, const IspPipeDev& d, int w, int h, const float* vigH, const float* vigV,
                         const unsigned short* toneTab) {
  const size_t n = (size_t)w * h;
  o->dOut.ensure(n * 3 * (d.outputBpp == 8 ? 1 : 2));
  o->dPlane.ensure((size_t)(w + 16) * (h + 16) * sizeof(float));
  o->dImg.ensure(n * 3 * sizeof(float));
  if (!d.fast) {
    o->dFlag.ensure((size_t)(w + 12) * (h + 12));
    o->dGreen.ensure((size_t)(w + 4) * (h + 4) * sizeof(float));
    o->dLp.ensure(n * 3 * sizeof(float));
    o->dScratch.ensure(n * 3 * sizeof(float));
    o->dState.ensure((size_t)3 * std::max(w, h) * sizeof(float));
  }
  IspPipeBufs P;
  P.site = o->dPlane.as<float>();
  P.green = o->dGreen.as<float>();
  P.tone = o->dImg.as<float>();
  P.low = o->dLp.as<float>();
  P.scratch = o->dScratch.as<float>();
  P.state = o->dState.as<float>();
  P.flag = o->dFlag.as<unsigned char>();
  P.vigH = vigH;
  P.vigV = vigV;
  P.toneTab = toneTab;
  P.exptab = o->dExp.as<unsigned long long>();
  isp_pipe_launch(st, d, o->dRaw.as<unsigned short>(), w, h, P, o->dOut.p);
  S360_HIP(hipGetLastError());
}

// Enqueues the ISP of the frame already in dRaw on `st`; the result (B,G,R, 8 or 16 bit) is left in dOut.
static void isp_enqueue(s360_isp* o, hipStream_t st, int inW, int inH) {
  const s360_isp_config& cfg = o->cfg;
  const int w = inW / cfg.resize, h = inH / cfg.resize;
  // the 9x9 homogeneity window and the reflected +-2 taps index up to 4 pixels past an edge (the reference reads out
  // of bounds below that size); the pipeline mirrors 8 pixels beyond every edge
  if (w < 8 || h < 8) throw Error(S360_ERR_INVALID_ARG, "image too small for the ISP (needs at least 8x8 after resize)");
  if (o->dev.stuckR > 0 && !cfg.pipe && w > 65536) throw Error(S360_ERR_INVALID_ARG, "stuck-pixel removal: images wider than 65536 are not supported");
  if (cfg.pipe && (w < 16 || h < 16)) throw Error(S360_ERR_INVALID_ARG, "image too small for the accelerated pipeline (needs at least 16x16)");
  const size_t n = (size_t)w * h;
  s360_isp::Curves* cur = nullptr;
  for (auto& cv : o->curves)
    if (cv.w == w && cv.h == h) cur = &cv;
  if (!cur) {  // vignette curves at every column / row (curveHAtPixel / curveVAtPixel), built once per size
    cur = &o->curves[o->curveNext];
    o->curveNext = (o->curveNext + 1) % 4;
    std::vector<float> ch, cv;
    isp_vignette_curves(cfg, w, h, ch, cv);
    if (cfg.pipe)  // initPipe stores curveHAtPixel's v[0], v[2], v[1] (CameraIspPipe.h:84-90); the vertical table is straight
      for (int j = 0; j < w; ++j) std::swap(ch[(size_t)j * 3 + 1], ch[(size_t)j * 3 + 2]);
    cur->h_.ensure(ch.size() * sizeof(float));
    cur->v_.ensure(cv.size() * sizeof(float));
    S360_HIP(hipMemcpyAsync(cur->h_.p, ch.data(), ch.size() * sizeof(float), hipMemcpyHostToDevice, st));
    S360_HIP(hipMemcpyAsync(cur->v_.p, cv.data(), cv.size() * sizeof(float), hipMemcpyHostToDevice, st));
    S360_HIP(hipStreamSynchronize(st));  // the staging vectors go out of scope
    cur->w = w;
    cur->h = h;
  }
  const size_t outBytes = n * 3 * (cfg.output_bpp == 8 ? 1 : 2);
  if (cfg.pipe) {
    pipe_enqueue(o, st, o->pipe, w, h, cur->h_.as<float>(), cur->v_.as<float>(), o->dToneTab.as<unsigned short>());
    return;
  }
  o->dPlane.ensure(n * sizeof(float));
  o->dImg.ensure(n * 3 * sizeof(float));
  o->dOut.ensure(outBytes);
  if (cfg.demosaic_filter == 2) {
    o->dGV.ensure(n * sizeof(float));
    o->dGH.ensure(n * sizeof(float));
    o->dGreen.ensure(n * sizeof(float));
    o->dFlag.ensure(n);
  }
  if (o->dev.stuckR > 0) {
    o->dStuck.ensure((n + w) * 5 + (size_t)w * 4 + 64);
    o->dStuckCount.ensure(2 * sizeof(unsigned));
    if (!o->hStuckCount) S360_HIP(hipHostMalloc((void**)&o->hStuckCount, 2 * sizeof(unsigned), hipHostMallocDefault));
  }
  if (o->dev.sharpen) {
    o->dLp.ensure(n * 3 * sizeof(float));
    o->dScratch.ensure(n * 3 * sizeof(float));
    o->dState.ensure((size_t)3 * std::max(w, h) * sizeof(float));
  }
  IspFrameBufs B;
  B.plane = o->dPlane.as<float>();
  B.gV = o->dGV.as<float>();
  B.gH = o->dGH.as<float>();
  B.green = o->dGreen.as<float>();
  B.img = o->dImg.as<float>();
  B.lp = o->dLp.as<float>();
  B.scratch = o->dScratch.as<float>();
  B.state = o->dState.as<float>();
  B.flag = o->dFlag.as<unsigned char>();
  B.stuckCand0 = o->dStuck.as<float>();  // floats: n (image), w (row); ints: w; bytes: n, w
  B.stuckCand = B.stuckCand0 + n;
  B.stuckDirty = reinterpret_cast<int*>(B.stuckCand + w);
  B.stuckAct0 = reinterpret_cast<unsigned char*>(B.stuckDirty + w);
  B.stuckAct = B.stuckAct0 + n;
  B.curveH = cur->h_.as<float>();
  B.curveV = cur->v_.as<float>();
  B.lut = o->dLut.as<float>();
  B.exptab = o->dExp.as<unsigned long long>();
  B.stuckCount = o->dStuckCount.as<unsigned>();
  B.stuckBudget = stuck_pixel_budget();
  isp_launch(st, o->dev, o->dRaw.as<unsigned short>(), inW, inH, B, o->dOut.p);
  S360_HIP(hipGetLastError());
  if (o->dev.stuckR > 0 && B.stuckBudget) {
    // removeStuckPixels where it changes pixels is a serial walk (k_isp_stuck): the one configuration of the ISP whose cost is not
    // bounded by the image's size alone. Its verdict comes back with a wait — this configuration pays tens of milliseconds for the
    // pass itself —, and a frame over the budget is refused instead of rendered from an unfiltered plane.
    S360_HIP(hipMemcpyAsync(o->hStuckCount, o->dStuckCount.p, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    S360_HIP(hipStreamSynchronize(st));
    if (o->hStuckCount[1])
      throw Error(S360_ERR_INVALID_ARG, "removeStuckPixels (stuckPixelRadius > 0, threshold " + std::to_string(o->dev.stuckThr) +
                                            ") would rewrite " + std::to_string(o->hStuckCount[0]) + " pixels one after the other (budget " +
                                            std::to_string(B.stuckBudget) + ", ~1-5 us each in one workgroup): refused; S360_ISP_STUCK_BUDGET=<pixels> "
                                            "raises the budget, 0 removes it");
  }
}

static void isp_run_uploaded(s360_isp* o, int inW, int inH, void* out) {
  isp_enqueue(o, o->st, inW, inH);
  const size_t outBytes = (size_t)(inW / o->cfg.resize) * (inH / o->cfg.resize) * 3 * (o->cfg.output_bpp == 8 ? 1 : 2);
  S360_HIP(hipMemcpyAsync(out, o->dOut.p, outBytes, hipMemcpyDeviceToHost, o->st));
  S360_HIP(hipStreamSynchronize(o->st));
}
// For callers that keep the result on the device (render.hip: camera images straight into a frame): the raw frame must
// already be in dRaw (isp_raw_buffer) and everything is enqueued on the caller's stream.
void* isp_raw_buffer(s360_isp* o, int inW, int inH) {
  o->dRaw.ensure((size_t)inW * inH * sizeof(uint16_t));
  return o->dRaw.p;
}
// ... and for a frame that arrives as the sensor's packed bytes (a capture container's frame): the caller copies them into
// this buffer on its stream, isp_unpack_on widens them into dRaw there (RawConverter.cpp:15-59)
void* isp_packed_buffer(s360_isp* o, int bits, int inW, int inH) {
  if (bits != 8 && bits != 12) throw Error(S360_ERR_INVALID_ARG, "packed frames are 8 or 12 bits per pixel");
  if (bits == 12 && (inW & 1)) throw Error(S360_ERR_INVALID_ARG, "12-bit packed frames need an even width");
  o->dPacked.ensure(isp_packed_bytes(bits, inW, inH));
  o->dRaw.ensure((size_t)inW * inH * sizeof(uint16_t));
  return o->dPacked.p;
}
size_t isp_packed_bytes(int bits, int inW, int inH) { return bits == 8 ? (size_t)inW * inH : (size_t)inH * (3 * (size_t)inW / 2); }
void isp_unpack_on(s360_isp* o, hipStream_t st, int bits, int inW, int inH) {
  isp_launch_unpack(st, o->dPacked.as<unsigned char>(), bits, inW, inH, o->dRaw.as<unsigned short>());
}
const void* isp_enqueue_on(s360_isp* o, hipStream_t st, unsigned long long ctxUid, int inW, int inH) {
  // Bound to the CONTEXT (its uid, never reused), not to the value of its stream handle: once that context has been
  // destroyed — s360_destroy waits for its upload stream, so nothing of it still runs on this object's buffers — the
  // object may feed another one; while it lives, a second context is refused.
  if (o->boundCtx && o->boundCtx != ctxUid && context_alive(o->boundCtx))
    throw Error(S360_ERR_STATE, "an ISP object feeds ONE context (s360_frame_upload_raw): create one per context");
  o->boundCtx = ctxUid;
  isp_enqueue(o, st, inW, inH);
  return o->dOut.p;
}

void isp_release(s360_isp* o) {
  if (o->hStuckCount) { (void)hipHostFree(o->hStuckCount); o->hStuckCount = nullptr; }
  if (o->st) {
    (void)hipSetDevice(o->device);
    (void)hipStreamSynchronize(o->st);
    (void)hipStreamDestroy(o->st);
    o->st = nullptr;
  }
}

}  // namespace s360
