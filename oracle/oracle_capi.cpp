// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
// PARITY UNPINNED (see cvlite.h header).
//
// ctypes-callable C entry points over the CPU restatement. Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>

#include "isp.h"
#include "isp_pipe.h"
#include "render.h"

using namespace orc;

extern "C" {

struct orc_camera_c {
  int type;  // 0 FTHETA, 1 RECTILINEAR
  int is_side;
  int has_fov;
  int pad_;
  double origin[3], forward[3], up[3], right[3];
  double resolution[2], principal[2], distortion[2], focal[2];
  double fov;
};

struct orc_params_c {
  double interpupilary_dist, zero_parallax_dist, sharpening;
  int side_alpha_feather_size, std_alpha_feather_size;
  int enable_top, enable_bottom;
  int eqr_width, eqr_height, final_eqr_width, final_eqr_height;
  int side_flow_search20, polar_flow_search20;
  int enable_pole_removal, poleremoval_flow_search20;
};
}

static Camera makeCamera(const orc_camera_c& c) {  // Camera.cpp:44-83
  Camera cam;
  cam.type = c.type;
  cam.position = V3(c.origin[0], c.origin[1], c.origin[2]);
  cam.setRotation(V3(c.forward[0], c.forward[1], c.forward[2]), V3(c.up[0], c.up[1], c.up[2]),
                  V3(c.right[0], c.right[1], c.right[2]));
  cam.resolution.x = c.resolution[0]; cam.resolution.y = c.resolution[1];
  cam.principal.x = c.principal[0]; cam.principal.y = c.principal[1];
  cam.distortion.x = c.distortion[0]; cam.distortion.y = c.distortion[1];
  if (c.has_fov) cam.setFov(c.fov); else cam.setDefaultFov();
  cam.focal.x = c.focal[0]; cam.focal.y = c.focal[1];
  cam.group = c.is_side ? "side camera" : "";
  return cam;
}
static RenderParams makeParams(const orc_params_c& p) {
  RenderParams P;
  P.interpupilary_dist = p.interpupilary_dist;
  P.zero_parallax_dist = p.zero_parallax_dist;
  P.sharpening = p.sharpening;
  P.side_alpha_feather_size = p.side_alpha_feather_size;
  P.std_alpha_feather_size = p.std_alpha_feather_size;
  P.enable_top = p.enable_top;
  P.enable_bottom = p.enable_bottom;
  P.eqr_width = p.eqr_width; P.eqr_height = p.eqr_height;
  P.final_eqr_width = p.final_eqr_width; P.final_eqr_height = p.final_eqr_height;
  P.side_flow_alg = p.side_flow_search20 ? "pixflow_search_20" : "pixflow_low";
  P.polar_flow_alg = p.polar_flow_search20 ? "pixflow_search_20" : "pixflow_low";
  P.enable_pole_removal = p.enable_pole_removal;
  P.poleremoval_flow_alg = p.poleremoval_flow_search20 ? "pixflow_search_20" : "pixflow_low";
  return P;
}
static ImgU8 wrapU8(const uint8_t* p, int w, int h, int c) {
  ImgU8 i(w, h, c);
  if (p) std::memcpy(i.d.data(), p, i.bytes());
  return i;
}
static ImgF wrapF(const float* p, int w, int h, int c) {
  ImgF i(w, h, c);
  if (p) std::memcpy(i.d.data(), p, i.bytes());
  return i;
}

extern "C" {

// ---- primitives (cvlite.h) -------------------------------------------------
void orc_resize_cubic_u8(const uint8_t* s, int sw, int sh, int c, uint8_t* d, int dw, int dh) {
  ImgU8 r = resizeCubicU8(wrapU8(s, sw, sh, c), dw, dh);
  std::memcpy(d, r.d.data(), r.bytes());
}
void orc_resize_cubic_f32(const float* s, int sw, int sh, int c, float* d, int dw, int dh) {
  ImgF r = resizeCubicF32(wrapF(s, sw, sh, c), dw, dh);
  std::memcpy(d, r.d.data(), r.bytes());
}
void orc_resize_linear_f32(const float* s, int sw, int sh, int c, float* d, int dw, int dh) {
  ImgF r = resizeLinearF32(wrapF(s, sw, sh, c), dw, dh);
  std::memcpy(d, r.d.data(), r.bytes());
}
void orc_remap_cubic_u8(const uint8_t* s, int sw, int sh, int c, const float* map, int dw, int dh, uint8_t* d) {
  ImgU8 r = remapCubicU8(wrapU8(s, sw, sh, c), wrapF(map, dw, dh, 2));
  std::memcpy(d, r.d.data(), r.bytes());
}
void orc_remap_cubic_f32(const float* s, int sw, int sh, int c, const float* map, int dw, int dh, float* d) {
  ImgF r = remapCubicF32(wrapF(s, sw, sh, c), wrapF(map, dw, dh, 2));
  std::memcpy(d, r.d.data(), r.bytes());
}
void orc_gaussian_blur_f32(const float* s, int w, int h, int c, int ksize, double sigma, float* d) {
  ImgF r = gaussianBlurF32(wrapF(s, w, h, c), ksize, sigma);
  std::memcpy(d, r.d.data(), r.bytes());
}
void orc_gaussian_kernel(int n, double sigma, float* out) {
  std::vector<float> k = gaussianKernel(n, sigma);
  std::memcpy(out, k.data(), n * sizeof(float));
}
void orc_sobel(const float* s, int w, int h, int dirY, float* d) {
  ImgF r = dirY ? sobelY(wrapF(s, w, h, 1)) : sobelX(wrapF(s, w, h, 1));
  std::memcpy(d, r.d.data(), r.bytes());
}
void orc_median5_f32(const float* s, int w, int h, int c, float* d) {
  ImgF r = medianBlur5(wrapF(s, w, h, c));
  std::memcpy(d, r.d.data(), r.bytes());
}
void orc_bicubic_tab(float* tabf /*1024*16*/, short* tabi /*1024*16*/) {
  const BicubicTab& T = bicubicTab();
  if (tabf) std::memcpy(tabf, T.f, sizeof(T.f));
  if (tabi) std::memcpy(tabi, T.i, sizeof(T.i));
}
void orc_feather_alpha_channel(const uint8_t* s, int w, int h, int erodeSize, uint8_t* d) {
  ImgU8 r = featherAlphaChannel(wrapU8(s, w, h, 4), erodeSize);
  std::memcpy(d, r.d.data(), r.bytes());
}
void orc_offset_horizontal_wrap(const uint8_t* s, int w, int h, int c, float offset, uint8_t* d) {
  ImgU8 r = offsetHorizontalWrap(wrapU8(s, w, h, c), offset);
  std::memcpy(d, r.d.data(), r.bytes());
}
void orc_flatten_layers_deghost_prefer_base(const uint8_t* base, const uint8_t* top, int w, int h, uint8_t* d) {
  ImgU8 r = flattenLayersDeghostPreferBase(wrapU8(base, w, h, 4), wrapU8(top, w, h, 4));
  std::memcpy(d, r.d.data(), r.bytes());
}
void orc_sharpen(uint8_t* bgr, int w, int h, float sharpening) {
  ImgU8 i = wrapU8(bgr, w, h, 3);
  sharpen(i, sharpening);
  std::memcpy(bgr, i.d.data(), i.bytes());
}

// ---- PixFlow ---------------------------------------------------------------
// number of pyramid levels and their sizes for an input of w x h (PixFlow.h:477-491)
int orc_pixflow_levels(int w, int h, int* lw /*cap 64*/, int* lh) {
  PixFlowParams p;
  int cw = int(w * p.downscaleFactor), ch = int(h * p.downscaleFactor), n = 0;
  for (;;) {
    if (lw) { lw[n] = cw; lh[n] = ch; }
    ++n;
    const int nw = int(cw * p.pyrScaleFactor + 0.5f), nh = int(ch * p.pyrScaleFactor + 0.5f);
    if (nh <= PixFlow::kPyrMinImageSize || nw <= PixFlow::kPyrMinImageSize || n >= 64) break;
    cw = nw; ch = nh;
  }
  return n;
}
// OpticalFlowInterface::computeOpticalFlow (OpticalFlowInterface.h:34-41). prev_* nullable.
// level_flows (nullable): concatenated per-level flow (coarsest first) for debugging.
// Returns 0, or -1 for an unknown algorithm name (the reference throws VrCamException).
int orc_compute_optical_flow(const char* alg, const uint8_t* i0, const uint8_t* i1, int w, int h,
                             const float* prev_flow, const uint8_t* prev_i0, const uint8_t* prev_i1, int hint,
                             float* flow_out, float* level_flows) {
  PixFlowParams fp;
  if (!pixflowParamsByName(alg, &fp)) return -1;
  PixFlow pf(fp);
  PixFlow::Debug dbg;
  if (level_flows) pf.dbg = &dbg;
  ImgF flow;
  ImgF pfl = prev_flow ? wrapF(prev_flow, w, h, 2) : ImgF();
  ImgU8 p0 = prev_flow ? wrapU8(prev_i0, w, h, 4) : ImgU8();
  ImgU8 p1 = prev_flow ? wrapU8(prev_i1, w, h, 4) : ImgU8();
  pf.computeOpticalFlow(wrapU8(i0, w, h, 4), wrapU8(i1, w, h, 4), pfl, p0, p1, flow, hint);
  std::memcpy(flow_out, flow.d.data(), flow.bytes());
  if (level_flows) {
    size_t off = 0;
    for (const ImgF& f : dbg.flowPerLevel) {
      std::memcpy(level_flows + off, f.d.data(), f.bytes());
      off += f.d.size();
    }
  }
  return 0;
}
// Entry stage only (PixFlow.h:93-138): downscale + grey/alpha + pre-blur at level 0.
void orc_pixflow_entry(const uint8_t* i0, int w, int h, uint8_t* down, float* I, float* alpha) {
  PixFlowParams fp;
  const int dw = int(w * fp.downscaleFactor), dh = int(h * fp.downscaleFactor);
  ImgU8 d0 = resizeCubicU8(wrapU8(i0, w, h, 4), dw, dh);
  std::memcpy(down, d0.d.data(), d0.bytes());
  const float inv255 = (float)(1.0 / 255.0);
  ImgF I0(dw, dh, 1);
  for (int y = 0; y < dh; ++y)
    for (int x = 0; x < dw; ++x) {
      const uint8_t* p = d0.px(y, x);
      I0.at(y, x) = float(bgr2gray(p[0], p[1], p[2])) * inv255;
      alpha[size_t(y) * dw + x] = float(p[3]) * inv255;
    }
  I0 = gaussianBlurF32(I0, PixFlow::kPreBlurKernelWidth, PixFlow::kPreBlurSigma);
  std::memcpy(I, I0.d.data(), I0.bytes());
}
// One level of patchMatchPropagationAndSearch (PixFlow.h:344-413) for kernel-level parity tests.
void orc_pixflow_level(const float* I0, const float* I1, const float* a0, const float* a1, int w, int h,
                       float* flow_inout, int has_flow, int hint, int search20) {
  PixFlowParams fp;
  if (search20) fp.maxPercentage = 20;
  PixFlow pf(fp);
  ImgF flow = has_flow ? wrapF(flow_inout, w, h, 2) : ImgF();
  pf.patchMatchPropagationAndSearch(wrapF(I0, w, h, 1), wrapF(I1, w, h, 1), wrapF(a0, w, h, 1), wrapF(a1, w, h, 1),
                                    flow, hint);
  std::memcpy(flow_inout, flow.d.data(), flow.bytes());
}

// ---- geometry ----------------------------------------------------------------
void orc_camera_rotation(const orc_camera_c* c, double* R9) {
  Camera cam = makeCamera(*c);
  std::memcpy(R9, cam.R, sizeof(cam.R));
}
void orc_camera_pixel(const orc_camera_c* c, const double* rig3, double* pix2) {
  V2 p = makeCamera(*c).pixel(V3(rig3[0], rig3[1], rig3[2]));
  pix2[0] = p.x; pix2[1] = p.y;
}
void orc_camera_rig_near_infinity(const orc_camera_c* c, const double* pix2, double* rig3) {
  V2 p; p.x = pix2[0]; p.y = pix2[1];
  V3 r = makeCamera(*c).rigNearInfinity(p);
  rig3[0] = r.x; rig3[1] = r.y; rig3[2] = r.z;
}
double orc_camera_get_fov(const orc_camera_c* c) { return makeCamera(*c).getFov(); }
int orc_camera_sees(const orc_camera_c* c, const double* rig3) {
  return makeCamera(*c).sees(V3(rig3[0], rig3[1], rig3[2])) ? 1 : 0;
}
double orc_camera_undistort_distort(const orc_camera_c* c, double r) {
  Camera cam = makeCamera(*c);
  return cam.undistort(cam.distort(r));
}
float orc_approximate_fov(const orc_camera_c* c, int vertical) { return approximateFov(makeCamera(*c), vertical != 0); }

void orc_spherical_warp_map(const orc_camera_c* c, int dw, int dh, float l, float r, float t, float b, float* map) {
  ImgF m = sphericalWarpMap(dw, dh, makeCamera(*c), l, r, t, b);
  std::memcpy(map, m.d.data(), m.bytes());
}
void orc_bicubic_remap_to_spherical(const orc_camera_c* c, const uint8_t* src, int sw, int sh, int sc, int dw, int dh,
                                    int dc, float l, float r, float t, float b, uint8_t* dst) {
  ImgU8 d = bicubicRemapToSpherical(dw, dh, dc, wrapU8(src, sw, sh, sc), makeCamera(*c), l, r, t, b);
  std::memcpy(dst, d.d.data(), d.bytes());
}

// ---- frame-level -------------------------------------------------------------
struct orc_frame {
  RigDescription rig;
  RenderParams P;
  SideGeometry g;
  FrameState state[2];
  int cur = 0;
  bool havePrev = false;
  FrameDebug dbg;
  ImgU8 out;
  double stage[5] = {0, 0, 0, 0, 0};
  std::map<std::string, const ImgU8*> named;
  PoleRemovalInput poleRemoval;
  bool havePoleRemoval = false;
};

orc_frame* orc_frame_create(const orc_camera_c* cams, int ncams, const orc_params_c* p) {
  orc_frame* f = new orc_frame;
  for (int i = 0; i < ncams; ++i) f->rig.rig.push_back(makeCamera(cams[i]));
  f->rig.finalize();
  f->P = makeParams(*p);
  f->g = sideGeometry(f->rig, f->P);
  return f;
}
void orc_frame_destroy(orc_frame* f) { delete f; }
// ints: camImageWidth, camImageHeight, overlapImageWidth, numNovelViews, poleRows(top), poleRows(bottom)
// floats: hRadians, vRadians, fovHorizontalRadians, vergeDisp, zeroParallaxShift
void orc_frame_geometry(orc_frame* f, int* ints6, float* floats5) {
  const SideGeometry& g = f->g;
  ints6[0] = g.camImageWidth; ints6[1] = g.camImageHeight; ints6[2] = g.overlapImageWidth; ints6[3] = g.numNovelViews;
  ints6[4] = int(f->P.eqr_height * f->rig.findCameraByDirection(V3(0, 0, 1)).getFov() / M_PI);
  ints6[5] = int(f->P.eqr_height * f->rig.findCameraByDirection(V3(0, 0, -1)).getFov() / M_PI);
  floats5[0] = g.hRadians; floats5[1] = g.vRadians; floats5[2] = g.fovHorizontalRadians;
  floats5[3] = g.vergeAtInfinitySlabDisplacement; floats5[4] = g.zeroParallaxNovelViewShiftPixels;
}
void orc_frame_pole_ramp(orc_frame* f, float* out4) {
  PoleRamp r = poleRamp(f->rig);
  out4[0] = r.poleCameraRadius; out4[1] = r.phiRampStart; out4[2] = r.phiMid; out4[3] = r.phiRampEnd;
}
// side: ncams pointers to w*h*ch images (ch 3 or 4). use_prev: temporal regularisation against the previous
// call's state (the reference's --prev_frame_data_dir). Returns seconds of total runtime.
double orc_frame_render(orc_frame* f, const uint8_t* const* side, int w, int h, int ch, const uint8_t* top,
                        const uint8_t* bottom, int pw, int ph, int use_prev, int threaded) {
  std::vector<ImgU8> imgs;
  for (size_t i = 0; i < f->rig.rigSideOnly.size(); ++i) imgs.push_back(wrapU8(side[i], w, h, ch));
  ImgU8 t = top ? wrapU8(top, pw, ph, 3) : ImgU8();
  ImgU8 b = bottom ? wrapU8(bottom, pw, ph, 3) : ImgU8();
  const FrameState* prev = (use_prev && f->havePrev) ? &f->state[f->cur ^ 1] : nullptr;
  FrameState* st = &f->state[f->cur];
  *st = FrameState();
  f->out = renderStereoPanorama(f->rig, f->P, imgs, t, b, prev, st, &f->dbg, threaded != 0, f->stage,
                                f->havePoleRemoval ? &f->poleRemoval : nullptr);
  f->havePrev = true;
  f->cur ^= 1;
  return f->stage[4];
}
// Secondary bottom camera image + the two red pole masks (BGR, same size) for --enable_pole_removal (TRSP:569-597).
void orc_frame_set_pole_removal(orc_frame* f, const uint8_t* bottom2, const uint8_t* mask, const uint8_t* mask2, int w,
                                int h) {
  f->poleRemoval.bottom2 = wrapU8(bottom2, w, h, 3);
  f->poleRemoval.mask = wrapU8(mask, w, h, 3);
  f->poleRemoval.mask2 = wrapU8(mask2, w, h, 3);
  f->havePoleRemoval = true;
}
// Index (in rig order) of RigDescription::findLargestDistCamAxisToRigCenter and Camera::approximateUsablePixelsRadius.
int orc_frame_bottom2_index(orc_frame* f) {
  const Camera& c = f->rig.findLargestDistCamAxisToRigCenter();
  return (int)(&c - &f->rig.rig[0]);
}
float orc_frame_usable_pixels_radius(orc_frame* f, int cam_idx) { return approximateUsablePixelsRadius(f->rig.rig[cam_idx]); }
void orc_frame_stage_seconds(orc_frame* f, double* out5) { std::memcpy(out5, f->stage, sizeof(f->stage)); }

static const ImgU8* frameImage(orc_frame* f, const char* name, int idx) {
  const std::string n(name);
  const FrameState& st = f->state[f->cur ^ 1];
  if (n == "out") return &f->out;
  if (n == "projection") return &f->dbg.projections[idx];
  if (n == "side_pano_l") return &f->dbg.sidePanoL;
  if (n == "side_pano_r") return &f->dbg.sidePanoR;
  if (n == "top_spherical") return &f->dbg.topSpherical;
  if (n == "bottom_spherical") return &f->dbg.bottomSpherical;
  if (n == "pole_warped") return &f->dbg.poleWarped[idx];
  if (n == "eye_l") return &f->dbg.eyeL;
  if (n == "eye_r") return &f->dbg.eyeR;
  if (n == "overlap_l") return &st.overlapL[idx];
  if (n == "overlap_r") return &st.overlapR[idx];
  if (n == "extended_side") return &st.pole[idx].extendedSide;
  if (n == "extended_fisheye") return &st.pole[idx].extendedFisheye;
  if (n == "bottom_image") return &st.poleRemoval.bottomImage;
  if (n == "bottom_image2") return &st.poleRemoval.bottomImage2;
  return nullptr;
}
static const ImgF* frameFlow(orc_frame* f, const char* name, int idx) {
  const std::string n(name);
  const FrameState& st = f->state[f->cur ^ 1];
  if (n == "flow_l_to_r") return &st.flowLtoR[idx];
  if (n == "flow_r_to_l") return &st.flowRtoL[idx];
  if (n == "flow_pole") return &st.pole[idx].flow;
  if (n == "flow_bottom_secondary") return &st.poleRemoval.flow;
  return nullptr;
}
// Query dims (whc3) then copy (dst may be null for a dims-only query). Returns 0 ok, -1 unknown name.
int orc_frame_get_u8(orc_frame* f, const char* name, int idx, int* whc3, uint8_t* dst) {
  const ImgU8* i = frameImage(f, name, idx);
  if (!i) return -1;
  whc3[0] = i->w; whc3[1] = i->h; whc3[2] = i->c;
  if (dst) std::memcpy(dst, i->d.data(), i->bytes());
  return 0;
}
int orc_frame_get_f32(orc_frame* f, const char* name, int idx, int* whc3, float* dst) {
  const ImgF* i = frameFlow(f, name, idx);
  if (!i) return -1;
  whc3[0] = i->w; whc3[1] = i->h; whc3[2] = i->c;
  if (dst) std::memcpy(dst, i->d.data(), i->bytes());
  return 0;
}

// Stereo cubemap of the last rendered frame (TRSP:917-935). Query dims with dst == nullptr. -1: no frame / bad format.
int orc_frame_cubemap(orc_frame* f, int face_w, int face_h, const char* format, int* whc3, uint8_t* dst) {
  const std::string fmt(format);
  if ((fmt != "video" && fmt != "photo") || f->dbg.eyeL.w == 0) return -1;
  whc3[0] = fmt == "video" ? 3 * face_w : face_w;
  whc3[1] = fmt == "video" ? 4 * face_h : 12 * face_h;
  whc3[2] = 3;
  if (dst) {
    const ImgU8 c = stereoCubemap(f->dbg.eyeL, f->dbg.eyeR, face_w, face_h, fmt);
    std::memcpy(dst, c.d.data(), c.bytes());
  }
  return 0;
}

// NovelViewGeneratorLazyFlow::combineLazyNovelViews for one pair (NovelView.cpp:226-268) using the frame geometry.
void orc_frame_combine_lazy_novel_views(orc_frame* f, const uint8_t* imgL, const uint8_t* imgR, const float* flowLtoR,
                                        const float* flowRtoL, uint8_t* chunkL, uint8_t* chunkR) {
  const SideGeometry& g = f->g;
  const int w = g.overlapImageWidth, h = g.camImageHeight;
  auto lr = combineLazyNovelViews(g, f->P.eqr_width, (int)f->rig.rigSideOnly.size(), wrapU8(imgL, w, h, 4),
                                  wrapU8(imgR, w, h, 4), wrapF(flowLtoR, w, h, 2), wrapF(flowRtoL, w, h, 2));
  std::memcpy(chunkL, lr.first.d.data(), lr.first.bytes());
  std::memcpy(chunkR, lr.second.d.data(), lr.second.bytes());
}
// poleToSideFlowThread (TRSP:388-561) standalone, no temporal state.
void orc_frame_pole_to_side_flow(orc_frame* f, const uint8_t* side, int sw, int sh, const uint8_t* pole, int pw, int ph,
                                 uint8_t* warped /*sw*sh*4*/, float* flow_out /*ext*ph*2, nullable*/) {
  PoleFlowState st;
  ImgU8 r = poleToSideFlow(f->rig, f->P, wrapU8(side, sw, sh, 4), wrapU8(pole, pw, ph, 4), nullptr, &st);
  std::memcpy(warped, r.d.data(), r.bytes());
  if (flow_out) std::memcpy(flow_out, st.flow.d.data(), st.flow.bytes());
}


// ---- soft ISP (isp.h; CameraIsp.h through Raw2Rgb's non-accelerated path) ----------------------
// cfg: orc::IspConfig (4-byte fields only; mirrored field by field by tests/oracle_lib.py IspConfigC)
int orc_isp_config_size() { return (int)sizeof(IspConfig); }
int orc_isp_run(const IspConfig* cfg, const uint16_t* raw, int w, int h, void* out, char* err, int err_cap) {
  try {
    ispRun(*cfg, raw, w, h, out);
    return 0;
  } catch (const std::exception& e) {
    if (err && err_cap > 0) { std::strncpy(err, e.what(), err_cap - 1); err[err_cap - 1] = 0; }
    return -1;
  }
}
// the accelerated ISP's arithmetic (isp_pipe.h; CameraIspGen.cpp restated and pinned to that generator executed, oracle/_ref/libref_isppipe.so). fast: CameraIspGenFast
int orc_isp_pipe_run(const IspConfig* cfg, int fast, const uint16_t* raw, int w, int h, void* out, char* err, int err_cap) {
  try {
    ispPipeRun(*cfg, fast != 0, raw, w, h, out);
    return 0;
  } catch (const std::exception& e) {
    if (err && err_cap > 0) { std::strncpy(err, e.what(), err_cap - 1); err[err_cap - 1] = 0; }
    return -1;
  }
}
void orc_isp_unpack_frame(int bits, const uint8_t* frame, int w, int h, uint16_t* out) { ispUnpackFrame(bits, frame, w, h, out); }
void orc_isp_tables(const IspConfig* cfg, float* ccm9, float* lut /*4096 x 3*/) {
  const IspTables t = ispSetup(*cfg);
  std::memcpy(ccm9, t.compositeCCM, sizeof(t.compositeCCM));
  std::memcpy(lut, t.toneLut.data(), t.toneLut.size() * sizeof(float));
}

}  // extern "C"
