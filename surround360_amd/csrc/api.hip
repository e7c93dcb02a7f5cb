// api.hip — the C ABI of libs360 (include/s360.h). Thin: argument checks, host<->device copies for
// the operator-level calls, exception -> error-code translation. No CPU fallback anywhere.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <sstream>

#include "../../include/s360.h"
#include "ctx.hpp"
#include "isp.hpp"
#include "render.hpp"

using namespace s360;

static thread_local std::string g_err;

// live contexts by uid
namespace {
std::mutex g_ctxMu;
std::vector<unsigned long long> g_ctxLive;
unsigned long long g_ctxNext = 0;
}  // namespace
namespace s360 {
bool context_alive(unsigned long long uid) {
  std::lock_guard<std::mutex> lk(g_ctxMu);
  return std::find(g_ctxLive.begin(), g_ctxLive.end(), uid) != g_ctxLive.end();
}
}  // namespace s360
// buffers handed out by s360_host_alloc (page-locked host memory)
namespace {
std::mutex g_pinMu;
std::vector<std::pair<const char*, size_t>> g_pinBlocks;
}  // namespace
namespace s360 {
bool host_is_pinned(const void* p, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_pinMu);
  const char* q = static_cast<const char*>(p);
  for (const auto& b : g_pinBlocks)
    if (q >= b.first && q + bytes <= b.first + b.second) return true;
  return false;
}
}  // namespace s360

template <typename F>
static int guard(s360_ctx* c, F&& f) {
  // thread-safe per context: concurrent callers of one context are serialised here
  std::unique_lock<std::recursive_mutex> lk;
  if (c) lk = std::unique_lock<std::recursive_mutex>(c->mu);
  try {
    if (c) c->make_current();
    f();
    return S360_OK;
  } catch (const Error& e) {
    (c ? c->err : g_err) = e.what();
    g_err = e.what();
    return e.code;
  } catch (const std::exception& e) {
    (c ? c->err : g_err) = e.what();
    g_err = e.what();
    return S360_ERR_STATE;
  }
}
// Entry points that BLOCK on the device (a finished frame's download, waiting for uploads) give the context back while
// they wait: f receives the held lock and may unlock / relock it around the wait, so that another host thread can feed
// the next frame meanwhile (a stream's uploader beside the thread that fetches and encodes).
template <typename F>
static int guard_l(s360_ctx* c, F&& f) {
  std::unique_lock<std::recursive_mutex> lk(c->mu);
  try {
    c->make_current();
    if (!c->frame_invalid.empty()) throw Error(S360_ERR_INVALID_ARG, c->frame_invalid);
    f(lk);
    if (!lk.owns_lock()) lk.lock();
    return S360_OK;
  } catch (const Error& e) {
    if (!lk.owns_lock()) lk.lock();
    c->err = e.what();
    g_err = e.what();
    return e.code;
  } catch (const std::exception& e) {
    if (!lk.owns_lock()) lk.lock();
    c->err = e.what();
    g_err = e.what();
    return S360_ERR_STATE;
  }
}
// s360_frame_* entry points: as guard, and refused when the context's flags describe no renderable frame (s360_create)
template <typename F>
static int frame_guard(s360_ctx* c, F&& f) {
  return guard(c, [&] {
    if (c && !c->frame_invalid.empty()) throw Error(S360_ERR_INVALID_ARG, c->frame_invalid);
    f();
  });
}
static void need(bool ok, const char* what) {
  if (!ok) throw Error(S360_ERR_INVALID_ARG, what);
}
// The sharded frame's split phases (exchange / pole units / gather / composite: comm.cpp) enqueue their RCCL calls on the
// main stream while frame pipelining runs the pole stage and the composite on the second one: nothing orders the two, so
// the combination is refused instead of racing (a stream keeps its temporal state on ONE GPU anyway: DESIGN.md section 7).
static void no_pipelining(s360_ctx* c) {
  if (c->pipeline) throw Error(S360_ERR_STATE, "the sharded (multi-GPU) frame is not available while frame pipelining is on (s360_set_frame_pipelining)");
}
static void h2d(s360_ctx* c, void* d, const void* h, size_t n) {
  S360_HIP(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, c->st));
}
static void check_sweep_error(s360_ctx* c) {
  unsigned e = 0;
  if (c->flow) e |= c->flow->take_error(c->st);
  if (c->flow_pole) e |= c->flow_pole->take_error(c->st);
  if (c->flow_pr) e |= c->flow_pr->take_error(c->st);
  if (e) throw Error(S360_ERR_HIP, "banded sweep timed out waiting for a neighbour band (results invalid)");
}
// Frame pipelining: frame_finish runs on st2. Anything enqueued on st that READS what frame_finish writes (panoramas,
// warped poles, extended pole images, the stacked equirect) must be ordered after it: the host waits for st2 first.
static void wait_finish_stream(s360_ctx* c) {
  if (c->st2) S360_HIP(hipStreamSynchronize(c->st2));
}
static void d2h(s360_ctx* c, void* h, const void* d, size_t n) {
  if (c->st2) S360_HIP(hipStreamSynchronize(c->st2));  // frame pipelining: the data may come from the finish stream
  S360_HIP(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, c->st));
  S360_HIP(hipStreamSynchronize(c->st));
  check_sweep_error(c);
}

extern "C" {

const char* s360_version(void) { return "surround360_amd 0.1 (gfx950)"; }
int s360_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
const char* s360_last_error(const s360_ctx* ctx) {
  if (!ctx) return g_err.c_str();
  // a copy per calling thread: another thread's failing call may replace ctx->err at any time
  static thread_local std::string copy;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  copy = ctx->err;
  return copy.c_str();
}

// ---- rig -------------------------------------------------------------------------------------
int s360_rig_load_json(const char* path, s360_camera* cams, int max_cams) {
  int n = 0;
  const int rc = guard(nullptr, [&] {
    need(path && cams, "null argument");
    std::ifstream f(path);
    if (!f) throw Error(S360_ERR_IO, std::string("could not read JSON file: ") + path);
    std::stringstream ss;
    ss << f.rdbuf();
    std::vector<s360_camera> v = parse_rig_json(ss.str());
    need((int)v.size() <= max_cams, "rig has more cameras than max_cams");
    for (size_t i = 0; i < v.size(); ++i) cams[i] = v[i];
    n = (int)v.size();
  });
  return rc == S360_OK ? n : rc;
}
int s360_camera_init(s360_camera* out, int type, const double origin[3], const double forward[3], const double up[3],
                     const double right[3], const double resolution[2], const double* principal,
                     const double* distortion, const double focal[2], const double* fov, const char* group,
                     const char* id) {
  return guard(nullptr, [&] {
    need(out && origin && forward && up && right && resolution && focal, "null argument");
    need(type == S360_CAM_FTHETA || type == S360_CAM_RECTILINEAR, "bad camera type");
    std::memset(out, 0, sizeof(*out));
    out->type = type;
    if (id) std::strncpy(out->id, id, sizeof(out->id) - 1);
    for (int i = 0; i < 3; ++i) out->position[i] = origin[i];
    camera_set_rotation(out, forward, up, right);
    for (int i = 0; i < 2; ++i) {
      out->resolution[i] = resolution[i];
      out->principal[i] = principal ? principal[i] : resolution[i] / 2;
      out->distortion[i] = distortion ? distortion[i] : 0.0;
      out->focal[i] = focal[i];
    }
    if (fov) camera_set_fov(out, *fov); else camera_set_default_fov(out);
    out->is_side = (group && std::strstr(group, "side")) ? 1 : 0;
  });
}
void s360_camera_pixel(const s360_camera* cam, const double rig_point[3], double pixel_out[2]) {
  camera_pixel(cam, rig_point, pixel_out);
}
double s360_camera_get_fov(const s360_camera* cam) { return camera_get_fov(cam); }
static int find_dir(const s360_camera* cams, int n, double z) {
  Rig r;
  r.all.assign(cams, cams + n);
  const double d[3] = {0, 0, z};
  return r.find_by_direction(d);
}
int s360_rig_find_top(const s360_camera* cams, int n) { return find_dir(cams, n, 1.0); }
int s360_rig_find_bottom(const s360_camera* cams, int n) { return find_dir(cams, n, -1.0); }
int s360_rig_find_bottom2(const s360_camera* cams, int n) {
  if (!cams || n <= 0) return -1;
  Rig r;
  r.all.assign(cams, cams + n);
  return r.find_largest_axis_dist();
}
float s360_camera_usable_pixels_radius(const s360_camera* cam) { return cam ? approximate_usable_pixels_radius(cam) : 0.f; }

int s360_derive_geometry(const s360_camera* cams, int n_cams, const s360_params* params, s360_geometry* out) {
  return guard(nullptr, [&] {
    need(cams && params && out && n_cams > 0, "null argument");
    Rig r;
    r.all.assign(cams, cams + n_cams);
    r.finalize();
    need(!r.side.empty(), "rig has no side cameras");
    *out = derive_geometry(r, *params);
  });
}
int s360_pole_ramp(const s360_camera* cams, int n_cams, float out4[4]) {
  return guard(nullptr, [&] {
    need(cams && out4 && n_cams > 0, "null argument");
    Rig r;
    r.all.assign(cams, cams + n_cams);
    r.finalize();
    const double down[3] = {0, 0, -1};
    need(!r.side.empty() && r.find_by_direction(down) >= 0, "rig needs side cameras and a bottom camera");
    const PoleRamp p = pole_ramp(r);
    out4[0] = p.poleCameraRadius; out4[1] = p.phiRampStart; out4[2] = p.phiMid; out4[3] = p.phiRampEnd;
  });
}

// ---- context -----------------------------------------------------------------------------------
int s360_create(s360_ctx** out, int device, const s360_camera* cams, int n_cams, const s360_params* params) {
  if (!out) return S360_ERR_INVALID_ARG;
  *out = nullptr;
  s360_ctx* c = nullptr;
  const int rc = guard(nullptr, [&] {
    need(cams && params && n_cams > 0, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
      throw Error(S360_ERR_NO_DEVICE, "no HIP device available (libs360 has no CPU path)");
    need(device >= 0 && device < ndev, "device index out of range");
    c = new s360_ctx;
    c->device = device;
    S360_HIP(hipSetDevice(device));
    S360_HIP(hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking));
    c->st_user = c->st;
    c->prof.st = c->st;
    c->rig.all.assign(cams, cams + n_cams);
    c->rig.finalize();
    need(!c->rig.side.empty(), "rig has no side cameras");  // RigDescription.cpp:27 CHECK_NE
    c->P = *params;
    pixflow_consts_by_name(c->P.side_flow_alg);   // validate early (throws S360_ERR_UNKNOWN_ALG)
    pixflow_consts_by_name(c->P.polar_flow_alg);
    if (c->P.enable_pole_removal && c->P.poleremoval_flow_alg[0]) pixflow_consts_by_name(c->P.poleremoval_flow_alg);
    c->g = derive_geometry(c->rig, c->P);
    const double up[3] = {0, 0, 1}, down[3] = {0, 0, -1};
    c->top_idx = c->rig.find_by_direction(up);
    c->bottom_idx = c->rig.find_by_direction(down);
    // Sizes no frame can have (the reference runs into OpenCV's assertions somewhere inside the frame, TRSP aborts):
    // every stage indexes pixels with 32-bit integers and sizes its buffers from these numbers. The operator-level
    // entry points (flows, remaps, blends) do not depend on them, so the context is still created; every s360_frame_*
    // call of such a context fails with this message (frame_guard).
    auto frame_needs = [&](bool ok, const char* what) {
      if (!ok && c->frame_invalid.empty()) c->frame_invalid = what;
    };
    frame_needs(c->P.eqr_width >= 1 && c->P.eqr_height >= 1 && c->P.eqr_width <= 65536 && c->P.eqr_height <= 65536,
                "eqr_width / eqr_height must be in 1..65536");
    frame_needs(c->P.final_eqr_width >= 0 && c->P.final_eqr_height >= 0 && c->P.final_eqr_width <= 65536 && c->P.final_eqr_height <= 65536,
                "final_eqr_width / final_eqr_height must be in 0..65536 (0 = no final resize)");
    frame_needs(c->g.cam_image_width >= 1 && c->g.cam_image_height >= 1 && c->g.overlap_image_width >= 1 && c->g.num_novel_views >= 1,
                "eqr_width / eqr_height too small for this rig: a side projection, its overlap or its strip would be empty");
    frame_needs(c->g.out_width >= 1 && c->g.out_height >= 2, "final_eqr_width / final_eqr_height leave no output pixels");
    {  // the flows of a frame run on overlap_image_width x cam_image_height and (eqr_width x 1.2) x pole rows images
      const PixFlowConsts ps = pixflow_consts_by_name(c->P.side_flow_alg), pp = pixflow_consts_by_name(c->P.polar_flow_alg);
      frame_needs(int(c->g.overlap_image_width * ps.downscaleFactor) >= 2 && int(c->g.cam_image_height * ps.downscaleFactor) >= 2,
                  "eqr_width / eqr_height too small for this rig: the side flows need overlap images of at least 2 x 2 pixels after the entry downscale");
      const int extW = int(float(c->P.eqr_width) * 1.2f);
      frame_needs(!c->P.enable_top || c->top_idx < 0 || (int(extW * pp.downscaleFactor) >= 2 && int(c->g.top_rows * pp.downscaleFactor) >= 2),
                  "eqr_width / eqr_height too small for this rig: the top pole flows need at least 2 x 2 pixels after the entry downscale");
      frame_needs(!c->P.enable_bottom || c->bottom_idx < 0 || (int(extW * pp.downscaleFactor) >= 2 && int(c->g.bottom_rows * pp.downscaleFactor) >= 2),
                  "eqr_width / eqr_height too small for this rig: the bottom pole flows need at least 2 x 2 pixels after the entry downscale");
    }
    if (c->bottom_idx >= 0) c->ramp = pole_ramp(c->rig);
    { c->flow.reset(new FlowEngine(&c->prof)); c->flow->set_sweep_mode(c->sweep_mode); }
  });
  if (rc != S360_OK) {
    delete c;
    return rc;
  }
  {
    std::lock_guard<std::mutex> lk(g_ctxMu);
    c->uid = ++g_ctxNext;
    g_ctxLive.push_back(c->uid);
  }
  *out = c;
  return S360_OK;
}
void s360_destroy(s360_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stUp) (void)hipStreamSynchronize(c->stUp);
  if (c->st) (void)hipStreamSynchronize(c->st);
  if (c->st2) (void)hipStreamSynchronize(c->st2);
  {  // (behind the waits: an ISP object bound to this context may be taken over by another one from here on)
    std::lock_guard<std::mutex> lk(g_ctxMu);
    g_ctxLive.erase(std::remove(g_ctxLive.begin(), g_ctxLive.end(), c->uid), g_ctxLive.end());
  }
  comm_destroy(c);
  c->slots.clear();
  c->slotScratch.reset();
  c->flow.reset();
  c->flow_pole.reset();
  c->flow_pr.reset();
  if (c->st) (void)hipStreamDestroy(c->st);
  if (c->st2) {
    (void)hipStreamDestroy(c->st2);
    (void)hipEventDestroy(c->evSideDone);
    (void)hipEventDestroy(c->evStripsFree);
  }
  if (c->evPoleSrcFree) (void)hipEventDestroy(c->evPoleSrcFree);
  if (c->stDown) (void)hipStreamDestroy(c->stDown);
  for (s360_ctx::PackedCache* pc : {&c->sidePk, &c->topPk, &c->botPk})
    for (auto& e : pc->e)
      if (e->ready) (void)hipEventDestroy(e->ready);
  if (c->evMaps) (void)hipEventDestroy(c->evMaps);
  if (c->evDown) (void)hipEventDestroy(c->evDown);
  if (c->downErr) (void)hipHostFree(c->downErr);
  if (c->pngMetaHost) (void)hipHostFree(c->pngMetaHost);
  if (c->evUpHost) (void)hipEventDestroy(c->evUpHost);
  if (c->stUp) {
    (void)hipStreamDestroy(c->stUp);
    (void)hipEventDestroy(c->evUploaded);
    (void)hipEventDestroy(c->evSideSrcFree);
    for (int i = 0; i < s360_ctx::kPinChunks; ++i) {
      if (c->pin[i]) (void)hipHostFree(c->pin[i]);
      if (c->pinEv[i]) (void)hipEventDestroy(c->pinEv[i]);
    }
  }
  delete c;
}
int s360_get_geometry(const s360_ctx* c, s360_geometry* out) {
  if (!c || !out) return S360_ERR_INVALID_ARG;
  *out = c->g;
  return S360_OK;
}
void* s360_stream(s360_ctx* c) { return c ? (void*)c->st_user : nullptr; }  // (immutable after s360_create: no lock needed)
int s360_synchronize(s360_ctx* c) {
  return guard(c, [&] {
    need(c, "null ctx");
    if (c->stUp) S360_HIP(hipStreamSynchronize(c->stUp));
    S360_HIP(hipStreamSynchronize(c->st));
    if (c->st2) S360_HIP(hipStreamSynchronize(c->st2));
    check_sweep_error(c);
  });
}
int s360_set_frame_pipelining(s360_ctx* c, int on) {
  return guard(c, [&] {
    need(c, "null ctx");
    c->make_current();
    S360_HIP(hipStreamSynchronize(c->st));
    if (c->st2) S360_HIP(hipStreamSynchronize(c->st2));
    if (on && !c->st2) {
      S360_HIP(hipStreamCreateWithFlags(&c->st2, hipStreamNonBlocking));
      S360_HIP(hipEventCreateWithFlags(&c->evSideDone, hipEventDisableTiming));
      S360_HIP(hipEventCreateWithFlags(&c->evStripsFree, hipEventDisableTiming));
      if (!c->evPoleSrcFree) S360_HIP(hipEventCreateWithFlags(&c->evPoleSrcFree, hipEventDisableTiming));
    }
    c->pipeline = on != 0;
    c->haveStripsFree = false;
    flow_engines_follow_pipelining(c);  // (the streams are idle: the finish stage's engines change buffer sets)
  });
}
int s360_set_output_double_buffer(s360_ctx* c, int on) {
  return guard(c, [&] {
    need(c, "null ctx");
    c->make_current();
    S360_HIP(hipStreamSynchronize(c->st));
    if (c->st2) S360_HIP(hipStreamSynchronize(c->st2));
    c->two_outputs = on != 0;
  });
}
int s360_set_sharpening(s360_ctx* c, double sharpening) {
  return guard(c, [&] {
    need(c && sharpening >= 0.0, "bad argument");
    c->P.sharpening = sharpening;  // read by the next frame_finish (TRSP:901-915)
  });
}
int s360_set_keep_intermediates(s360_ctx* c, int on) {
  return guard(c, [&] { need(c, "null ctx"); frame_state(c).keep_intermediates = on != 0; });
}

// ---- operator level ------------------------------------------------------------------------------
int s360_compute_optical_flow_batch(s360_ctx* c, const char* alg, int batch, const uint8_t* i0, const uint8_t* i1,
                                    int w, int h, const float* prev_flow, const uint8_t* prev_i0,
                                    const uint8_t* prev_i1, int hint, float* flow_out) {
  return guard(c, [&] {
    need(c && alg && i0 && i1 && flow_out, "null argument");
    need(w >= 4 && h >= 4, "image too small: the reference's bilinear taps need a 2x2 image after the x0.5 entry downscale (PixFlow.h:457-475)");
    need(batch >= 1 && batch <= kMaxFlows, "batch out of range");
    need(hint >= 0 && hint <= 4, "bad direction hint");
    need(!prev_flow || (prev_i0 && prev_i1), "prev_flow given without previous images");
    const PixFlowConsts pc = pixflow_consts_by_name(alg);
    const size_t n = (size_t)w * h, B = batch;
    c->op_a.ensure(2 * B * n * 4);
    c->op_b.ensure(B * n * sizeof(float2));
    h2d(c, c->op_a.p, i0, B * n * 4);
    h2d(c, c->op_a.as<uint8_t>() + B * n * 4, i1, B * n * 4);
    const uchar4* pimg = nullptr;
    const float2* pflow = nullptr;
    if (prev_flow) {
      c->op_c.ensure(2 * B * n * 4);
      c->op_d.ensure(B * n * sizeof(float2));
      h2d(c, c->op_c.p, prev_i0, B * n * 4);
      h2d(c, c->op_c.as<uint8_t>() + B * n * 4, prev_i1, B * n * 4);
      h2d(c, c->op_d.p, prev_flow, B * n * sizeof(float2));
      pimg = c->op_c.as<uchar4>();
      pflow = c->op_d.as<float2>();
    }
    FlowBatch fb;
    fb.add_images(c->op_a.as<uchar4>(), 2 * batch, n);
    if (pimg) fb.add_prev_images(pimg, 2 * batch, n);
    for (int b = 0; b < batch; ++b) fb.add_flow(b, batch + b, c->op_b.as<float2>() + n * b, pflow ? pflow + n * b : nullptr);
    c->flow->compute(c->st, pc, fb, w, h, hint);
    d2h(c, flow_out, c->op_b.p, B * n * sizeof(float2));
  });
}
int s360_compute_optical_flow(s360_ctx* c, const char* alg, const uint8_t* i0, const uint8_t* i1, int w, int h,
                              const float* prev_flow, const uint8_t* prev_i0, const uint8_t* prev_i1, int hint,
                              float* flow_out) {
  return s360_compute_optical_flow_batch(c, alg, 1, i0, i1, w, h, prev_flow, prev_i0, prev_i1, hint, flow_out);
}
// test tap: flow after every pyramid level (coarsest first) of the last single-pair call configuration
int s360_debug_flow_levels(s360_ctx* c, const char* alg, const uint8_t* i0, const uint8_t* i1, int w, int h, int hint,
                           float* levels_out, size_t cap_floats, int* n_levels) {
  return guard(c, [&] {
    need(c && alg && i0 && i1 && levels_out, "null argument");
    const PixFlowConsts pc = pixflow_consts_by_name(alg);
    const size_t n = (size_t)w * h;
    c->op_a.ensure(2 * n * 4);
    c->op_b.ensure(n * sizeof(float2));
    h2d(c, c->op_a.p, i0, n * 4);
    h2d(c, c->op_a.as<uint8_t>() + n * 4, i1, n * 4);
    FlowBatch fb;
    fb.add_images(c->op_a.as<uchar4>(), 2, n);
    fb.add_flow(0, 1, c->op_b.as<float2>());
    std::vector<std::vector<float>> lv;
    c->flow->capture_levels = &lv;
    try {
      c->flow->compute(c->st, pc, fb, w, h, hint);
    } catch (...) {
      c->flow->capture_levels = nullptr;
      throw;
    }
    c->flow->capture_levels = nullptr;
    S360_HIP(hipStreamSynchronize(c->st));
    size_t off = 0;
    for (auto& v : lv) {
      need(off + v.size() <= cap_floats, "levels_out too small");
      std::memcpy(levels_out + off, v.data(), v.size() * sizeof(float));
      off += v.size();
    }
    if (n_levels) *n_levels = (int)lv.size();
  });
}

int s360_spherical_warp_map(s360_ctx* c, float* map_out, int dw, int dh, const s360_camera* cam, float l, float r,
                            float t, float b) {
  return guard(c, [&] {
    need(c && map_out && cam && dw > 0 && dh > 0, "bad argument");
    c->op_a.ensure((size_t)dw * dh * sizeof(float2));
    build_spherical_map(c, c->op_a.as<float2>(), dw, dh, *cam, l, r, t, b);
    d2h(c, map_out, c->op_a.p, (size_t)dw * dh * sizeof(float2));
  });
}
int s360_bicubic_remap_to_spherical(s360_ctx* c, uint8_t* dst, int dw, int dh, int dc, const uint8_t* src, int sw,
                                    int sh, int sc, const s360_camera* cam, float l, float r, float t, float b) {
  return guard(c, [&] {
    need(c && dst && src && cam, "null argument");
    need((dc == 3 || dc == 4) && (sc == 3 || sc == 4), "channels must be 3 or 4");
    need(!(sc == 4 && dc == 3), "4-channel source: the reference's remap() re-creates the destination with the source's type (ImageWarper.cpp:173) - pass a 4-channel destination");
    FrameState& F = frame_state(c);
    const size_t sn = (size_t)sw * sh, dn = (size_t)dw * dh;
    c->op_a.ensure(dn * sizeof(float2));
    c->op_b.ensure(sn * 4);
    c->op_c.ensure(sn * sizeof(uchar4));
    c->op_d.ensure(dn * sizeof(uchar4));
    build_spherical_map(c, c->op_a.as<float2>(), dw, dh, *cam, l, r, t, b);
    h2d(c, c->op_b.p, src, sn * sc);
    launch_bgr_to_bgra(c->st, c->op_b.as<uint8_t>(), sc, c->op_c.as<uchar4>(), sn);
    launch_remap_cubic_u8c4(c->st, c->op_c.as<uchar4>(), sw, sh, c->op_a.as<float2>(), c->op_d.as<uchar4>(), dw, dh,
                            F.tab.dev, 0, 0, 1);
    if (dc == 4) {
      d2h(c, dst, c->op_d.p, dn * 4);
    } else {
      c->op_e.ensure(dn * 3);
      launch_pack_bgr(c->st, c->op_d.as<uchar4>(), dw, dh, c->op_e.as<uint8_t>());
      d2h(c, dst, c->op_e.p, dn * 3);
    }
  });
}
int s360_combine_lazy_novel_views(s360_ctx* c, const uint8_t* image_l, const uint8_t* image_r, const float* flow_l_to_r,
                                  const float* flow_r_to_l, uint8_t* chunk_l, uint8_t* chunk_r) {
  return guard(c, [&] {
    need(c && image_l && image_r && flow_l_to_r && flow_r_to_l && chunk_l && chunk_r, "null argument");
    FrameState& F = frame_state(c);
    const s360_geometry& g = c->g;
    const int P = F.P, ow = g.overlap_image_width, camH = g.cam_image_height, stripW = c->P.eqr_width / P;
    const size_t on = (size_t)ow * camH, sn = (size_t)stripW * camH;
    c->op_a.ensure(2 * on * 4);
    c->op_b.ensure(2 * on * sizeof(float2));
    c->op_c.ensure(2 * sn * 4);
    h2d(c, c->op_a.p, image_l, on * 4);
    h2d(c, c->op_a.as<uint8_t>() + on * 4, image_r, on * 4);
    h2d(c, c->op_b.p, flow_l_to_r, on * sizeof(float2));
    h2d(c, c->op_b.as<float2>() + on, flow_r_to_l, on * sizeof(float2));
    NovelViewParams nv;
    nv.overlapW = ow; nv.camH = camH; nv.stripW = stripW; nv.numNovelViews = g.num_novel_views;
    nv.numPairs = 1; nv.numLocal = 1;
    nv.camImageWidthHalf = float(g.cam_image_width) * 0.5f;
    nv.disp = g.verge_at_infinity_slab_displacement;
    launch_novel_view(c->st, c->op_a.as<uchar4>(), c->op_b.as<float2>(), c->op_c.as<uchar4>(), nv, 0, 1, F.tab.dev);
    d2h(c, chunk_l, c->op_c.p, sn * 4);
    d2h(c, chunk_r, c->op_c.as<uint8_t>() + sn * 4, sn * 4);
  });
}
int s360_flatten_layers_deghost_prefer_base(s360_ctx* c, const uint8_t* bottom_layer, const uint8_t* top_layer, int w,
                                            int h, uint8_t* out) {
  return guard(c, [&] {
    need(c && bottom_layer && top_layer && out && w > 0 && h > 0, "bad argument");
    FrameState& F = frame_state(c);
    const size_t n = (size_t)w * h;
    c->op_a.ensure(n * 4); c->op_b.ensure(n * 4); c->op_c.ensure(n * 4);
    h2d(c, c->op_a.p, bottom_layer, n * 4);
    h2d(c, c->op_b.p, top_layer, n * 4);
    launch_flatten(c->st, c->op_a.as<uchar4>(), c->op_b.as<uchar4>(), c->op_c.as<uchar4>(), w, h, 0, F.tab.dev);
    d2h(c, out, c->op_c.p, n * 4);
  });
}
int s360_offset_horizontal_wrap(s360_ctx* c, const uint8_t* src, int w, int h, int channels, float offset,
                                uint8_t* out) {
  return guard(c, [&] {
    need(c && src && out && w > 0 && h > 0, "bad argument");
    need(channels == 3 || channels == 4, "channels must be 3 or 4");
    const size_t n = (size_t)w * h;
    c->op_a.ensure(n * 4); c->op_b.ensure(n * 4);
    if (channels == 4) {
      h2d(c, c->op_a.p, src, n * 4);
    } else {  // BGR: through the BGRA kernels, alpha dropped again
      c->op_c.ensure(n * 3);
      h2d(c, c->op_c.p, src, n * 3);
      launch_bgr_to_bgra(c->st, c->op_c.as<uint8_t>(), 3, c->op_a.as<uchar4>(), n);
    }
    // one "strip" of the full width, no padding: pure offsetHorizontalWrap
    launch_assemble_pano(c->st, c->op_a.as<uchar4>(), 1, h, w, offset, c->op_b.as<uchar4>(), w, h);
    if (channels == 4) {
      d2h(c, out, c->op_b.p, n * 4);
    } else {
      launch_pack_bgr(c->st, c->op_b.as<uchar4>(), w, h, c->op_c.as<uint8_t>());
      d2h(c, out, c->op_c.p, n * 3);
    }
  });
}
int s360_feather_alpha_channel(s360_ctx* c, const uint8_t* src, int w, int h, int erode_size, uint8_t* out) {
  return guard(c, [&] {
    need(c && src && out && w > 0 && h > 0, "bad argument");
    need(erode_size >= 1 && erode_size <= 32, "erode_size must be in 1..32");
    need(erode_size % 2 == 1, "erode_size must be odd (it is the GaussianBlur kernel size, CvUtil.cpp:151)");
    const size_t n = (size_t)w * h;
    c->op_a.ensure(n * 4); c->op_b.ensure(n * 4);
    h2d(c, c->op_a.p, src, n * 4);
    if (erode_size == c->P.std_alpha_feather_size) {
      dev_feather_alpha_to_ext(c, c->op_a.as<uchar4>(), w, h, c->op_b.as<uchar4>(), w);
    } else {
      const std::vector<int> taps = feather_gauss_taps(erode_size);
      c->op_e.ensure(taps.size() * sizeof(int));
      h2d(c, c->op_e.p, taps.data(), taps.size() * sizeof(int));
      dev_feather_alpha_to_ext(c, c->op_a.as<uchar4>(), w, h, c->op_b.as<uchar4>(), w, erode_size, c->op_e.as<int>());
    }
    d2h(c, out, c->op_b.p, n * 4);
  });
}
int s360_pole_to_side_flow(s360_ctx* c, const uint8_t* side, const uint8_t* pole, int pole_rows, uint8_t* warped_out,
                           float* flow_out) {
  return guard(c, [&] {
    need(c && side && pole && warped_out && pole_rows > 0, "bad argument");
    need(c->bottom_idx >= 0, "rig has no bottom camera (pole ramp uses its fov, TRSP:461)");
    const int W = c->P.eqr_width, H = c->P.eqr_height;
    need(pole_rows <= H, "pole_rows larger than eqr_height");
    const int extW = int(float(W) * 1.2f);
    const size_t en = (size_t)W * H, pn = (size_t)W * pole_rows, xn = (size_t)extW * pole_rows;
    c->op_a.ensure(en * 4); c->op_b.ensure(pn * 4); c->op_c.ensure(2 * xn * 4); c->op_d.ensure(xn * sizeof(float2));
    c->op_e.ensure(en * 4);
    h2d(c, c->op_a.p, side, en * 4);
    h2d(c, c->op_b.p, pole, pn * 4);
    uchar4* ext = c->op_c.as<uchar4>();
    dev_feather_alpha_to_ext(c, c->op_a.as<uchar4>(), W, pole_rows, ext, extW);
    launch_extend_wrap(c->st, c->op_b.as<uchar4>(), nullptr, W, pole_rows, ext + xn, extW);
    (void)flow_engine(c, 1);
    FlowBatch fb;
    fb.add_images(ext, 2, xn);
    fb.add_flow(0, 1, c->op_d.as<float2>());
    c->flow_pole->compute(c->st, pixflow_consts_by_name(c->P.polar_flow_alg), fb, extW, pole_rows, S360_HINT_DOWN);
    dev_pole_unit_post(c, ext + xn, c->op_d.as<float2>(), W, pole_rows, extW, c->op_e.as<uchar4>(), H);
    d2h(c, warped_out, c->op_e.p, en * 4);
    if (flow_out) d2h(c, flow_out, c->op_d.p, xn * sizeof(float2));
  });
}
int s360_sharpen(s360_ctx* c, uint8_t* bgr, int w, int h, float sharpening) {
  return guard(c, [&] {
    need(c && bgr && w > 1 && h > 1, "bad argument");
    const size_t n = (size_t)w * h;
    c->op_a.ensure(n * 3); c->op_b.ensure(n * 4); c->op_c.ensure(n * 4); c->op_d.ensure(sharpen_scratch_bytes(w, h));
    h2d(c, c->op_a.p, bgr, n * 3);
    launch_bgr_to_bgra(c->st, c->op_a.as<uint8_t>(), 3, c->op_b.as<uchar4>(), n);
    launch_sharpen(c->st, c->op_b.as<uchar4>(), c->op_c.as<uchar4>(), c->op_d.as<float>(), w, h, 1.0f + sharpening);
    launch_pack_bgr(c->st, c->op_b.as<uchar4>(), w, h, c->op_a.as<uint8_t>());
    d2h(c, bgr, c->op_a.p, n * 3);
  });
}

// ---- frame level ------------------------------------------------------------------------------------
int s360_frame_upload_side(s360_ctx* c, int side_idx, const uint8_t* img, int w, int h, int channels) {
  return frame_guard(c, [&] { need(c && img && w > 0 && h > 0, "bad argument"); frame_upload_side(c, side_idx, img, w, h, channels); });
}
int s360_frame_upload_top(s360_ctx* c, const uint8_t* bgr, int w, int h) {
  return frame_guard(c, [&] { need(c && bgr && w > 0 && h > 0, "bad argument"); frame_upload_pole(c, true, bgr, w, h); });
}
int s360_frame_upload_bottom(s360_ctx* c, const uint8_t* bgr, int w, int h) {
  return frame_guard(c, [&] { need(c && bgr && w > 0 && h > 0, "bad argument"); frame_upload_pole(c, false, bgr, w, h); });
}
int s360_frame_upload_raw(s360_ctx* c, s360_isp* isp, int camera, const uint16_t* raw16, int w, int h) {
  return frame_guard(c, [&] { need(c && isp && raw16 && w > 0 && h > 0, "bad argument"); frame_upload_raw(c, isp, camera, raw16, 16, w, h); });
}
int s360_frame_upload_packed(s360_ctx* c, s360_isp* isp, int camera, const uint8_t* frame, int bits, int w, int h) {
  return frame_guard(c, [&] {
    need(c && isp && frame && w > 0 && h > 0, "bad argument");
    need(bits == 8 || bits == 12, "packed frames are 8 or 12 bits per pixel");
    frame_upload_raw(c, isp, camera, frame, bits, w, h);
  });
}
int s360_frame_upload_pole_removal(s360_ctx* c, const uint8_t* bottom2, const uint8_t* mask, const uint8_t* mask2, int w,
                                   int h) {
  return frame_guard(c, [&] {
    need(c && bottom2 && mask && mask2 && w > 0 && h > 0, "bad argument");
    frame_upload_pole_removal(c, bottom2, mask, mask2, w, h);
  });
}
int s360_frame_set_prev_pole_removal(s360_ctx* c, const float* flow, const uint8_t* bottom_image, const uint8_t* bottom_image2,
                                     int w, int h) {
  return frame_guard(c, [&] {
    need(c && flow && bottom_image && bottom_image2 && w > 0 && h > 0, "bad argument");
    FrameState& F = frame_state(c);
    const size_t n = (size_t)w * h;
    const int prv = F.last_pr;  // becomes "the previous frame" of the next render
    F.prImgs[prv].ensure(2 * n * sizeof(uchar4));
    F.prFlow[prv].ensure(n * sizeof(float2));
    h2d(c, F.prImgs[prv].as<uchar4>(), bottom_image, n * sizeof(uchar4));
    h2d(c, F.prImgs[prv].as<uchar4>() + n, bottom_image2, n * sizeof(uchar4));
    h2d(c, F.prFlow[prv].p, flow, n * sizeof(float2));
    S360_HIP(hipStreamSynchronize(c->st));
    F.have_prev_pr = true;
  });
}
int s360_frame_render_pairs(s360_ctx* c, int p0, int p1, int use_prev) {
  return frame_guard(c, [&] { need(c, "null ctx"); frame_render_pairs(c, p0, p1, use_prev); });
}
int s360_frame_finish(s360_ctx* c, int pole_mask, int use_prev) {
  return frame_guard(c, [&] { need(c, "null ctx"); frame_finish(c, pole_mask, use_prev); });
}
int s360_frame_render(s360_ctx* c, int use_prev) {
  return frame_guard(c, [&] {
    need(c, "null ctx");
    frame_render_pairs(c, 0, (int)c->rig.side.size(), use_prev);
    frame_finish(c, 15, use_prev);
  });
}
int s360_set_frame_slots(s360_ctx* c, int n) {
  return frame_guard(c, [&] { need(c, "null ctx"); set_frame_slots(c, n); });
}
int s360_select_frame_slot(s360_ctx* c, int k) {
  return frame_guard(c, [&] {
    need(c, "null ctx");
    need(k >= 0 && k < (int)std::max<size_t>(c->slots.size(), 1), "frame slot out of range");
    c->slot = k;
  });
}
int s360_frame_render_batch(s360_ctx* c, int use_prev) {
  return frame_guard(c, [&] { need(c, "null ctx"); frame_render_batch(c, use_prev); });
}
int s360_frame_render_slots(s360_ctx* c, const int* slots, int n, int use_prev) {
  return frame_guard(c, [&] {
    need(c && slots && n > 0, "bad argument");
    frame_render_slots(c, slots, n, use_prev);
  });
}
int s360_frame_set_prev_side(s360_ctx* c, int pair, const float* flow_l_to_r, const float* flow_r_to_l,
                             const uint8_t* overlap_l, const uint8_t* overlap_r) {
  return frame_guard(c, [&] {
    need(c && flow_l_to_r && flow_r_to_l && overlap_l && overlap_r, "bad argument");
    FrameState& F = frame_state(c);
    const int P = F.P;
    need(pair >= 0 && pair < P, "pair_idx out of range");
    const size_t on = (size_t)c->g.overlap_image_width * c->g.cam_image_height;
    // previous-frame slot of the double buffer, laid out like frame_render_pairs lays out the block [p0, p1) this
    // context renders (s360_frame_set_partition; default all pairs): L images / LtoR flows of the block first, then
    // R images / RtoL flows
    if (!F.partition_declared) { F.side_p0 = 0; F.side_p1 = P; }
    const int p0 = F.side_p0, n = F.side_p1 - F.side_p0;
    need(pair >= p0 && pair < F.side_p1, "pair_idx outside the block declared with s360_frame_set_partition");
    const int j = pair - p0;
    const int prv = F.last_side;  // becomes "the previous frame" of the next render
    F.overlaps[prv].ensure(2 * n * on * sizeof(uchar4));
    F.sideFlows.ensure(2 * n * on * sizeof(float2));
    h2d(c, F.overlaps[prv].as<uchar4>() + on * j, overlap_l, on * sizeof(uchar4));
    h2d(c, F.overlaps[prv].as<uchar4>() + on * (n + j), overlap_r, on * sizeof(uchar4));
    h2d(c, F.sideFlows.as<float2>() + on * j, flow_l_to_r, on * sizeof(float2));
    h2d(c, F.sideFlows.as<float2>() + on * (n + j), flow_r_to_l, on * sizeof(float2));
    S360_HIP(hipStreamSynchronize(c->st));  // the caller's buffers may be reused as soon as this returns
    F.have_prev_side = true;
  });
}
int s360_frame_set_prev_pole(s360_ctx* c, int unit, const float* flow, const uint8_t* ext_side,
                             const uint8_t* ext_fisheye) {
  return frame_guard(c, [&] {
    need(c && flow && ext_side && ext_fisheye && unit >= 0 && unit < 4, "bad argument");
    FrameState& F = frame_state(c);
    const int extW = int(float(c->P.eqr_width) * 1.2f);
    const int rows = unit < 2 ? c->g.top_rows : c->g.bottom_rows;
    const size_t xn = (size_t)extW * rows;                                       // one image / flow of this unit
    const size_t xs = (size_t)extW * std::max(c->g.top_rows, c->g.bottom_rows);  // slot stride (frame_finish)
    const int prv = F.last_pole;  // becomes "the previous frame" of the next render
    F.extImgs[prv].ensure(6 * xs * sizeof(uchar4));
    F.poleFlows.ensure(4 * xs * sizeof(float2));
    h2d(c, F.extImgs[prv].as<uchar4>() + xs * unit, ext_side, xn * sizeof(uchar4));
    h2d(c, F.extImgs[prv].as<uchar4>() + xs * (unit < 2 ? 4 : 5), ext_fisheye, xn * sizeof(uchar4));
    h2d(c, F.poleFlows.as<float2>() + xs * unit, flow, xn * sizeof(float2));
    S360_HIP(hipStreamSynchronize(c->st));  // the caller's buffers may be reused as soon as this returns
    F.extW = extW;
    F.extStride = xs;
    F.poleRowsT = c->g.top_rows;
    F.poleRowsB = c->g.bottom_rows;
    F.have_prev_pole = true;
  });
}
int s360_comm_get_unique_id(void* id_out) {
  return guard(nullptr, [&] { need(id_out, "null argument"); comm_unique_id(id_out); });
}
const char* s360_comm_library_path(void) {
  const char* p = nullptr;
  (void)guard(nullptr, [&] { p = comm_library_path(); });
  return p;
}
int s360_comm_init_rank(s360_ctx* c, const void* id, int rank, int nranks) {
  return guard(c, [&] { need(c && id, "null argument"); comm_init_rank(c, id, rank, nranks); });
}
int s360_comm_init_all(s360_ctx* const* ctxs, int n) {
  return guard(nullptr, [&] { need(ctxs && n > 0, "bad argument"); comm_init_all(ctxs, n); });
}
int s360_comm_destroy(s360_ctx* c) {
  return guard(c, [&] { need(c, "null ctx"); S360_HIP(hipStreamSynchronize(c->st)); comm_destroy(c); });
}
int s360_comm_size(s360_ctx* c) {
  int n = -1;
  (void)guard(c, [&] { need(c, "null ctx"); n = comm_size(c); });
  return n;
}
int s360_comm_rank(s360_ctx* c) {
  int r = -1;
  (void)guard(c, [&] { need(c, "null ctx"); r = comm_rank(c); });
  return r;
}
int s360_comm_stats(s360_ctx* c, int which, unsigned long long out[3]) {
  return guard(c, [&] {
    need(c && out, "null argument");
    need(which == 0 || which == 1, "which: 0 = strips exchange, 1 = pole-layer gather");
    out[0] = c->comm_stats[which].calls;
    out[1] = c->comm_stats[which].sent;
    out[2] = c->comm_stats[which].received;
  });
}
int s360_frame_gather_strips(s360_ctx* c, const int* bounds, int root) {
  return frame_guard(c, [&] { need(c && bounds, "null argument"); no_pipelining(c); frame_gather_strips(c, bounds, root); });
}
int s360_frame_exchange_strips(s360_ctx* c, const int* bounds, const int* need_mask) {
  return frame_guard(c, [&] { need(c && bounds && need_mask, "null argument"); no_pipelining(c); frame_exchange_strips(c, bounds, need_mask); });
}
int s360_frame_gather_pole_layers(s360_ctx* c, const int owner[4], int root) {
  return frame_guard(c, [&] { need(c && owner, "null argument"); no_pipelining(c); frame_gather_pole_layers(c, owner, root); });
}
int s360_frame_pole_units(s360_ctx* c, int pole_mask, int use_prev) {
  return frame_guard(c, [&] { need(c, "null ctx"); no_pipelining(c); frame_pole_units(c, pole_mask, use_prev); });
}
int s360_frame_composite(s360_ctx* c, int pole_mask) {
  return frame_guard(c, [&] { need(c, "null ctx"); no_pipelining(c); frame_composite(c, pole_mask); });
}
int s360_comm_loopback(s360_ctx* c, int src_pair, int dst_pair) {
  return guard(c, [&] { need(c, "null ctx"); comm_loopback(c, src_pair, dst_pair); });
}
int s360_frame_set_partition(s360_ctx* c, int p0, int p1) {
  return frame_guard(c, [&] {
    need(c, "null ctx");
    FrameState& F = frame_state(c);
    need(p0 >= 0 && p1 <= F.P && p0 <= p1, "bad pair range");
    if (F.side_p0 != p0 || F.side_p1 != p1) F.have_prev_side = false;  // state of another block is of no use
    F.side_p0 = p0;
    F.side_p1 = p1;
    F.partition_declared = true;
  });
}
int s360_frame_strip_ptr(s360_ctx* c, int eye, void** dev_ptr, size_t* bytes_per_pair) {
  return frame_guard(c, [&] {
    need(c && dev_ptr && (eye == 0 || eye == 1), "bad argument");
    FrameState& F = frame_state(c);
    const int P = F.P, camH = c->g.cam_image_height, stripW = c->P.eqr_width / P;
    const size_t per = (size_t)camH * stripW * sizeof(uchar4);
    F.strips.ensure(2 * P * per);
    *dev_ptr = F.strips.as<uint8_t>() + (size_t)eye * P * per;
    if (bytes_per_pair) *bytes_per_pair = per;
  });
}
int s360_frame_equirect_dev(s360_ctx* c, void** dev_ptr, size_t* bytes) {
  return frame_guard(c, [&] {
    need(c && dev_ptr, "bad argument");
    FrameState& F = frame_state(c);
    need(F.frames_done > 0, "no frame rendered yet");
    *dev_ptr = F.outBGR[F.out_cur].p;
    if (bytes) *bytes = (size_t)c->g.out_width * c->g.out_height * 3;
  });
}
int s360_frame_download_equirect(s360_ctx* c, uint8_t* out_bgr) {
  return frame_guard(c, [&] {
    need(c && out_bgr, "bad argument");
    FrameState& F = frame_state(c);
    need(F.frames_done > 0, "no frame rendered yet");
    d2h(c, out_bgr, F.outBGR[F.out_cur].p, (size_t)c->g.out_width * c->g.out_height * 3);
  });
}
// the state of frame slot `slot` (-1: the selected one) without touching the selection: a fetching thread names its slot, so that
// the thread that feeds the context can go on selecting slots for its uploads
static FrameState& slot_state(s360_ctx* c, int slot) {
  if (slot < 0) return frame_state(c);
  need(slot < (int)std::max<size_t>(c->slots.size(), 1), "frame slot out of range");
  const int saved = c->slot;
  c->slot = slot;
  struct Restore { s360_ctx* c; int s; ~Restore() { c->slot = s; } } restore{c, saved};
  return frame_state(c);
}
static int download_equirect_impl(s360_ctx* c, int slot, int age, uint8_t* out_bgr) {
  if (!c) return S360_ERR_INVALID_ARG;
  return guard_l(c, [&](std::unique_lock<std::recursive_mutex>& lk) {
    need(out_bgr && (age == 0 || age == 1), "bad argument (age is 0 = latest enqueued frame or 1 = the one before)");
    FrameState& F = slot_state(c, slot);
    need(F.frames_done > age, "that frame has not been rendered");
    need(age == 0 || ((c->pipeline || c->two_outputs) && F.outBGR[F.out_cur ^ 1].p),
         "age 1 needs two output buffers (s360_set_frame_pipelining or s360_set_output_double_buffer)");
    const int b = age == 0 ? F.out_cur : F.out_cur ^ 1;
    // wait for THAT frame only (its event sits behind its last kernel), then copy on a stream of its own so that the
    // transfer does not queue behind the kernels of the frame enqueued after it
    if (!c->stDown) S360_HIP(hipStreamCreateWithFlags(&c->stDown, hipStreamNonBlocking));
    if (!c->evDown) S360_HIP(hipEventCreateWithFlags(&c->evDown, hipEventDisableTiming | hipEventBlockingSync));
    if (!c->downErr) {
      S360_HIP(hipHostMalloc((void**)&c->downErr, 4 * sizeof(unsigned), hipHostMallocDefault));
      std::memset(c->downErr, 0, 4 * sizeof(unsigned));
    }
    if (!F.downRead[b]) S360_HIP(hipEventCreateWithFlags(&F.downRead[b], hipEventDisableTiming));
    S360_HIP(hipStreamWaitEvent(c->stDown, F.outDone[b], 0));
    S360_HIP(hipMemcpyAsync(out_bgr, F.outBGR[b].p, (size_t)c->g.out_width * c->g.out_height * 3, hipMemcpyDeviceToHost, c->stDown));
    // the frame's error words travel with it (a later frame's finish stage rewrites them as soon as downRead[b] has fired): a device
    // -> page-locked host copy on the download's own stream (a host -> host "async" copy would block this thread with the lock held)
    if (F.outErrDev[b].p) S360_HIP(hipMemcpyAsync(c->downErr, F.outErrDev[b].p, 3 * sizeof(unsigned), hipMemcpyDeviceToHost, c->stDown));
    S360_HIP(hipEventRecord(F.downRead[b], c->stDown));
    S360_HIP(hipEventRecord(c->evDown, c->stDown));
    const hipEvent_t ev = c->evDown;
    const unsigned* errw = F.outErrDev[b].p ? c->downErr : nullptr;  // (nothing of the slot is touched after the lock is back: it may be gone)
    // The wait is most of a frame long: the context is free meanwhile (the next frame's uploads and enqueue need it).
    // One fetching thread per context: evDown is re-recorded by the next call.
    lk.unlock();
    const hipError_t rc = hipEventSynchronize(ev);
    lk.lock();
    S360_HIP(rc);
    // the frame's sweep error words were snapshotted in front of outDone[b] (render.hpp): a timed-out banded sweep
    // fails THIS frame's download, before the host hands the pixels to an encoder
    if (errw && (errw[0] | errw[1] | errw[2])) {
      // reported once: the engines' cumulative error words are reset, so that the frames enqueued from now on are judged on
      // their own sweeps (frames already in flight carry the snapshot they took)
      for (FlowEngine* e : {c->flow.get(), c->flow_pole.get(), c->flow_pr.get()})
        if (e) (void)e->take_error(c->st);
      throw Error(S360_ERR_HIP, "banded sweep timed out waiting for a neighbour band (results invalid)");
    }
  });
}
int s360_frame_download_equirect_of(s360_ctx* c, int age, uint8_t* out_bgr) { return download_equirect_impl(c, -1, age, out_bgr); }
int s360_frame_download_equirect_slot(s360_ctx* c, int slot, int age, uint8_t* out_bgr) {
  if (slot < 0) return S360_ERR_INVALID_ARG;
  return download_equirect_impl(c, slot, age, out_bgr);
}
/* ---- the equirect as a PNG file, encoded on the device (png.hip; replaces imwriteExceptionOnFail, TRSP:938-961) ---- */
static int png_crc_threads() {
  const char* e = std::getenv("S360_PNG_CRC_THREADS");
  const int n = e ? std::atoi(e) : 4;
  return n < 1 ? 1 : (n > 64 ? 64 : n);
}
int s360_set_png_encode(s360_ctx* c, int on) {
  return guard(c, [&] { need(c, "null ctx"); c->png_encode = on != 0; });
}
size_t s360_png_bound(int w, int h) {
  size_t n = 0;
  (void)guard(nullptr, [&] { n = PngPlan::make(w, h).file_bound; });
  return n;
}
size_t s360_frame_png_bound(s360_ctx* c) {
  size_t n = 0;
  if (!c) return 0;
  (void)guard(c, [&] { n = PngPlan::make(c->g.out_width, c->g.out_height).file_bound; });
  return n;
}
// band table to the host (event wait with the context released), then the file image's bytes, then the CRCs off the lock
static void png_fetch(s360_ctx* c, std::unique_lock<std::recursive_mutex>& lk, const PngPlan& plan, const DevBuf& meta, const DevBuf& file,
                      hipEvent_t after /* nullable: what the copy waits for */, hipEvent_t readDone /* nullable: recorded behind the copies */,
                      const void* errDev, uint8_t* out, size_t cap, size_t* n_out, bool release = true) {
  if (!c->stDown) S360_HIP(hipStreamCreateWithFlags(&c->stDown, hipStreamNonBlocking));
  if (!c->evDown) S360_HIP(hipEventCreateWithFlags(&c->evDown, hipEventDisableTiming | hipEventBlockingSync));
  if (!c->downErr) {
    S360_HIP(hipHostMalloc((void**)&c->downErr, 4 * sizeof(unsigned), hipHostMallocDefault));
    std::memset(c->downErr, 0, 4 * sizeof(unsigned));
  }
  const size_t mbytes = ((size_t)plan.nbands + 1) * sizeof(PngBandMeta);
  if (c->pngMetaHostBytes < mbytes) {
    if (c->pngMetaHost) (void)hipHostFree(c->pngMetaHost);
    c->pngMetaHost = nullptr;
    c->pngMetaHostBytes = 0;
    S360_HIP(hipHostMalloc(&c->pngMetaHost, mbytes, hipHostMallocDefault));
    c->pngMetaHostBytes = mbytes;
  }
  if (after) S360_HIP(hipStreamWaitEvent(c->stDown, after, 0));
  S360_HIP(hipMemcpyAsync(c->pngMetaHost, meta.p, mbytes, hipMemcpyDeviceToHost, c->stDown));
  if (errDev) S360_HIP(hipMemcpyAsync(c->downErr, errDev, 3 * sizeof(unsigned), hipMemcpyDeviceToHost, c->stDown));
  S360_HIP(hipEventRecord(c->evDown, c->stDown));
  hipEvent_t ev = c->evDown;
  if (release) lk.unlock();
  hipError_t rc = hipEventSynchronize(ev);
  if (release) lk.lock();
  S360_HIP(rc);
  std::vector<PngBandMeta> m((size_t)plan.nbands + 1);
  std::memcpy(m.data(), c->pngMetaHost, mbytes);
  const unsigned errw = errDev ? (c->downErr[0] | c->downErr[1] | c->downErr[2]) : 0u;
  const size_t end_bands = (size_t)m[plan.nbands].file_off;
  if (end_bands > plan.file_bound || end_bands + 28 > cap) {
    if (readDone) S360_HIP(hipEventRecord(readDone, c->stDown));
    throw Error(S360_ERR_INVALID_ARG, "png: output buffer too small (s360_frame_png_bound / s360_png_bound give the size to allocate)");
  }
  S360_HIP(hipMemcpyAsync(out, file.p, end_bands, hipMemcpyDeviceToHost, c->stDown));
  if (readDone) S360_HIP(hipEventRecord(readDone, c->stDown));
  S360_HIP(hipEventRecord(c->evDown, c->stDown));
  ev = c->evDown;
  if (release) lk.unlock();
  rc = hipEventSynchronize(ev);
  if (rc != hipSuccess) { if (release) lk.lock(); S360_HIP(rc); }
  if (errw) {
    if (release) lk.lock();
    for (FlowEngine* e : {c->flow.get(), c->flow_pole.get(), c->flow_pr.get()})
      if (e) (void)e->take_error(c->st);
    throw Error(S360_ERR_HIP, "banded sweep timed out waiting for a neighbour band (results invalid)");
  }
  // the file's frame around the bands and the chunks' CRCs: host work on the caller's buffer, the context stays free meanwhile
  const size_t n = png_finish_host(out, cap, plan, m.data(), png_crc_threads());
  if (release) lk.lock();
  *n_out = n;
}
static int download_png_impl(s360_ctx* c, int slot, int age, uint8_t* out, size_t cap, size_t* n_out) {
  if (!c) return S360_ERR_INVALID_ARG;
  return guard_l(c, [&](std::unique_lock<std::recursive_mutex>& lk) {
    need(out && n_out && (age == 0 || age == 1), "bad argument (age is 0 = latest enqueued frame or 1 = the one before)");
    FrameState& F = slot_state(c, slot);
    need(F.frames_done > age, "that frame has not been rendered");
    need(age == 0 || ((c->pipeline || c->two_outputs) && F.outBGR[F.out_cur ^ 1].p),
         "age 1 needs two output buffers (s360_set_frame_pipelining or s360_set_output_double_buffer)");
    const int b = age == 0 ? F.out_cur : F.out_cur ^ 1;
    if (F.pngFrame[b] != F.frames_done - 1 - age || !F.pngFile[b].p)
      throw Error(S360_ERR_STATE, "that frame was rendered without s360_set_png_encode");
    if (!F.downRead[b]) S360_HIP(hipEventCreateWithFlags(&F.downRead[b], hipEventDisableTiming));
    png_fetch(c, lk, F.pngPlan[b], F.pngMeta[b], F.pngFile[b], F.outDone[b], F.downRead[b], F.outErrDev[b].p, out, cap, n_out);
  });
}
int s360_frame_download_png(s360_ctx* c, int age, uint8_t* out, size_t cap, size_t* n_out) { return download_png_impl(c, -1, age, out, cap, n_out); }
int s360_frame_download_png_slot(s360_ctx* c, int slot, int age, uint8_t* out, size_t cap, size_t* n_out) {
  if (slot < 0) return S360_ERR_INVALID_ARG;
  return download_png_impl(c, slot, age, out, cap, n_out);
}
int s360_encode_png(s360_ctx* c, const uint8_t* bgr, int w, int h, uint8_t* out, size_t cap, size_t* n_out) {
  if (!c) return S360_ERR_INVALID_ARG;
  return guard_l(c, [&](std::unique_lock<std::recursive_mutex>& lk) {
    need(bgr && out && n_out && w > 0 && h > 0, "bad argument");
    const PngPlan plan = PngPlan::make(w, h);
    const size_t nb = (size_t)w * h * 3;
    c->op_a.ensure((nb + 3) & ~(size_t)3);
    c->op_b.ensure(plan.file_bound);
    S360_HIP(hipMemcpyAsync(c->op_a.p, bgr, nb, hipMemcpyHostToDevice, c->st));
    png_encode_enqueue(c->st, c->op_a.as<uint8_t>(), plan, c->op_c, c->op_d, c->op_b.as<uint8_t>());
    S360_HIP(hipStreamSynchronize(c->st));
    png_fetch(c, lk, plan, c->op_d, c->op_b, nullptr, nullptr, nullptr, out, cap, n_out, false);  // (the operator scratch stays locked)
  });
}
/* ---- page-locked host buffers for streaming hosts ---- */
void* s360_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) {
    (void)hipGetLastError();
    g_err = "s360_host_alloc: hipHostMalloc failed";
    return nullptr;
  }
  std::lock_guard<std::mutex> lk(g_pinMu);
  g_pinBlocks.emplace_back(static_cast<const char*>(p), bytes);
  return p;
}
void s360_host_free(void* p) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lk(g_pinMu);
    for (size_t i = 0; i < g_pinBlocks.size(); ++i)
      if (g_pinBlocks[i].first == p) { g_pinBlocks.erase(g_pinBlocks.begin() + i); break; }
  }
  (void)hipHostFree(p);
}
int s360_frame_uploads_complete(s360_ctx* c) {
  if (!c) return S360_ERR_INVALID_ARG;
  return guard_l(c, [&](std::unique_lock<std::recursive_mutex>& lk) {
    if (!c->haveUploaded) return;
    if (!c->evUpHost) S360_HIP(hipEventCreateWithFlags(&c->evUpHost, hipEventDisableTiming | hipEventBlockingSync));
    S360_HIP(hipEventRecord(c->evUpHost, c->stUp));
    const hipEvent_t ev = c->evUpHost;
    lk.unlock();
    const hipError_t rc = hipEventSynchronize(ev);
    lk.lock();
    S360_HIP(rc);
  });
}

int s360_frame_cubemap(s360_ctx* c, int face_w, int face_h, const char* format, int whc[3], uint8_t* out_bgr) {
  return frame_guard(c, [&] {
    need(c && format && whc, "bad argument");
    const std::string f(format);
    if (f != "video" && f != "photo")  // CvUtil.cpp:134-137
      throw Error(S360_ERR_INVALID_ARG, "unexpected cubemap format: " + f + ". valid formats are: video,photo");
    need(face_w > 0 && face_h > 0 && face_w <= 16384 && face_h <= 16384, "cubemap face size must be in 1..16384");
    whc[0] = f == "video" ? 3 * face_w : face_w;
    whc[1] = f == "video" ? 4 * face_h : 12 * face_h;
    whc[2] = 3;
    if (!out_bgr) return;
    int ow = 0, oh = 0;
    wait_finish_stream(c);  // the cubemap kernel reads the composited panoramas
    frame_cubemap(c, face_w, face_h, f == "video", &ow, &oh);
    d2h(c, out_bgr, frame_state(c).cubeOut.p, (size_t)ow * oh * 3);
  });
}
int s360_frame_get_u8(s360_ctx* c, const char* name, int idx, int whc[3], uint8_t* dst) {
  return frame_guard(c, [&] {
    need(c && name && whc, "bad argument");
    FrameState& F = frame_state(c);
    const s360_geometry& g = c->g;
    const int W = c->P.eqr_width, H = c->P.eqr_height, P = F.P;
    const std::string n(name);
    const void* src = nullptr;
    int w = 0, h = 0, ch = 4;
    const int nloc = F.side_p1 - F.side_p0;
    if (n == "projection") {
      need(idx >= 0 && idx < P && F.sc->proj.p, "projection not available");
      w = g.cam_image_width; h = g.cam_image_height;
      src = F.sc->proj.as<uchar4>() + (size_t)w * h * idx;
    } else if (n == "overlap_l" || n == "overlap_r") {
      need(idx >= F.side_p0 && idx < F.side_p1 && F.overlaps[F.last_side].p, "overlap not available");
      w = g.overlap_image_width; h = g.cam_image_height;
      const int j = idx - F.side_p0 + (n == "overlap_r" ? nloc : 0);
      src = F.overlaps[F.last_side].as<uchar4>() + (size_t)w * h * j;
    } else if (n == "side_pano_l" || n == "side_pano_r") {
      const int e = n == "side_pano_r";
      need(F.panoDbg[e].p, "enable keep_intermediates before rendering");
      w = W; h = H; src = F.panoDbg[e].p;
    } else if (n == "top_spherical") {
      need(F.sc->topSph.p, "not available"); w = W; h = g.top_rows; src = F.sc->topSph.p;
    } else if (n == "bottom_spherical") {
      need(F.sc->botSph.p, "not available"); w = W; h = g.bottom_rows; src = F.sc->botSph.p;
    } else if (n == "pole_warped") {
      need(idx >= 0 && idx < 4 && F.sc->poleWarped[idx].p, "not available"); w = W; h = H; src = F.sc->poleWarped[idx].p;
    } else if (n == "extended_side" || n == "extended_fisheye") {
      need(idx >= 0 && idx < 4 && F.extImgs[F.last_pole].p, "not available");
      w = F.extW; h = idx < 2 ? F.poleRowsT : F.poleRowsB;
      const int slot = n == "extended_side" ? idx : (idx < 2 ? 4 : 5);
      src = F.extImgs[F.last_pole].as<uchar4>() + F.extStride * slot;
    } else if (n == "bottom_image" || n == "bottom_image2") {
      need(F.prImgs[F.last_pr].p && F.have_prev_pr, "not available (pole removal not run)");
      w = F.poleW; h = F.poleH;
      src = F.prImgs[F.last_pr].as<uchar4>() + (n == "bottom_image2" ? (size_t)w * h : 0);
    } else if (n == "eye_l" || n == "eye_r") {
      const int e = n == "eye_r";
      need(F.pano[e].p, "not available");
      w = W; h = H; ch = 3;
      if (dst) {
        c->op_f.ensure((size_t)w * h * 3);
        wait_finish_stream(c);  // the pack kernel reads the composited panorama
        launch_pack_bgr(c->st, F.pano[e].as<uchar4>(), w, h, c->op_f.as<uint8_t>());
        src = c->op_f.p;
      }
    } else {
      throw Error(S360_ERR_INVALID_ARG, "unknown intermediate name: " + n);
    }
    whc[0] = w; whc[1] = h; whc[2] = ch;
    if (dst) d2h(c, dst, src, (size_t)w * h * ch);
  });
}
int s360_frame_get_f32(s360_ctx* c, const char* name, int idx, int whc[3], float* dst) {
  return frame_guard(c, [&] {
    need(c && name && whc, "bad argument");
    FrameState& F = frame_state(c);
    const s360_geometry& g = c->g;
    const std::string n(name);
    const void* src = nullptr;
    int w = 0, h = 0;
    const int nloc = F.side_p1 - F.side_p0;
    if (n == "flow_l_to_r" || n == "flow_r_to_l") {
      need(idx >= F.side_p0 && idx < F.side_p1 && F.sideFlows.p, "flow not available");
      w = g.overlap_image_width; h = g.cam_image_height;
      const int j = idx - F.side_p0 + (n == "flow_r_to_l" ? nloc : 0);
      src = F.sideFlows.as<float2>() + (size_t)w * h * j;
    } else if (n == "flow_bottom_secondary") {
      need(F.prFlow[F.last_pr].p && F.have_prev_pr, "not available (pole removal not run)");
      w = F.poleW; h = F.poleH;
      src = F.prFlow[F.last_pr].p;
    } else if (n == "flow_pole") {
      need(idx >= 0 && idx < 4 && F.poleFlows.p, "flow not available");
      w = F.extW; h = idx < 2 ? F.poleRowsT : F.poleRowsB;
      src = F.poleFlows.as<float2>() + F.extStride * idx;
    } else {
      throw Error(S360_ERR_INVALID_ARG, "unknown intermediate name: " + n);
    }
    whc[0] = w; whc[1] = h; whc[2] = 2;
    if (dst) d2h(c, dst, src, (size_t)w * h * sizeof(float2));
  });
}

int s360_set_sweep_mode(s360_ctx* c, const char* mode) {
  return guard(c, [&] {
    need(c && mode, "bad argument");
    const std::string m(mode);
    if (m == "latency") c->sweep_mode = 2;
    else if (m == "throughput") c->sweep_mode = 3;
    else throw Error(S360_ERR_INVALID_ARG, "sweep mode must be \"latency\" or \"throughput\"");
    if (c->flow) c->flow->set_sweep_mode(c->sweep_mode);
    if (c->flow_pole) c->flow_pole->set_sweep_mode(c->sweep_mode);
    if (c->flow_pr) c->flow_pr->set_sweep_mode(c->sweep_mode);
  });
}

// ---- measurement -----------------------------------------------------------------------------------
int s360_profile_enable(s360_ctx* c, int on) {
  return guard(c, [&] {
    need(c, "null ctx");
    S360_HIP(hipStreamSynchronize(c->st));
    c->prof.clear();
    c->prof.on = on != 0;
  });
}
int s360_profile_get(s360_ctx* c, char* names_out, size_t names_cap, float* ms_out, int* launches_out, int cap) {
  int n = 0;
  const int rc = guard(c, [&] {
    need(c && names_out && ms_out, "bad argument");
    std::vector<float> ms;
    std::vector<int> cnt;
    c->prof.collect(ms, cnt);
    std::string names;
    for (size_t i = 0; i < c->prof.names.size() && (int)i < cap; ++i) {
      if (i) names += ';';
      names += c->prof.names[i];
      ms_out[i] = ms[i];
      if (launches_out) launches_out[i] = cnt[i];
      n = (int)i + 1;
    }
    need(names.size() + 1 <= names_cap, "names buffer too small");
    std::memcpy(names_out, names.c_str(), names.size() + 1);
  });
  return rc == S360_OK ? n : rc;
}

// ---- flow state files (CvUtil.cpp:159-199) -----------------------------------------------------------
int s360_save_flow_to_file(const char* path, const float* flow, int w, int h) {
  return guard(nullptr, [&] {
    need(path && flow && w > 0 && h > 0, "bad argument");
    FILE* f = std::fopen(path, "wb");
    if (!f) throw Error(S360_ERR_IO, std::string("file not found: ") + path);
    const int rows = h, cols = w;
    bool ok = std::fwrite(&rows, sizeof(rows), 1, f) == 1 && std::fwrite(&cols, sizeof(cols), 1, f) == 1;
    ok = ok && std::fwrite(flow, sizeof(float) * 2, (size_t)w * h, f) == (size_t)w * h;
    std::fclose(f);
    if (!ok) throw Error(S360_ERR_IO, std::string("short write: ") + path);
  });
}
int s360_read_flow_from_file(const char* path, float* flow_out, int* w, int* h, size_t cap_floats) {
  return guard(nullptr, [&] {
    need(path && w && h, "bad argument");
    FILE* f = std::fopen(path, "rb");
    if (!f) throw Error(S360_ERR_IO, std::string("file not found: ") + path);
    int rows = 0, cols = 0;
    bool ok = std::fread(&rows, sizeof(rows), 1, f) == 1 && std::fread(&cols, sizeof(cols), 1, f) == 1;
    if (ok && rows > 0 && cols > 0) {
      *w = cols; *h = rows;
      if (flow_out) {
        const size_t n = (size_t)rows * cols * 2;
        if (n > cap_floats) { std::fclose(f); throw Error(S360_ERR_INVALID_ARG, "flow_out too small"); }
        ok = std::fread(flow_out, sizeof(float), n, f) == n;
      }
    } else ok = false;
    std::fclose(f);
    if (!ok) throw Error(S360_ERR_IO, std::string("bad flow file: ") + path);
  });
}


// ---- soft ISP (isp.cpp / isp_kernels.hip; Raw2Rgb's non-accelerated path, CameraIsp.h) ------------------------------
void s360_isp_config_defaults(s360_isp_config* cfg) {
  if (cfg) isp_config_defaults(cfg);
}
int s360_isp_config_from_json(const char* json_text, s360_isp_config* cfg) {
  return guard(nullptr, [&] {
    need(json_text && cfg, "null argument");
    isp_config_from_json(json_text, cfg);
  });
}
int s360_isp_create(s360_isp** out, int device, const s360_isp_config* cfg) {
  if (!out) return S360_ERR_INVALID_ARG;
  *out = nullptr;
  s360_isp* o = nullptr;
  const int rc = guard(nullptr, [&] {
    need(cfg != nullptr, "null argument");
    o = new s360_isp;
    isp_init(o, device, *cfg);
  });
  if (rc != S360_OK) {
    if (o) isp_release(o);
    delete o;
    return rc;
  }
  *out = o;
  return S360_OK;
}
void s360_isp_destroy(s360_isp* isp) {
  if (!isp) return;
  isp_release(isp);
  delete isp;
}
int s360_isp_process(s360_isp* isp, const uint16_t* raw16, int w, int h, void* out_bgr) {
  return guard(nullptr, [&] {
    need(isp && raw16 && out_bgr && w > 0 && h > 0, "bad argument");
    isp_process(isp, raw16, w, h, out_bgr);
  });
}
int s360_isp_process_packed(s360_isp* isp, const uint8_t* frame, int bits, int w, int h, void* out_bgr) {
  return guard(nullptr, [&] {
    need(isp && frame && out_bgr && w > 0 && h > 0, "bad argument");
    isp_process_packed(isp, frame, bits, w, h, out_bgr);
  });
}
int s360_isp_pipe_generated(s360_isp* isp, const s360_camera_isp_gen_args* args) {
  return guard(nullptr, [&] {
    need(isp && args, "null argument");
    isp_pipe_generated(isp, *args);
  });
}
int s360_isp_config_tables(const s360_isp_config* cfg, float* ccm9, float* lut, int w, int h, float* curve_h,
                           float* curve_v) {
  return guard(nullptr, [&] {
    need(cfg != nullptr, "null argument");
    IspDev d;
    std::vector<float> l, ch, cv;
    isp_derive(*cfg, d, l);
    if (ccm9) std::memcpy(ccm9, d.ccm, 9 * sizeof(float));
    if (lut) std::memcpy(lut, l.data(), l.size() * sizeof(float));
    if (curve_h || curve_v) {
      need(w > 0 && h > 0, "bad frame size");
      isp_vignette_curves(*cfg, w, h, ch, cv);
      if (curve_h) std::memcpy(curve_h, ch.data(), ch.size() * sizeof(float));
      if (curve_v) std::memcpy(curve_v, cv.data(), cv.size() * sizeof(float));
    }
  });
}

}  // extern "C"
