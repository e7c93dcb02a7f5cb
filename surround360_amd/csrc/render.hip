// render.hip — renderStereoPanorama (TRSP:716-972) as a device-resident HIP pipeline.
// Everything between "inputs uploaded" and "stacked equirect in HBM" runs on the context stream
// with no host round trips; buffers are persistent (sized once from rig + eqr size).
#include "render.hpp"

#include "isp.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "devmath.hpp"

namespace s360 {

// ------------------------------------------------------------------------------------------
// Host-built lookup tables (the host's libm is the reference's libm: tanhf / exp of NovelView.cpp:138-141
// and CvUtil.cpp:239-246 are evaluated here for every possible 8-bit input, so the device needs no
// transcendental for them).
// 8-bit Gaussian of featherAlphaChannel (CvUtil.cpp:140-157): ksize = erodeSize, sigma = erodeSize / 2.0f, taps
// round(k*256) (the fixed-point row/column filter of GaussianBlur on CV_8U)
std::vector<int> feather_gauss_taps(int erode_size) {
  const int n = erode_size;
  std::vector<int> ik(std::max(n, 1));
  std::vector<float> k(std::max(n, 1));
  const double sigma = erode_size / 2.0f;
  const double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
  const double scale2X = -0.5 / (sigmaX * sigmaX);
  double sum = 0;
  for (int i = 0; i < n; ++i) {
    const double x = i - (n - 1) * 0.5;
    k[i] = (float)std::exp(scale2X * x * x);
    sum += k[i];
  }
  sum = 1. / sum;
  for (int i = 0; i < n; ++i) ik[i] = cv_round((float)(k[i] * sum) * 256.f);
  return ik;
}

void Tables::build(hipStream_t st, int std_feather) {
  // OpenCV initInterTab2D(INTER_CUBIC, fixpt) — see SURVEY App. A.2
  std::vector<float> tf(1024 * 16);
  std::vector<short> ti(1024 * 16);
  float t1[32][4];
  for (int k = 0; k < 32; ++k) cubic_coeffs(k * (1.f / 32), t1[k]);
  for (int iy = 0; iy < 32; ++iy)
    for (int ix = 0; ix < 32; ++ix) {
      float* f = &tf[(iy * 32 + ix) * 16];
      short* i = &ti[(iy * 32 + ix) * 16];
      int isum = 0;
      for (int k1 = 0; k1 < 4; ++k1)
        for (int k2 = 0; k2 < 4; ++k2) {
          const float v = t1[iy][k1] * t1[ix][k2];
          f[k1 * 4 + k2] = v;
          isum += i[k1 * 4 + k2] = (short)sat_s16(cv_round(v * 32768.f));
        }
      if (isum != 32768) {
        const int diff = isum - 32768;
        int Mk1 = 2, Mk2 = 2, mk1 = 2, mk2 = 2;
        for (int k1 = 2; k1 < 4; ++k1)
          for (int k2 = 2; k2 < 4; ++k2) {
            if (i[k1 * 4 + k2] < i[mk1 * 4 + mk2]) mk1 = k1, mk2 = k2;
            else if (i[k1 * 4 + k2] > i[Mk1 * 4 + Mk2]) Mk1 = k1, Mk2 = k2;
          }
        if (diff < 0) i[Mk1 * 4 + Mk2] = (short)(i[Mk1 * 4 + Mk2] - diff);
        else i[mk1 * 4 + mk2] = (short)(i[mk1 * 4 + mk2] - diff);
      }
    }
  // The table taken apart again (render_kernels.hip, k_remap_cubic_u8c4_packed<.., true>): entry (iy, ix) = round-to-even of the 16
  // float products t1[iy][k1] * t1[ix][k2] * 32768, saturated to int16, plus ONE residue on one of four taps. Rebuilt here the way
  // the kernel rebuilds it (float product with x's taps pre-scaled by 2^15 — exact —, the one entry that reaches 32768 clamped, the
  // 16-bit wrap of the residue add) and compared with all 1024 x 16 entries: only then are the pieces handed to the kernels.
  std::vector<float> w1(2 * 32 * 4);
  std::vector<short> res(1024, 0);
  bool rebuilt = true;
  for (int k = 0; k < 32; ++k)
    for (int q = 0; q < 4; ++q) { w1[k * 4 + q] = t1[k][q]; w1[128 + k * 4 + q] = t1[k][q] * 32768.f; }
  for (int e = 0; e < 1024 && rebuilt; ++e) {
    const int iy = e >> 5, ix = e & 31;
    short r[16];
    for (int k1 = 0; k1 < 4; ++k1)
      for (int k2 = 0; k2 < 4; ++k2) {
        volatile float p = w1[iy * 4 + k1] * w1[128 + ix * 4 + k2];  // (volatile: rounded to float here, like the device's v_mul_f32)
        r[k1 * 4 + k2] = (short)cv_round(std::min((float)p, 32767.f));
      }
    int pos = -1, diff = 0;
    for (int t = 0; t < 16; ++t)
      if (r[t] != ti[e * 16 + t]) {
        if (pos >= 0) rebuilt = false;
        pos = t;
        diff = (int)ti[e * 16 + t] - (int)r[t];
      }
    if (pos >= 0) {
      const int k1 = pos >> 2, k2 = pos & 3;
      if (k1 < 2 || k2 < 2 || diff < -8192 || diff > 8191) rebuilt = false;
      else res[e] = (short)((diff << 2) | ((k1 - 2) * 2 + (k2 - 2)));
    }
  }
  std::vector<float> t10(766), t5(766), fs(256);
  for (int s = 0; s < 766; ++s) {
    const float colorDiff = (float)s / 255.0f;
    t10[s] = tanhf(colorDiff * 10.0f);
    t5[s] = tanhf(colorDiff * 5.0f);
  }
  for (int a = 0; a < 256; ++a) {  // CvUtil.cpp:243-250
    const float alphaR = (float)a / 255.0f;
    const float alphaL = 1.0f - alphaR;
    // `using namespace std` + float arguments: the reference resolves exp() to expf here
    const double expL = (double)expf(5.0f * alphaL * 2.0f);
    const double expR = (double)expf(5.0f * alphaR);
    const double sumExp = expL + expR + 0.00001;
    fs[a] = float(expL / sumExp);
  }
  gauss_ksize = std_feather;
  const std::vector<int> ik = feather_gauss_taps(std_feather);
  bi.ensure(ti.size() * sizeof(short));
  bf.ensure(tf.size() * sizeof(float));
  this->t10.ensure(766 * sizeof(float));
  this->t5.ensure(766 * sizeof(float));
  this->fs.ensure(256 * sizeof(float));
  gik.ensure(ik.size() * sizeof(int));
  S360_HIP(hipMemcpyAsync(bi.p, ti.data(), ti.size() * sizeof(short), hipMemcpyHostToDevice, st));
  S360_HIP(hipMemcpyAsync(bf.p, tf.data(), tf.size() * sizeof(float), hipMemcpyHostToDevice, st));
  S360_HIP(hipMemcpyAsync(this->t10.p, t10.data(), 766 * sizeof(float), hipMemcpyHostToDevice, st));
  S360_HIP(hipMemcpyAsync(this->t5.p, t5.data(), 766 * sizeof(float), hipMemcpyHostToDevice, st));
  S360_HIP(hipMemcpyAsync(this->fs.p, fs.data(), 256 * sizeof(float), hipMemcpyHostToDevice, st));
  S360_HIP(hipMemcpyAsync(gik.p, ik.data(), ik.size() * sizeof(int), hipMemcpyHostToDevice, st));
  S360_HIP(hipStreamSynchronize(st));
  dev.bicubic_i = bi.as<short>();
  dev.bicubic_f = bf.as<float>();
  dev.bicubic_w1 = nullptr;
  dev.bicubic_res = nullptr;
  if (rebuilt) {
    bw1.ensure(w1.size() * sizeof(float));
    bres.ensure(res.size() * sizeof(short));
    S360_HIP(hipMemcpyAsync(bw1.p, w1.data(), w1.size() * sizeof(float), hipMemcpyHostToDevice, st));
    S360_HIP(hipMemcpyAsync(bres.p, res.data(), res.size() * sizeof(short), hipMemcpyHostToDevice, st));
    S360_HIP(hipStreamSynchronize(st));
    dev.bicubic_w1 = bw1.as<float>();
    dev.bicubic_res = bres.as<short>();
  }
  dev.tanh10 = this->t10.as<float>();
  dev.tanh5 = this->t5.as<float>();
  dev.flat_softmaxL = this->fs.as<float>();
}

static DevCamera dev_camera(const s360_camera& c) {
  DevCamera d;
  d.type = c.type;
  for (int i = 0; i < 3; ++i) d.pos[i] = c.position[i];
  for (int i = 0; i < 9; ++i) d.R[i] = c.rotation[i];
  for (int i = 0; i < 2; ++i) {
    d.principal[i] = c.principal[i];
    d.focal[i] = c.focal[i];
    d.distortion[i] = c.distortion[i];
  }
  return d;
}

void build_spherical_map(s360_ctx* c, float2* map, int dw, int dh, const s360_camera& cam, float l, float r, float t,
                         float b) {
  // ImageWarper.cpp:151-162: the angles depend on x or y only; cos/sin resolve to the float overloads.
  std::vector<float> trig(2 * dw + 2 * dh);
  float* cosX = trig.data();
  float* sinX = cosX + dw;
  float* cosY = sinX + dw;
  float* sinY = cosY + dh;
  for (int x = 0; x < dw; ++x) {
    const float xFrac = (x + 0.5f) / dw;
    const float xAngle = (1 - xFrac) * l + xFrac * r;
    cosX[x] = cosf(xAngle);
    sinX[x] = sinf(xAngle);
  }
  for (int y = 0; y < dh; ++y) {
    const float yFrac = (y + 0.5f) / dh;
    const float yAngle = (1 - yFrac) * t + yFrac * b;
    cosY[y] = cosf(yAngle);
    sinY[y] = sinf(yAngle);
  }
  DevBuf d;
  d.ensure(trig.size() * sizeof(float));
  S360_HIP(hipMemcpyAsync(d.p, trig.data(), trig.size() * sizeof(float), hipMemcpyHostToDevice, c->st));
  const float* dp = d.as<float>();
  launch_spherical_map(c->st, map, dw, dh, dev_camera(cam), dp, dp + dw, dp + 2 * dw, dp + 2 * dw + dh);
  S360_HIP(hipStreamSynchronize(c->st));  // trig buffer is freed on return
}

// ------------------------------------------------------------------------------------------
FrameState& frame_state(s360_ctx* c) {
  if (c->slots.empty()) c->slots.resize(1);
  if (c->slot < 0 || c->slot >= (int)c->slots.size()) throw Error(S360_ERR_STATE, "bad frame slot");
  std::shared_ptr<FrameState>& f = c->slots[c->slot];
  if (!f) {
    f = std::make_shared<FrameState>();
    if (!c->slotScratch) c->slotScratch = std::make_shared<SlotScratch>();
    f->sc = c->slotScratch;
    f->P = (int)c->rig.side.size();
    f->tab.build(c->st, c->P.std_alpha_feather_size);
  }
  return *f;
}
void flow_engines_follow_pipelining(s360_ctx* c) {
  if (!c->flow) return;
  std::shared_ptr<FlowBufs> finish = c->flow->buffers();
  if (c->pipeline) {
    // a set of their own for the engines of the finish stage (kept once made: a stream switches pipelining on once)
    if (c->flow_pole && c->flow_pole->buffers() != c->flow->buffers()) finish = c->flow_pole->buffers();
    else if (c->flow_pr && c->flow_pr->buffers() != c->flow->buffers()) finish = c->flow_pr->buffers();
    else finish = std::make_shared<FlowBufs>();
  }
  if (c->flow_pole) c->flow_pole->share_buffers(finish);
  if (c->flow_pr) c->flow_pr->share_buffers(finish);
}
FlowEngine& flow_engine(s360_ctx* c, int which) {
  std::unique_ptr<FlowEngine>& e = which == 0 ? c->flow : which == 1 ? c->flow_pole : c->flow_pr;
  if (!e) {
    if (which != 0) (void)flow_engine(c, 0);
    e.reset(new FlowEngine(&c->prof));
    e->set_sweep_mode(c->sweep_mode);
    flow_engines_follow_pipelining(c);
  }
  return *e;
}
void set_frame_slots(s360_ctx* c, int n) {
  if (n < 1 || n > 64) throw Error(S360_ERR_INVALID_ARG, "frame slots must be 1..64");
  S360_HIP(hipStreamSynchronize(c->st));
  if (c->st2) S360_HIP(hipStreamSynchronize(c->st2));
  if (c->stUp) S360_HIP(hipStreamSynchronize(c->stUp));
  c->slots.resize(n);
  if (c->slot >= n) c->slot = 0;
}
namespace {
struct SlotScope {  // the helpers below find their scratch buffers through frame_state(c): select the slot they work on
  s360_ctx* c;
  int saved;
  SlotScope(s360_ctx* c_, int k) : c(c_), saved(c_->slot) { c->slot = k; }
  ~SlotScope() { c->slot = saved; }
};
}  // namespace

// ---- uploads: pinned chunk ring + upload stream (see ctx.hpp) ----------------------------------------------------
static void ensure_upload_stream(s360_ctx* c) {
  if (c->stUp) return;
  S360_HIP(hipStreamCreateWithFlags(&c->stUp, hipStreamNonBlocking));
  S360_HIP(hipEventCreateWithFlags(&c->evUploaded, hipEventDisableTiming));
  S360_HIP(hipEventCreateWithFlags(&c->evSideSrcFree, hipEventDisableTiming));
  if (!c->evPoleSrcFree) S360_HIP(hipEventCreateWithFlags(&c->evPoleSrcFree, hipEventDisableTiming));
  for (int i = 0; i < s360_ctx::kPinChunks; ++i) {
    S360_HIP(hipHostMalloc(&c->pin[i], s360_ctx::kPinChunkBytes, hipHostMallocDefault));
    S360_HIP(hipEventCreateWithFlags(&c->pinEv[i], hipEventDisableTiming));
  }
}
// host -> device through the pinned ring on stUp; `src` may be reused as soon as this returns — unless it lies in a
// buffer from s360_host_alloc: then the copy is enqueued straight from it (no staging copy, no waiting for ring chunks:
// the call costs microseconds) and the buffer must stay untouched until s360_frame_uploads_complete has returned.
static void upload_bytes(s360_ctx* c, void* dst, const void* src, size_t bytes) {
  const char* s = static_cast<const char*>(src);
  char* d = static_cast<char*>(dst);
  if (host_is_pinned(src, bytes)) {
    S360_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stUp));
    return;
  }
  for (size_t off = 0; off < bytes; off += s360_ctx::kPinChunkBytes) {
    const size_t len = std::min(s360_ctx::kPinChunkBytes, bytes - off);
    const int i = c->pinNext;
    c->pinNext = (i + 1) % s360_ctx::kPinChunks;
    if (c->pinUsed[i]) S360_HIP(hipEventSynchronize(c->pinEv[i]));  // the chunk's previous copy has left the host
    std::memcpy(c->pin[i], s + off, len);
    S360_HIP(hipMemcpyAsync(d + off, c->pin[i], len, hipMemcpyHostToDevice, c->stUp));
    S360_HIP(hipEventRecord(c->pinEv[i], c->stUp));
    c->pinUsed[i] = true;
  }
}
static void uploads_done(s360_ctx* c) {
  S360_HIP(hipEventRecord(c->evUploaded, c->stUp));
  c->haveUploaded = true;
}
// called by the render stages before they read the source images / after they have read them
static void wait_for_uploads(s360_ctx* c, hipStream_t st) {
  if (c->haveUploaded) S360_HIP(hipStreamWaitEvent(st, c->evUploaded, 0));
}

void frame_upload_side(s360_ctx* c, int idx, const uint8_t* img, int w, int h, int ch) {
  FrameState& F = frame_state(c);
  if (idx < 0 || idx >= F.P) throw Error(S360_ERR_INVALID_ARG, "side_idx out of range");
  if (ch != 3 && ch != 4) throw Error(S360_ERR_INVALID_ARG, "side image must have 3 or 4 channels");
  if (F.have_side && (w != F.srcW || h != F.srcH)) throw Error(S360_ERR_INVALID_ARG, "side image size changed");
  if (F.P > 64) throw Error(S360_ERR_INVALID_ARG, "more than 64 side cameras");
  ensure_upload_stream(c);
  F.srcW = w;
  F.srcH = h;
  const size_t n = (size_t)w * h;
  F.sideSrc.ensure(F.P * n * sizeof(uchar4));
  F.staging.ensure(n * 4);
  upload_bytes(c, F.staging.p, img, n * ch);  // stUp is in order: the previous image's conversion has read the staging buffer
  if (c->haveSideSrcFree) S360_HIP(hipStreamWaitEvent(c->stUp, c->evSideSrcFree, 0));  // the previous frame's projections
  launch_prepare_side_src(c->stUp, F.staging.as<uint8_t>(), ch, F.sideSrc.as<uchar4>() + n * idx, w, h,
                          c->P.side_alpha_feather_size);
  uploads_done(c);
  F.have_side = true;
  F.side_uploaded |= 1ull << idx;
}
void frame_upload_pole(s360_ctx* c, bool top, const uint8_t* bgr, int w, int h) {
  FrameState& F = frame_state(c);
  ensure_upload_stream(c);
  if (top) { F.topW = w; F.topH = h; } else { F.poleW = w; F.poleH = h; }
  const size_t n = (size_t)w * h;
  DevBuf& dst = top ? F.topSrc : F.botSrc;
  dst.ensure(n * sizeof(uchar4));
  F.staging.ensure(n * 4);
  upload_bytes(c, F.staging.p, bgr, n * 3);
  if (c->havePoleSrcFree) S360_HIP(hipStreamWaitEvent(c->stUp, c->evPoleSrcFree, 0));  // the previous frame's pole projections
  launch_bgr_to_bgra(c->stUp, F.staging.as<uint8_t>(), 3, dst.as<uchar4>(), n);
  uploads_done(c);
  (top ? F.have_top : F.have_bottom) = true;
}

// A camera's raw Bayer frame through the ISP straight into the frame's source slot ("feeding the GPU
// directly"): what the reference does through files — Unpacker writes the ISP's 16-bit result as a PNG, the renderer's
// imread decodes it to 8 bits, i.e. keeps the high byte — happens on the upload stream without leaving the device.
// which: side index, -1 top, -2 bottom.
// bits: 16 = raw16 samples; 8 / 12 = the sensor's packed bytes as a capture container holds them (widened on the device).
void frame_upload_raw(s360_ctx* c, s360_isp* isp, int which, const void* raw, int bits, int inW, int inH) {
  if (isp->cfg.output_bpp != 16)
    throw Error(S360_ERR_INVALID_ARG, "uploading through the ISP needs output_bpp 16 (the reference's chain stores 16-bit PNGs and imread keeps their high byte)");
  if (isp->device != c->device) throw Error(S360_ERR_INVALID_ARG, "the ISP object lives on another device");
  FrameState& F = frame_state(c);
  if (which < -2 || which >= F.P) throw Error(S360_ERR_INVALID_ARG, "camera index out of range");
  ensure_upload_stream(c);
  const int w = inW / isp->cfg.resize, h = inH / isp->cfg.resize;
  const size_t n = (size_t)w * h;
  if (bits == 16) {
    upload_bytes(c, isp_raw_buffer(isp, inW, inH), raw, (size_t)inW * inH * sizeof(uint16_t));
  } else {
    upload_bytes(c, isp_packed_buffer(isp, bits, inW, inH), raw, isp_packed_bytes(bits, inW, inH));
    isp_unpack_on(isp, c->stUp, bits, inW, inH);
  }
  const void* out16 = isp_enqueue_on(isp, c->stUp, c->uid, inW, inH);
  F.staging.ensure(n * 4);
  launch_u16_high_byte(c->stUp, static_cast<const unsigned short*>(out16), F.staging.as<uint8_t>(), n * 3);
  if (which >= 0) {  // as frame_upload_side from the staging buffer
    if (F.have_side && (w != F.srcW || h != F.srcH)) throw Error(S360_ERR_INVALID_ARG, "side image size changed");
    if (F.P > 64) throw Error(S360_ERR_INVALID_ARG, "more than 64 side cameras");
    F.srcW = w;
    F.srcH = h;
    F.sideSrc.ensure(F.P * n * sizeof(uchar4));
    if (c->haveSideSrcFree) S360_HIP(hipStreamWaitEvent(c->stUp, c->evSideSrcFree, 0));
    launch_prepare_side_src(c->stUp, F.staging.as<uint8_t>(), 3, F.sideSrc.as<uchar4>() + n * which, w, h,
                            c->P.side_alpha_feather_size);
    F.have_side = true;
    F.side_uploaded |= 1ull << which;
  } else {  // as frame_upload_pole
    const bool top = which == -1;
    if (top) { F.topW = w; F.topH = h; } else { F.poleW = w; F.poleH = h; }
    DevBuf& dst = top ? F.topSrc : F.botSrc;
    dst.ensure(n * sizeof(uchar4));
    if (c->havePoleSrcFree) S360_HIP(hipStreamWaitEvent(c->stUp, c->evPoleSrcFree, 0));
    launch_bgr_to_bgra(c->stUp, F.staging.as<uint8_t>(), 3, dst.as<uchar4>(), n);
    (top ? F.have_top : F.have_bottom) = true;
  }
  uploads_done(c);
}

void frame_upload_pole_removal(s360_ctx* c, const uint8_t* bottom2, const uint8_t* mask, const uint8_t* mask2, int w, int h) {
  FrameState& F = frame_state(c);
  if (F.have_bottom && (w != F.poleW || h != F.poleH))
    throw Error(S360_ERR_INVALID_ARG, "the secondary bottom image and the pole masks must have the bottom camera's size");
  ensure_upload_stream(c);
  const size_t n = (size_t)w * h;
  F.botSrc2.ensure(n * sizeof(uchar4));
  F.staging.ensure(n * 4);
  for (int k = 0; k < 2; ++k) F.prRed[k].ensure(n);
  upload_bytes(c, F.staging.p, bottom2, n * 3);
  if (c->havePoleSrcFree) S360_HIP(hipStreamWaitEvent(c->stUp, c->evPoleSrcFree, 0));
  launch_bgr_to_bgra(c->stUp, F.staging.as<uint8_t>(), 3, F.botSrc2.as<uchar4>(), n);
  const uint8_t* masks[2] = {mask, mask2};
  for (int k = 0; k < 2; ++k) {
    upload_bytes(c, F.staging.p, masks[k], n * 3);
    launch_red_mask(c->stUp, F.staging.as<uint8_t>(), F.prRed[k].as<uint8_t>(), n);
  }
  uploads_done(c);
  F.have_pr_inputs = true;
}

// featherAlphaChannel (CvUtil.cpp:140-157) of a whole BGRA image, in place.
static void dev_feather_in_place(s360_ctx* c, uchar4* img, int w, int h) {
  FrameState& F = frame_state(c);
  const size_t n = (size_t)w * h;
  F.a8a.ensure(n);
  F.a8b.ensure(n);
  launch_erode_alpha(c->st, img, F.a8b.as<uint8_t>(), w, h, c->P.std_alpha_feather_size);
  const uint8_t* alpha = F.a8b.as<uint8_t>();
  if (F.tab.gauss_ksize > 1) {
    launch_gauss_u8(c->st, F.a8b.as<uint8_t>(), F.a8a.as<uint8_t>(), w, h, F.tab.gik.as<int>(), F.tab.gauss_ksize / 2);
    alpha = F.a8a.as<uint8_t>();
  }
  launch_extend_wrap(c->st, img, alpha, w, h, img, w);  // extW == cols: every thread rewrites its own pixel
}

// combineBottomImagesWithPoleRemoval (PoleRemoval.cpp:32-188): merged BGRA bottom image in F.prMerged.
static void dev_pole_removal(s360_ctx* c, FrameState& F, bool use_prev) {
  if (!F.have_pr_inputs) throw Error(S360_ERR_STATE, "enable_pole_removal: secondary bottom image / pole masks not uploaded");
  hipStream_t st = c->st;
  const int w = F.poleW, h = F.poleH;
  const size_t n = (size_t)w * h;
  const int bi = c->bottom_idx, b2 = c->rig.find_largest_axis_dist();
  const s360_camera& cam = c->rig.all[bi];
  const s360_camera& cam2 = c->rig.all[b2];
  const float radius = approximate_usable_pixels_radius(&cam), radius2 = approximate_usable_pixels_radius(&cam2);
  const double* up1 = cam.rotation + 3;
  const double* up2 = cam2.rotation + 3;
  const bool flip180 = up1[0] * up2[0] + up1[1] * up2[1] + up1[2] * up2[2] < 0;  // TRSP:580
  const bool usePrev = use_prev && F.have_prev_pr;
  // temporal double buffer: the previous state is only kept apart when this frame reads it
  const int prv = F.last_pr, cur = usePrev ? prv ^ 1 : prv;
  F.prImgs[cur].ensure(2 * n * sizeof(uchar4));
  F.prFlow[cur].ensure(n * sizeof(float2));
  F.prTmp.ensure(n * sizeof(uchar4));
  F.prWarp.ensure(n * sizeof(uchar4));
  F.prMerged.ensure(n * sizeof(uchar4));
  uchar4* img1 = F.prImgs[cur].as<uchar4>();
  uchar4* img2 = img1 + n;
  {
    ProfScope ps(c->prof, "pole_removal_prepare");
    launch_circle_alpha(st, F.botSrc.as<uchar4>(), F.prRed[0].as<uint8_t>(), img1, w, h, radius);
    dev_feather_in_place(c, img1, w, h);
    uchar4* t2 = flip180 ? F.prTmp.as<uchar4>() : img2;
    launch_circle_alpha(st, F.botSrc2.as<uchar4>(), F.prRed[1].as<uint8_t>(), t2, w, h, radius2);
    dev_feather_in_place(c, t2, w, h);
    if (flip180) launch_flip_both(st, t2, img2, w, h, h);
  }
  {
    (void)flow_engine(c, 2);
    const std::string alg = c->P.poleremoval_flow_alg[0] ? c->P.poleremoval_flow_alg : "pixflow_low";
    FlowBatch fb;
    fb.add_images(img1, 2, n);
    if (usePrev) fb.add_prev_images(F.prImgs[prv].as<uchar4>(), 2, n);
    fb.add_flow(0, 1, F.prFlow[cur].as<float2>(), usePrev ? F.prFlow[prv].as<float2>() : nullptr);
    c->flow_pr->compute(st, pixflow_consts_by_name(alg), fb, w, h, S360_HINT_DOWN);
  }
  {
    ProfScope ps(c->prof, "pole_removal_merge");
    launch_remap_by_flow(st, img2, w, h, F.prFlow[cur].as<float2>(), F.prWarp.as<uchar4>(), F.tab.dev);
    S360_HIP(hipMemcpyAsync(F.prMerged.p, img1, n * sizeof(uchar4), hipMemcpyDeviceToDevice, st));
    launch_pole_removal_combine(st, F.prMerged.as<uchar4>(), F.prWarp.as<uchar4>(), n);
    launch_circle_alpha(st, F.prMerged.as<uchar4>(), nullptr, F.prMerged.as<uchar4>(), w, h, radius);
    dev_feather_in_place(c, F.prMerged.as<uchar4>(), w, h);
  }
  F.have_prev_pr = true;
  F.last_pr = cur;
}

static void ensure_maps(s360_ctx* c) {
  if (c->maps_ready) return;
  const s360_geometry& g = c->g;
  const int P = (int)c->rig.side.size();
  const size_t mn = (size_t)g.cam_image_width * g.cam_image_height;
  c->sideMaps.ensure(P * mn * sizeof(float2));
  for (int i = 0; i < P; ++i) {
    float l, r, t, b;
    side_camera_angles(g, i, P, &l, &r, &t, &b);
    build_spherical_map(c, c->sideMaps.as<float2>() + mn * i, g.cam_image_width, g.cam_image_height, c->rig.side[i], l, r,
                        t, b);
  }
  if (c->P.enable_top && c->top_idx >= 0) {  // TRSP:655-667
    const s360_camera& cam = c->rig.all[c->top_idx];
    c->topMap.ensure((size_t)c->P.eqr_width * g.top_rows * sizeof(float2));
    build_spherical_map(c, c->topMap.as<float2>(), c->P.eqr_width, g.top_rows, cam, (float)(2.0f * M_PI), 0.f,
                        (float)(M_PI / 2.0f), (float)(M_PI / 2.0f - camera_get_fov(&cam)));
  }
  if (c->P.enable_bottom && c->bottom_idx >= 0) {  // TRSP:606-618
    const s360_camera& cam = c->rig.all[c->bottom_idx];
    c->botMap.ensure((size_t)c->P.eqr_width * g.bottom_rows * sizeof(float2));
    build_spherical_map(c, c->botMap.as<float2>(), c->P.eqr_width, g.bottom_rows, cam, 0.f, (float)(2.0f * M_PI),
                        (float)(-(M_PI / 2.0f)), (float)(-(M_PI / 2.0f - camera_get_fov(&cam))));
  }
  S360_HIP(hipEventCreateWithFlags(&c->evMaps, hipEventDisableTiming));
  S360_HIP(hipEventRecord(c->evMaps, c->st));
  c->maps_ready = true;
}

// coordinates + tile boxes of a cached map for sources of sw x sh (once per rig and source size): the entry for that size,
// its users on `st` ordered behind the pack kernel by an event (no host wait in the frame's enqueue path)
static s360_ctx::PackedMap& ensure_packed(s360_ctx* c, s360_ctx::PackedCache& cache, const float2* map, int sw, int sh, int dw, int dh,
                                          int batch, hipStream_t st) {
  for (auto& e : cache.e)
    if (e->sw == sw && e->sh == sh) {
      S360_HIP(hipStreamWaitEvent(st, e->ready, 0));
      return *e;
    }
  if (cache.e.size() >= 4) {  // (never seen in practice: a fifth source size for one map. Nothing may still read the oldest entry)
    S360_HIP(hipDeviceSynchronize());
    if (cache.e.front()->ready) (void)hipEventDestroy(cache.e.front()->ready);
    cache.e.erase(cache.e.begin());
  }
  if (c->evMaps) S360_HIP(hipStreamWaitEvent(st, c->evMaps, 0));
  cache.e.emplace_back(new s360_ctx::PackedMap());
  s360_ctx::PackedMap& pk = *cache.e.back();
  pk.packed.ensure((size_t)batch * dw * dh * sizeof(unsigned));
  pk.tiles.ensure((size_t)batch * remap_packed_tiles(dw, dh) * 16);
  launch_remap_pack_map(st, map, sw, sh, dw, dh, pk.packed.as<unsigned>(), pk.tiles.p, batch);
  S360_HIP(hipEventCreateWithFlags(&pk.ready, hipEventDisableTiming));
  S360_HIP(hipEventRecord(pk.ready, st));
  pk.sw = sw;
  pk.sh = sh;
  return pk;
}

// Side stage for pairs [p0,p1) of a set of frame slots: per slot the projections of the cameras those pairs touch and
// the overlap crops; then the two flows per pair of ALL slots in one FlowEngine batch; then per slot the fused
// novel-view/blend into the strip buffers. One slot = the classic single frame.
static void side_stage(s360_ctx* c, const std::vector<int>& slotIds, int p0, int p1, int use_prev) {
  const s360_geometry& g = c->g;
  const int P = (int)c->rig.side.size();
  if (p0 < 0 || p1 > P || p1 < p0) throw Error(S360_ERR_INVALID_ARG, "bad pair range");
  if (c->P.eqr_width % P != 0)
    throw Error(S360_ERR_INVALID_ARG, "eqr_width must be evenly divisible by the number of cameras");  // TRSP:729-738
  if (g.num_novel_views != c->P.eqr_width / P)
    throw Error(S360_ERR_INVALID_ARG, "numNovelViews != eqr_width / numCams for this rig");
  Profiler& prof = c->prof;
  hipStream_t st = c->st;
  const int camW = g.cam_image_width, camH = g.cam_image_height, ow = g.overlap_image_width;
  const int stripW = c->P.eqr_width / P;
  const size_t pn = (size_t)camW * camH, on = (size_t)ow * camH;
  const int n = p1 - p0;
  std::vector<FrameState*> Fs;
  for (int k : slotIds) {
    SlotScope ss(c, k);
    FrameState& F = frame_state(c);
    if (!F.have_side) throw Error(S360_ERR_STATE, "side images not uploaded");
    for (int p = p0; p < p1; ++p)
      if (!((F.side_uploaded >> p) & 1) || !((F.side_uploaded >> ((p + 1) % P)) & 1))
        throw Error(S360_ERR_STATE, "side image of camera " + std::to_string(((F.side_uploaded >> p) & 1) ? (p + 1) % P : p) +
                                        " not uploaded (needed by pair " + std::to_string(p) + ")");
    F.sc->proj.ensure(P * pn * sizeof(uchar4));
    F.strips.ensure((size_t)2 * P * camH * stripW * sizeof(uchar4));
    Fs.push_back(&F);
  }
  ensure_maps(c);
  if (n == 0) {
    // a rank that owns no pair still takes part in the frame's stream protocol: the strips it is about to RECEIVE
    // (frame_gather_strips on st) must wait for the previous frame's assemble_pano, and the events later stages and
    // the next upload wait for must be recorded
    if (c->evSideSrcFree) { S360_HIP(hipEventRecord(c->evSideSrcFree, st)); c->haveSideSrcFree = true; }
    if (c->pipeline && c->haveStripsFree) S360_HIP(hipStreamWaitEvent(st, c->evStripsFree, 0));
    if (c->pipeline) S360_HIP(hipEventRecord(c->evSideDone, st));
    return;
  }
  if (2 * n * (int)Fs.size() > kMaxFlows) throw Error(S360_ERR_INVALID_ARG, "too many flows for one batch");
  wait_for_uploads(c, st);
  // temporal state is used only if every slot of the batch has it (one FlowEngine batch = one setting)
  bool usePrev = use_prev != 0;
  for (FrameState* F : Fs) usePrev = usePrev && F->have_prev_side && F->side_p0 == p0 && F->side_p1 == p1;
  // temporal double buffer: frame k's overlaps / flows are kept apart from frame k-1's only when frame k reads them
  // (independent frames keep writing the same half: 1.4 GB less per frame slot at 8K)
  for (FrameState* F : Fs) F->cur_side = usePrev ? F->last_side ^ 1 : F->last_side;
  for (FrameState* Fp : Fs) {
    FrameState& F = *Fp;
    const size_t sn = (size_t)F.srcW * F.srcH;
    {
      ProfScope ps(prof, "project_side");
      std::vector<char> need(P, 0);
      for (int p = p0; p < p1; ++p) need[p] = need[(p + 1) % P] = 1;
      for (int i = 0; i < P;) {  // one launch per run of consecutive cameras (all of them unless the frame is sharded)
        if (!need[i]) { ++i; continue; }
        int j = i;
        while (j < P && need[j]) ++j;
        s360_ctx::PackedMap& pk = ensure_packed(c, c->sidePk, c->sideMaps.as<float2>(), F.srcW, F.srcH, camW, camH, P, st);
        launch_remap_cubic_u8c4_packed(st, F.sideSrc.as<uchar4>() + sn * i, F.srcW, F.srcH, c->sideMaps.as<float2>() + pn * i,
                                       pk.packed.as<unsigned>() + pn * i,
                                       (const char*)pk.tiles.p + 16 * remap_packed_tiles(camW, camH) * i,
                                       F.sc->proj.as<uchar4>() + pn * i, camW, camH, F.tab.dev, 0, 0, 1, j - i);
        i = j;
      }
    }
    const int cur = F.cur_side;
    F.overlaps[cur].ensure(2 * n * on * sizeof(uchar4));
    F.sideFlows.ensure(2 * n * on * sizeof(float2));
    {
      ProfScope ps(prof, "crop_overlaps");
      launch_crop_overlaps(st, F.sc->proj.as<uchar4>(), camW, camH, P, ow, F.overlaps[cur].as<uchar4>(), p0, p1);
    }
  }
  if (c->evSideSrcFree) {  // the next frame's side images may be converted into sideSrc from here on
    S360_HIP(hipEventRecord(c->evSideSrcFree, st));
    c->haveSideSrcFree = true;
  }
  {
    // NovelViewGeneratorAsymmetricFlow::prepare (NovelView.cpp:270-299): flowLtoR = flow(I0=L, I1=R, LEFT),
    // flowRtoL = flow(I0=R, I1=L, RIGHT). Both hints only matter for pixflow_search_20; the batch is split by
    // hint in that case. Per slot: images [L_0..L_{n-1}, R_0..R_{n-1}], flows [LtoR_0.., RtoL_0..].
    (void)flow_engine(c, 0);
    const PixFlowConsts pc = pixflow_consts_by_name(c->P.side_flow_alg);
    // The flows are updated in place (previous-flow input and output of one compute): a compute that throws half-way leaves no
    // usable previous flow, so the slots' temporal state is withdrawn here and given back behind the stage (below).
    for (FrameState* F : Fs) F->have_prev_side = false;
    auto build = [&](bool ltor, bool rtol) {
      FlowBatch fb;
      for (size_t k = 0; k < Fs.size(); ++k) {
        FrameState& F = *Fs[k];
        const int cur = F.cur_side, prv = cur ^ 1;
        const int base = (int)fb.images.size();
        fb.add_images(F.overlaps[cur].as<uchar4>(), 2 * n, on);
        if (usePrev) fb.add_prev_images(F.overlaps[prv].as<uchar4>(), 2 * n, on);
        float2* out = F.sideFlows.as<float2>();
        const float2* pf = usePrev ? F.sideFlows.as<float2>() : nullptr;  // (in place: read at the flow's entry, written at its end)
        if (ltor)
          for (int j = 0; j < n; ++j) fb.add_flow(base + j, base + n + j, out + on * j, pf ? pf + on * j : nullptr);
        if (rtol)
          for (int j = 0; j < n; ++j) fb.add_flow(base + n + j, base + j, out + on * (n + j), pf ? pf + on * (n + j) : nullptr);
      }
      return fb;
    };
    if (pc.maxPercentage == 0) {
      c->flow->compute(st, pc, build(true, true), ow, camH, S360_HINT_LEFT);
    } else {
      c->flow->compute(st, pc, build(true, false), ow, camH, S360_HINT_LEFT);
      c->flow->compute(st, pc, build(false, true), ow, camH, S360_HINT_RIGHT);
    }
  }
  for (FrameState* Fp : Fs) {
    FrameState& F = *Fp;
    const int cur = F.cur_side;
    {
      ProfScope ps(prof, "novel_view");
      NovelViewParams nv;
      nv.overlapW = ow; nv.camH = camH; nv.stripW = stripW; nv.numNovelViews = g.num_novel_views;
      nv.numPairs = P; nv.numLocal = n;
      nv.camImageWidthHalf = float(camW) * 0.5f;
      nv.disp = g.verge_at_infinity_slab_displacement;
      // pipelined video stream: the previous frame's panoramas must have been assembled from the strips
      if (c->pipeline && c->haveStripsFree) S360_HIP(hipStreamWaitEvent(st, c->evStripsFree, 0));
      launch_novel_view(st, F.overlaps[cur].as<uchar4>(), F.sideFlows.as<float2>(), F.strips.as<uchar4>(), nv, p0,
                        p1, F.tab.dev);
    }
    F.side_p0 = p0;
    F.side_p1 = p1;
    F.have_prev_side = true;
    F.last_side = cur;
  }
  if (c->pipeline) S360_HIP(hipEventRecord(c->evSideDone, st));
}
void frame_render_pairs(s360_ctx* c, int p0, int p1, int use_prev) { side_stage(c, {c->slot}, p0, p1, use_prev); }

// featherAlphaChannel (CvUtil.cpp:140-157) on `rows` rows of a pano, then the x % cols extension (TRSP:399-411).
void dev_feather_alpha_to_ext(s360_ctx* c, const uchar4* pano, int cols, int rows, uchar4* ext, int extW, int erode_size,
                              const int* taps) {
  FrameState& F = frame_state(c);
  hipStream_t st = c->st;
  const size_t n = (size_t)cols * rows;
  F.a8a.ensure(n);
  F.a8b.ensure(n);
  const int e = erode_size >= 0 ? erode_size : c->P.std_alpha_feather_size;
  const int ksize = erode_size >= 0 ? erode_size : F.tab.gauss_ksize;
  const int* gik = erode_size >= 0 ? taps : F.tab.gik.as<int>();
  launch_erode_alpha(st, pano, F.a8b.as<uint8_t>(), cols, rows, e);
  if (ksize > 1) {
    launch_gauss_u8(st, F.a8b.as<uint8_t>(), F.a8a.as<uint8_t>(), cols, rows, gik, ksize / 2);
    launch_extend_wrap(st, pano, F.a8a.as<uint8_t>(), cols, rows, ext, extW);
  } else {
    launch_extend_wrap(st, pano, F.a8b.as<uint8_t>(), cols, rows, ext, extW);
  }
}

void dev_pole_unit_post(s360_ctx* c, const uchar4* extFisheye, const float2* flow, int cols, int rows, int extW,
                        uchar4* warped_out, int eqrH) {
  FrameState& F = frame_state(c);
  PoleWarpParams pw;
  pw.cols = cols; pw.rows = rows; pw.extW = extW;
  pw.maxBlendX = int(float(cols) * (1.2f - 1.0f));  // TRSP:507 (kExtendFrac - 1.0f)
  pw.poleCameraRadius = c->ramp.poleCameraRadius;
  pw.phiRampStart = c->ramp.phiRampStart;
  pw.phiMid = c->ramp.phiMid;
  pw.phiRampEnd = c->ramp.phiRampEnd;
  F.sc->warpedExt.ensure((size_t)extW * rows * sizeof(uchar4));
  // two kernels: this frame's warp as packed coordinates + tile boxes, then the packed remap (1.00 against 1.24 ms per
  // 8K frame for the one-kernel form, profiles/r03_v2_*)
  F.sc->warpPacked.ensure((size_t)extW * rows * sizeof(unsigned));
  F.sc->warpTiles.ensure(remap_packed_tiles(extW, rows) * 16);
  launch_pole_warp_packed(c->st, extFisheye, flow, F.sc->warpedExt.as<uchar4>(), pw, F.tab.dev, F.sc->warpPacked.as<unsigned>(),
                          F.sc->warpTiles.p);
  launch_pole_finish(c->st, F.sc->warpedExt.as<uchar4>(), warped_out, eqrH, pw);
}

namespace {
// Frame pipelining: everything frame_finish enqueues (its helpers read c->st) goes to the second stream.
struct FinishStream {
  s360_ctx* c;
  hipStream_t saved, savedProf;
  explicit FinishStream(s360_ctx* c_) : c(c_), saved(c_->st), savedProf(c_->prof.st) {
    if (c->pipeline) {
      c->st = c->st2;
      c->prof.st = c->st2;
      S360_HIP(hipStreamWaitEvent(c->st2, c->evSideDone, 0));  // the side stage of THIS frame
    }
  }
  ~FinishStream() {
    c->st = saved;
    c->prof.st = savedProf;
  }
};
}  // namespace

// Pole stage + composite of a set of frame slots: per slot panorama assembly, pole projections and the flow inputs of
// the enabled pole units; then the pole flows of ALL slots in one FlowEngine batch (two when the top and bottom pole
// projections differ in height); then per slot warp, composite, sharpen / final resize / stacking.
// `phases`: 1 = panorama assembly + the pole units of `pole_mask` (their warped layers stay in F.sc->poleWarped), 2 = composite
// of the layers of `composite_mask` (computed here or received: frame_gather_pole_layers) + sharpen / resize / stack.
// s360_frame_finish is both phases with the same mask; a frame whose pole units are spread over GPUs
// runs phase 1 on every owner and phase 2 on the root.
static void finish_stage(s360_ctx* c, const std::vector<int>& slotIds, int pole_mask, int use_prev, int phases = 3,
                         int composite_mask = -1) {
  if (composite_mask < 0) composite_mask = pole_mask;
  if (!(phases & 1)) pole_mask = 0;
  const s360_geometry& g = c->g;
  Profiler& prof = c->prof;
  FinishStream finishStream(c);
  hipStream_t st = c->st;
  const int P = (int)c->rig.side.size(), W = c->P.eqr_width, H = c->P.eqr_height, camH = g.cam_image_height, stripW = W / P;
  const size_t en = (size_t)W * H;
  if (!c->P.enable_top) { pole_mask &= ~3; composite_mask &= ~3; }
  if (!c->P.enable_bottom) { pole_mask &= ~12; composite_mask &= ~12; }
  const int rowsT = g.top_rows, rowsB = g.bottom_rows;
  const int extW = int(float(W) * 1.2f);  // TRSP:400-401
  const size_t xs = (size_t)extW * std::max(rowsT, rowsB);  // slot stride of the extended images / pole flows
  auto rowsOf = [&](int u) { return u < 2 ? rowsT : rowsB; };
  if (pole_mask) ensure_maps(c);
  bool usePrev = use_prev != 0 && pole_mask != 0;
  for (int k : slotIds) {
    SlotScope ss(c, k);
    FrameState& F = frame_state(c);
    usePrev = usePrev && F.have_prev_pole && F.extW == extW && F.poleRowsT == rowsT && F.poleRowsB == rowsB;
  }
  for (int k : slotIds) {
    SlotScope ss(c, k);
    FrameState& F = frame_state(c);
    F.cur_pole = usePrev ? F.last_pole ^ 1 : F.last_pole;
  }
  for (int k : slotIds) {
    SlotScope ss(c, k);
    FrameState& F = frame_state(c);
    if (!(phases & 1)) {
      if (!F.pano[0].p || !F.pano[1].p) throw Error(S360_ERR_STATE, "composite without assembled panoramas (s360_frame_pole_units first)");
      continue;
    }
    if (!F.strips.p) throw Error(S360_ERR_STATE, "no strips rendered for this frame");
    for (int e = 0; e < 2; ++e) F.pano[e].ensure(en * sizeof(uchar4));
    F.sc->panoTmp.ensure(en * sizeof(uchar4));
    {
      ProfScope ps(prof, "assemble_pano");  // TRSP:380-384, 806-807
      const float sh = g.zero_parallax_novel_view_shift_pixels;
      launch_assemble_pano(st, F.strips.as<uchar4>(), P, camH, stripW, sh, F.pano[0].as<uchar4>(), W, H);
      launch_assemble_pano(st, F.strips.as<uchar4>() + (size_t)P * camH * stripW, P, camH, stripW, -sh,
                           F.pano[1].as<uchar4>(), W, H);
    }
    if (F.keep_intermediates)
      for (int e = 0; e < 2; ++e) {
        F.panoDbg[e].ensure(en * sizeof(uchar4));
        S360_HIP(hipMemcpyAsync(F.panoDbg[e].p, F.pano[e].p, en * sizeof(uchar4), hipMemcpyDeviceToDevice, st));
      }
    if (!pole_mask) continue;
    // ---- pole units (TRSP:811-860): 0 top_left, 1 top_right, 2 bottom_left, 3 bottom_right ----
    const int cur = F.cur_pole;
    F.extImgs[cur].ensure(6 * xs * sizeof(uchar4));
    F.poleFlows.ensure(4 * xs * sizeof(float2));
    uchar4* ext = F.extImgs[cur].as<uchar4>();
    wait_for_uploads(c, st);
    {
      ProfScope ps(prof, "project_pole");
      if (pole_mask & 3) {
        if (!F.have_top) throw Error(S360_ERR_STATE, "top image not uploaded");
        F.sc->topSph.ensure((size_t)W * rowsT * sizeof(uchar4));
        const int yfs = rowsT - 1 - c->P.std_alpha_feather_size;
        s360_ctx::PackedMap& pk = ensure_packed(c, c->topPk, c->topMap.as<float2>(), F.topW, F.topH, W, rowsT, 1, st);
        launch_remap_cubic_u8c4_packed(st, F.topSrc.as<uchar4>(), F.topW, F.topH, c->topMap.as<float2>(),
                                       pk.packed.as<unsigned>(), pk.tiles.p, F.sc->topSph.as<uchar4>(), W, rowsT,
                                       F.tab.dev, 1, yfs, c->P.std_alpha_feather_size);
        launch_extend_wrap(st, F.sc->topSph.as<uchar4>(), nullptr, W, rowsT, ext + 4 * xs, extW);
      }
      if (pole_mask & 12) {
        if (!F.have_bottom) throw Error(S360_ERR_STATE, "bottom image not uploaded");
        F.sc->botSph.ensure((size_t)W * rowsB * sizeof(uchar4));
        const int yfs = rowsB - 1 - c->P.std_alpha_feather_size;
        if (c->P.enable_pole_removal) {  // TRSP:569-597: the bottom source is the merge of the two bottom cameras
          dev_pole_removal(c, F, use_prev != 0);
          s360_ctx::PackedMap& pk = ensure_packed(c, c->botPk, c->botMap.as<float2>(), F.poleW, F.poleH, W, rowsB, 1, st);
          launch_remap_cubic_u8c4_packed(st, F.prMerged.as<uchar4>(), F.poleW, F.poleH, c->botMap.as<float2>(),
                                         pk.packed.as<unsigned>(), pk.tiles.p, F.sc->botSph.as<uchar4>(), W, rowsB,
                                         F.tab.dev, 2, yfs, c->P.std_alpha_feather_size);
        } else {
          s360_ctx::PackedMap& pk = ensure_packed(c, c->botPk, c->botMap.as<float2>(), F.poleW, F.poleH, W, rowsB, 1, st);
          launch_remap_cubic_u8c4_packed(st, F.botSrc.as<uchar4>(), F.poleW, F.poleH, c->botMap.as<float2>(),
                                         pk.packed.as<unsigned>(), pk.tiles.p, F.sc->botSph.as<uchar4>(), W, rowsB,
                                         F.tab.dev, 1, yfs, c->P.std_alpha_feather_size);
        }
        launch_extend_wrap(st, F.sc->botSph.as<uchar4>(), nullptr, W, rowsB, ext + 5 * xs, extW);
      }
    }
    {
      ProfScope ps(prof, "pole_prepare");
      if (pole_mask & 12) {
        for (int e = 0; e < 2; ++e) {
          F.sc->panoFlip[e].ensure(en * sizeof(uchar4));
          launch_flip_both(st, F.pano[e].as<uchar4>(), F.sc->panoFlip[e].as<uchar4>(), W, H, rowsB);  // TRSP:842-843 (only the rows the pole unit reads)
        }
      }
      for (int u = 0; u < 4; ++u)
        if (pole_mask & (1 << u)) {
          const uchar4* side = (u < 2) ? F.pano[u & 1].as<uchar4>() : F.sc->panoFlip[u & 1].as<uchar4>();
          dev_feather_alpha_to_ext(c, side, W, rowsOf(u), ext + u * xs, extW);
        }
    }
  }
  if (c->pipeline && (phases & 1)) {  // the next frame's novel views may overwrite the strips from here on
    S360_HIP(hipEventRecord(c->evStripsFree, st));
    c->haveStripsFree = true;
  }
  if (pole_mask) {
    if (c->evPoleSrcFree) {  // the next frame's pole images may be converted into topSrc / botSrc from here on
      S360_HIP(hipEventRecord(c->evPoleSrcFree, st));
      c->havePoleSrcFree = true;
    }
    // computeOpticalFlow(extendedSide, extendedFisheye, ..., DOWN) for every enabled unit (TRSP:438-448)
    (void)flow_engine(c, 1);
    const PixFlowConsts pc = pixflow_consts_by_name(c->P.polar_flow_alg);
    for (int k : slotIds) {  // in-place flows: see side_stage — withdrawn here, given back behind the stage
      SlotScope ss(c, k);
      frame_state(c).have_prev_pole = false;
    }
    auto run = [&](int mask, int rows) {
      if (!mask) return;
      FlowBatch fb;
      for (int k : slotIds) {
        SlotScope ss(c, k);
        FrameState& F = frame_state(c);
        const int cur = F.cur_pole, prv = cur ^ 1;
        const uchar4* ext = F.extImgs[cur].as<uchar4>();
        const uchar4* pext = usePrev ? F.extImgs[prv].as<uchar4>() : nullptr;
        int fishIdx[2] = {-1, -1};
        for (int u = 0; u < 4; ++u) {
          if (!(mask & (1 << u))) continue;
          const int pole = u >> 1;  // 0 top, 1 bottom
          if (fishIdx[pole] < 0) {
            fishIdx[pole] = (int)fb.images.size();
            fb.images.push_back(ext + (4 + pole) * xs);
            if (usePrev) fb.prev_images.push_back(pext + (4 + pole) * xs);
          }
          const int sideIdx = (int)fb.images.size();
          fb.images.push_back(ext + u * xs);
          if (usePrev) fb.prev_images.push_back(pext + u * xs);
          fb.add_flow(sideIdx, fishIdx[pole], F.poleFlows.as<float2>() + u * xs,
                      usePrev ? F.poleFlows.as<float2>() + u * xs : nullptr);
        }
      }
      c->flow_pole->compute(st, pc, fb, extW, rows, S360_HINT_DOWN);
    };
    if (rowsT == rowsB) {
      run(pole_mask, rowsT);
    } else {
      run(pole_mask & 3, rowsT);
      run(pole_mask & 12, rowsB);
    }
  }
  for (int k : slotIds) {
    SlotScope ss(c, k);
    FrameState& F = frame_state(c);
    if (pole_mask) {
      const int cur = F.cur_pole;
      const uchar4* ext = F.extImgs[cur].as<uchar4>();
      {
        ProfScope ps(prof, "pole_warp");
        for (int u = 0; u < 4; ++u)
          if (pole_mask & (1 << u)) {
            F.sc->poleWarped[u].ensure(en * sizeof(uchar4));
            dev_pole_unit_post(c, ext + (u < 2 ? 4 : 5) * xs, F.poleFlows.as<float2>() + u * xs, W, rowsOf(u), extW,
                               F.sc->poleWarped[u].as<uchar4>(), H);
            F.poleFrame[u] = F.frames_done;
            F.sc->poleOwner[u] = &F;
          }
      }
      F.extW = extW;
      F.extStride = xs;
      F.poleRowsT = rowsT;
      F.poleRowsB = rowsB;
      F.have_prev_pole = true;
      F.last_pole = cur;
    }
    if (!(phases & 2)) continue;
    {
      ProfScope ps(prof, "flatten");  // TRSP:864-885
      F.sc->panoTmp.ensure(en * sizeof(uchar4));
      for (int u = 0; u < 4; ++u)
        if ((composite_mask & (1 << u)) && (!F.sc->poleWarped[u].p || F.poleFrame[u] != F.frames_done || F.sc->poleOwner[u] != &F))  // (an earlier frame's, or another slot's, layer is not this frame's)
          throw Error(S360_ERR_STATE, "composite: the warped layer of pole unit " + std::to_string(u) + " of this frame is neither computed nor received");
      for (int e = 0; e < 2; ++e) {
        if (composite_mask & (1 << e)) {
          launch_flatten(st, F.pano[e].as<uchar4>(), F.sc->poleWarped[e].as<uchar4>(), F.sc->panoTmp.as<uchar4>(), W, H, 0,
                         F.tab.dev);
          std::swap(F.pano[e].p, F.sc->panoTmp.p);
          std::swap(F.pano[e].cap, F.sc->panoTmp.cap);
        }
        if (composite_mask & (4 << e)) {
          launch_flatten(st, F.pano[e].as<uchar4>(), F.sc->poleWarped[2 + e].as<uchar4>(), F.sc->panoTmp.as<uchar4>(), W, H, 1,
                         F.tab.dev);
          std::swap(F.pano[e].p, F.sc->panoTmp.p);
          std::swap(F.pano[e].cap, F.sc->panoTmp.cap);
        }
      }
    }
  }
  if (!(phases & 2)) return;
  // sharpenThread for the eyes of several slots per set of launches (TRSP:901-915 runs the two eyes on two threads): the
  // IIR passes are one serial chain per (row, channel) — the 2 x 4096 rows of one frame are 512 waves, half a wave per
  // SIMD — so the slots of a batch are sharpened in groups of four (8 images: 2 waves per SIMD cover each other's
  // dependent chains), group after group, all groups through the same scratch (SlotScratch)
  if (c->P.sharpening > 0.0) {
    ProfScope ps(prof, "finish");
    std::vector<uchar4*> imgs;
    for (int k : slotIds) {
      SlotScope ss(c, k);
      FrameState& F = frame_state(c);
      for (int e = 0; e < 2; ++e) imgs.push_back(F.pano[e].as<uchar4>());
    }
    SlotScratch& S = *c->slotScratch;
    const int per = std::min(sharpen_max_images(), (int)SlotScratch::kSharpenGroup);
    uchar4* lps[SlotScratch::kSharpenGroup];
    float* scr[SlotScratch::kSharpenGroup];
    for (int i = 0; i < per && i < (int)imgs.size(); ++i) {
      S.sharpLp[i].ensure(en * sizeof(uchar4));
      S.sharpBuf[i].ensure(sharpen_scratch_bytes(W, H));
      lps[i] = S.sharpLp[i].as<uchar4>();
      scr[i] = S.sharpBuf[i].as<float>();
    }
    for (size_t i = 0; i < imgs.size(); i += per) {  // (group after group on this stream: the scratch is the group's)
      const int n = (int)std::min<size_t>(per, imgs.size() - i);
      launch_sharpen_many(st, imgs.data() + i, lps, scr, n, W, H, 1.0f + (float)c->P.sharpening);
    }
  }
  for (int k : slotIds) {
    SlotScope ss(c, k);
    FrameState& F = frame_state(c);
    {
      ProfScope ps(prof, "finish");  // TRSP:890-961
      const int outW = g.out_width, outH = g.out_height, eyeH = outH / 2;
      // frame pipelining (a streaming host fetches frame k while frame k+1 renders) alternates between two output
      // buffers; otherwise the same one is reused
      const int ob = (c->pipeline || c->two_outputs) ? F.out_cur ^ 1 : F.out_cur;
      F.outBGR[ob].ensure((size_t)outW * outH * 3);
      if (!F.outDone[ob]) S360_HIP(hipEventCreateWithFlags(&F.outDone[ob], hipEventDisableTiming));
      if (F.downRead[ob]) S360_HIP(hipStreamWaitEvent(st, F.downRead[ob], 0));  // a fetch of the frame this buffer held may still run
      const bool resize = (outW != W) || (eyeH != H);
      for (int e = 0; e < 2; ++e) {
        uchar4* eye = F.pano[e].as<uchar4>();
        if (resize) {
          F.sc->eyeFinal[e].ensure((size_t)outW * eyeH * sizeof(uchar4));
          launch_resize_cubic_u8c4(st, eye, W, H, en, F.sc->eyeFinal[e].as<uchar4>(), outW, eyeH, (size_t)outW * eyeH, 1);
          eye = F.sc->eyeFinal[e].as<uchar4>();
        }
        launch_pack_bgr(st, eye, outW, eyeH, F.outBGR[ob].as<uint8_t>() + (size_t)e * outW * eyeH * 3);
      }
      if (c->png_encode) {  // imwriteExceptionOnFail's PngEncoder (TRSP:938-961) on the device: png.hip
        ProfScope ps(prof, "png_encode");
        F.pngPlan[ob] = PngPlan::make(outW, outH);
        F.pngFile[ob].ensure(F.pngPlan[ob].file_bound);
        png_encode_enqueue(st, F.outBGR[ob].as<uint8_t>(), F.pngPlan[ob], F.sc->pngScratch, F.pngMeta[ob], F.pngFile[ob].as<uint8_t>());
        F.pngFrame[ob] = F.frames_done;
      }
      if (!F.outErrDev[ob].p) {
        F.outErrDev[ob].ensure(4 * sizeof(unsigned));
        S360_HIP(hipMemsetAsync(F.outErrDev[ob].p, 0, 4 * sizeof(unsigned), st));
      }
      {
        FlowEngine* eng[3] = {c->flow.get(), c->flow_pole.get(), c->flow_pr.get()};
        for (int i = 0; i < 3; ++i) {
          if (eng[i] && eng[i]->error_word()) {
            S360_HIP(hipMemcpyAsync(F.outErrDev[ob].as<unsigned>() + i, eng[i]->error_word(), sizeof(unsigned), hipMemcpyDeviceToDevice, st));
          }
        }
      }
      S360_HIP(hipEventRecord(F.outDone[ob], st));
      F.out_cur = ob;
      ++F.frames_done;
    }
  }
}
void frame_finish(s360_ctx* c, int pole_mask, int use_prev) { finish_stage(c, {c->slot}, pole_mask, use_prev); }
void frame_pole_units(s360_ctx* c, int pole_mask, int use_prev) { finish_stage(c, {c->slot}, pole_mask, use_prev, 1, 0); }
void frame_composite(s360_ctx* c, int pole_mask) { finish_stage(c, {c->slot}, 0, 0, 2, pole_mask); }

// Every slot at once (independent frames of a multi-stream job): one launch sequence, the 28 side flows of every slot
// in ONE batch of the flow kernels, the 4 pole flows of every slot in another. Results per slot are those of
// s360_frame_render on that slot.
void frame_render_batch(s360_ctx* c, int use_prev) {
  // (frame pipelining applies to batches as to frames: batch k's pole stage and composite on the second stream beside batch k+1's
  // side stage — round 6; the events between the stages are per context, recorded behind the last slot's work of a stage)
  std::vector<int> ids;
  for (int k = 0; k < (int)std::max<size_t>(c->slots.size(), 1); ++k) ids.push_back(k);
  side_stage(c, ids, 0, (int)c->rig.side.size(), use_prev);
  finish_stage(c, ids, 15, use_prev);
}

void frame_render_slots(s360_ctx* c, const int* slots, int n, int use_prev) {

  const int have = (int)std::max<size_t>(c->slots.size(), 1);
  std::vector<int> ids(slots, slots + n);
  for (int k = 0; k < n; ++k)
    if (ids[k] < 0 || ids[k] >= have || (k > 0 && ids[k] <= ids[k - 1]))
      throw Error(S360_ERR_INVALID_ARG, "frame slots must be distinct, ascending and below the number of slots");
  side_stage(c, ids, 0, (int)c->rig.side.size(), use_prev);
  finish_stage(c, ids, 15, use_prev);
}

// ---- cubemap output (TRSP:917-935) -------------------------------------------------------------------------------
// Face warp maps of convertSphericalToCubemapBicubicRemap (ImageWarper.cpp:26-128): float arithmetic with the host's
// acosf / sqrt exactly as the reference evaluates it; depends only on the sizes, so it is built once and cached.
namespace {
enum CubeFace { CUBE_BACK = 0, CUBE_LEFT, CUBE_TOP, CUBE_BOTTOM, CUBE_FRONT, CUBE_RIGHT };
void cube_index_to_vec3(float x, float y, int face, float out[3]) {
  const float dir[3] = {x, y, 0.5f};
  out[0] = dir[0]; out[1] = dir[1]; out[2] = dir[2];
  switch (face) {
    case CUBE_BACK: out[0] = dir[0]; out[1] = dir[2]; out[2] = -dir[1]; break;
    case CUBE_LEFT: out[0] = -dir[2]; out[1] = dir[0]; out[2] = -dir[1]; break;
    case CUBE_TOP: break;
    case CUBE_BOTTOM: out[0] = dir[0]; out[1] = -dir[1]; out[2] = -dir[2]; break;
    case CUBE_FRONT: out[0] = -dir[0]; out[1] = -dir[2]; out[2] = -dir[1]; break;
    case CUBE_RIGHT: out[0] = dir[2]; out[1] = -dir[0]; out[2] = -dir[1]; break;
  }
}
void cube_map_entry(float x, float y, int face, int srcCols, int srcRows, float fov, float* srcX, float* srcY) {
  float dir[3];
  cube_index_to_vec3(x, y, face, dir);
  const float r = sqrtf(dir[0] * dir[0] + dir[1] * dir[1]);
  float s2 = 0.f;  // cv::norm(Vec3f): float accumulation, sqrt, returned as double
  for (int i = 0; i < 3; ++i) s2 += dir[i] * dir[i];
  const double nrm = (double)std::sqrt(s2);
  const float phi = acosf((float)((double)dir[2] / nrm));
  float theta = r > 0.0f ? acosf(std::fabs(dir[0] / r)) : 0.0f;
  if (dir[0] > 0 && dir[1] > 0) {
  } else if (dir[0] <= 0 && dir[1] > 0) {
    theta = (float)(M_PI - theta);
  } else if (dir[0] <= 0 && dir[1] <= 0) {
    theta = (float)(M_PI + theta);
  } else {
    theta = (float)(2 * M_PI - theta);
  }
  const float phiPrime = std::min(std::max(phi, 0.0f), fov);
  const float thetaPrime = std::min(std::max(theta, 0.0f), float(2.0f * M_PI));
  *srcX = (float)(float(srcCols) * thetaPrime / (2.0f * M_PI));
  *srcY = float(srcRows) * phiPrime / fov;
}
}  // namespace

void frame_cubemap(s360_ctx* c, int fw, int fh, bool video, int* ow, int* oh) {
  FrameState& F = frame_state(c);
  if (fw <= 0 || fh <= 0) throw Error(S360_ERR_INVALID_ARG, "cubemap face size must be positive");
  if (!F.pano[0].p || !F.pano[1].p || !F.frames_done) throw Error(S360_ERR_STATE, "no frame rendered yet");
  const int W = c->P.eqr_width, H = c->P.eqr_height;
  if (F.cubeW != fw || F.cubeH != fh || F.cubeSrcW != W || F.cubeSrcH != H) {
    static const int faces[6] = {CUBE_RIGHT, CUBE_LEFT, CUBE_TOP, CUBE_BOTTOM, CUBE_BACK, CUBE_FRONT};
    std::vector<float> m((size_t)6 * fw * fh * 2);
    const float dy = 1.0f / float(fw), dx = 1.0f / float(fh);
    for (int f = 0; f < 6; ++f)
      for (int j = 0; j < fh; ++j)
        for (int i = 0; i < fw; ++i) {
          float* e = &m[(((size_t)f * fh + j) * fw + i) * 2];
          cube_map_entry(float(i) * dy - 0.5f, float(j) * dx - 0.5f, faces[f], W, H, (float)M_PI, e, e + 1);
        }
    F.cubeMaps.ensure(m.size() * sizeof(float));
    S360_HIP(hipMemcpyAsync(F.cubeMaps.p, m.data(), m.size() * sizeof(float), hipMemcpyHostToDevice, c->st));
    S360_HIP(hipStreamSynchronize(c->st));  // `m` is freed on return
    F.cubeW = fw; F.cubeH = fh; F.cubeSrcW = W; F.cubeSrcH = H;
  }
  *ow = video ? 3 * fw : fw;
  *oh = video ? 4 * fh : 12 * fh;
  F.cubeOut.ensure((size_t)*ow * *oh * 3);
  ProfScope ps(c->prof, "cubemap");
  launch_cubemap(c->st, F.pano[0].as<uchar4>(), F.pano[1].as<uchar4>(), W, H, F.cubeMaps.as<float2>(), fw, fh, video ? 1 : 0,
                 F.cubeOut.as<uint8_t>(), F.tab.dev);
}

}  // namespace s360
