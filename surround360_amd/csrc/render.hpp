// render.hpp — device-resident per-frame pipeline state (renderStereoPanorama, TRSP:716-972).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "core.hpp"
#include "ctx.hpp"
#include "png.hpp"
#include "render_kernels.hpp"

namespace s360 {

struct Tables {
  DevBuf bi, bf, t10, t5, fs, gik, bw1, bres;
  DevTables dev;
  int gauss_ksize = 0;
  void build(hipStream_t st, int std_alpha_feather_size);
};

// Warp map of bicubicRemapToSpherical for (camera, dst size, angles): built once, cached in HBM.
void build_spherical_map(s360_ctx* c, float2* map, int dw, int dh, const s360_camera& cam, float l, float r, float t,
                         float b);

// Buffers a frame only needs INSIDE one of its own kernel sequences — produced and consumed between two neighbouring
// launches of the same slot — are shared by all frame slots of a context (the slots of a batch take their per-slot kernels
// one after the other on one stream): the side projections (dead once the overlaps are cropped), the flipped panoramas
// and pole projections (dead once the extended flow inputs exist), the pole warp's intermediates and warped layers (dead
// once composited), the composite's ping-pong buffer, the resized eyes (dead once packed). 1.7 GB per 8K slot that 13 of
// 14 slots no longer hold: 2 x 16 slots fit where 2 x 14 did. With several slots the getters of these intermediates
// (s360_frame_get_u8 "projection", "top_spherical", "bottom_spherical", "pole_warped") show the slot rendered last.
struct SlotScratch {
  DevBuf proj;
  DevBuf panoFlip[2], panoTmp;
  DevBuf topSph, botSph;
  DevBuf warpedExt, poleWarped[4];
  const void* poleOwner[4] = {nullptr, nullptr, nullptr, nullptr};  // the slot (FrameState) whose frame poleWarped[u] holds: with the
                                                                    // split phases two slots could interleave (frame_composite checks)
  DevBuf warpPacked, warpTiles;  // this frame's pole warp as packed coordinates + tile boxes (launch_pole_warp_packed)
  DevBuf eyeFinal[2];
  DevBuf pngScratch;  // the device PNG encoder's per-band segments before they are gathered into a slot's file image (png.hip)
  // the sharpen passes' low-pass image and float scratch for ONE group of kSharpenGroup images: the eyes of a batch are
  // sharpened group after group on one stream (8 images = 4 slots' eyes are 2 waves per SIMD in the row passes: enough to
  // cover each other's dependent chains), so the 1.1 GB per 8K slot these were is 4.4 GB per context
  static constexpr int kSharpenGroup = 8;
  DevBuf sharpLp[kSharpenGroup], sharpBuf[kSharpenGroup];
};

struct FrameState {
  Tables tab;
  std::shared_ptr<SlotScratch> sc;  // the context's (frame_state())
  int P = 0;                       // number of side cameras / pairs
  int srcW = 0, srcH = 0;
  int topW = 0, topH = 0, poleW = 0, poleH = 0;  // top camera image; bottom camera image (poleW/poleH: also pole removal)
  unsigned long long side_uploaded = 0;          // bit i: side camera i has an image (cleared by nothing: images persist)
  bool have_side = false, have_top = false, have_bottom = false;
  DevBuf staging, sideSrc, topSrc, botSrc;
  DevBuf overlaps[2];  // [cur/prev] temporal double buffer of the flow INPUTS (the motion map needs this frame's and the previous one's)
  // The flows themselves are ONE buffer (round 5): PixFlow reads the previous flow once, at its entry (the downscale into the previous
  // flow's pyramid, PixFlow.h:103-104), and writes the new one once, at its end (the final upscale + blur) — in place on one stream.
  // 1.16 GB less per temporally chained 8K slot (side 0.48 + pole 0.68): two more streams per context.
  DevBuf sideFlows;
  int side_p0 = 0, side_p1 = 0;       // pairs held by overlaps/sideFlows (local partition)
  bool partition_declared = false;    // s360_frame_set_partition was called (set_prev_side then fills that block)
  DevBuf strips;                      // [2][P][camH][stripW]
  DevBuf pano[2];
  DevBuf a8a, a8b, gtmp;
  DevBuf extImgs[2], poleFlows;       // extImgs [cur/prev]; slots: ext 0-3 side units, 4 top fisheye, 5 bottom fisheye; poleFlows: in place

  // Stacked equirect of the last two frames (alternating): a streaming host downloads frame k from one buffer while
  // frame k+1 is composited into the other (s360_frame_download_equirect_of). outDone[i] is recorded behind the
  // kernels that fill outBGR[i].
  DevBuf outBGR[2];
  hipEvent_t outDone[2] = {nullptr, nullptr};
  // device snapshot of the three flow engines' sweep error words, copied on the render stream just in front of
  // outDone[i]; a download's own stream copies it out with the pixels (ctx downErr): s360_frame_download_equirect_of refuses a
  // frame whose sweeps timed out without waiting for the frame that renders behind it (the words are cumulative, so a
  // non-zero value may also come from that next frame's side flows — either way the stream's results are invalid from here on)
  DevBuf outErrDev[2];
  // s360_frame_download_equirect_of releases the context while it waits: downRead[i] is recorded on the download stream behind
  // its copy of outBGR[i] / outErrDev[i], and the finish stage that next writes buffer i waits for it (a feeder two frames ahead of
  // the fetching thread must not overwrite a frame that is still being transferred)
  hipEvent_t downRead[2] = {nullptr, nullptr};
  // s360_set_png_encode: the frame in outBGR[i] also as a PNG file image (band chunks at their final offsets, png.hip) and its
  // band table, written by the finish stage in front of outDone[i]; pngFrame[i] = the frame (frames_done) they belong to
  DevBuf pngFile[2], pngMeta[2];
  PngPlan pngPlan[2];
  long long pngFrame[2] = {-1, -1};
  int out_cur = 0;           // buffer of the most recently ENQUEUED frame
  long long frames_done = 0;  // frames enqueued so far
  long long poleFrame[4] = {-1, -1, -1, -1};  // the frame (value of frames_done) pole unit u's warped layer was computed / received for
  ~FrameState() {
    for (auto& e : outDone) if (e) (void)hipEventDestroy(e);
    for (auto& e : downRead) if (e) (void)hipEventDestroy(e);
  }
  int cur_side = 0, cur_pole = 0, last_side = 0, last_pole = 0;
  bool have_prev_side = false, have_prev_pole = false;
  bool keep_intermediates = false;  // copy panoramas before the pole composite (parity tests)
  DevBuf panoDbg[2];
  int extW = 0, poleRowsT = 0, poleRowsB = 0;  // geometry the pole temporal state was produced with
  size_t extStride = 0;                        // pixels between the slots of extImgs / poleFlows (extW * max rows)
  // pole removal: secondary bottom source (BGRA), red-mask planes, flow inputs [cur/prev][2][n], flow [cur/prev][n]
  DevBuf botSrc2, prRed[2], prImgs[2], prFlow[2], prTmp, prWarp, prMerged;
  bool have_pr_inputs = false, have_prev_pr = false;
  int last_pr = 0;
  DevBuf cubeMaps, cubeOut;  // cached face warp maps [6][fh][fw] float2 and the stacked BGR cubemap
  int cubeW = 0, cubeH = 0, cubeSrcW = 0, cubeSrcH = 0;
};

FrameState& frame_state(s360_ctx* c);
void frame_upload_side(s360_ctx* c, int idx, const uint8_t* img, int w, int h, int ch);
void frame_upload_pole(s360_ctx* c, bool top, const uint8_t* bgr, int w, int h);
void frame_upload_raw(s360_ctx* c, struct s360_isp* isp, int which, const void* raw, int bits, int inW, int inH);
void frame_upload_pole_removal(s360_ctx* c, const uint8_t* bottom2, const uint8_t* mask, const uint8_t* mask2, int w, int h);
void frame_render_pairs(s360_ctx* c, int p0, int p1, int use_prev);
void frame_finish(s360_ctx* c, int pole_mask, int use_prev);
// the two halves of frame_finish for a frame whose pole units are spread over GPUs
void frame_pole_units(s360_ctx* c, int pole_mask, int use_prev);
void frame_composite(s360_ctx* c, int pole_mask);
// all slots at once: per-frame kernels slot by slot, the side flows of all slots in one FlowEngine batch, the pole
// flows of all slots in another
void frame_render_batch(s360_ctx* c, int use_prev);
void frame_render_slots(s360_ctx* c, const int* slots, int n, int use_prev);  // ... a subset of them (ascending, distinct)
void set_frame_slots(s360_ctx* c, int n);
// stereo cubemap of the last finished frame into F.cubeOut; returns its width/height through ow/oh
void frame_cubemap(s360_ctx* c, int face_w, int face_h, bool video, int* ow, int* oh);

// RCCL strip gather of the sharded frame (comm.cpp)
void comm_unique_id(void* id128);
const char* comm_library_path();  // the file the RCCL entry points were resolved from (S360_RCCL_LIB overrides the search)
void comm_init_rank(s360_ctx* c, const void* id128, int rank, int nranks);
void comm_init_all(s360_ctx* const* ctxs, int n);
void comm_destroy(s360_ctx* c);
int comm_size(s360_ctx* c);  // ncclCommCount of the context's communicator (0: none)
int comm_rank(s360_ctx* c);  // ncclCommUserRank (-1: none)
void frame_gather_strips(s360_ctx* c, const int* bounds, int root);
void frame_exchange_strips(s360_ctx* c, const int* bounds, const int* need_mask);
void frame_gather_pole_layers(s360_ctx* c, const int* owner, int root);
void comm_loopback(s360_ctx* c, int src_pair, int dst_pair);

// operator-level helpers on device buffers
// erode_size < 0: the context's std_alpha_feather_size and its cached Gaussian taps; otherwise `taps` (device, erode_size ints)
void dev_feather_alpha_to_ext(s360_ctx* c, const uchar4* pano_top_rows, int cols, int rows, uchar4* ext, int extW,
                              int erode_size = -1, const int* taps = nullptr);
std::vector<int> feather_gauss_taps(int erode_size);
void dev_pole_unit_post(s360_ctx* c, const uchar4* extFisheye, const float2* flow, int cols, int rows, int extW,
                        uchar4* warped_out /*cols x eqrH*/, int eqrH);

// api.hip: does [p, p + bytes) lie in a buffer from s360_host_alloc (page-locked: uploads need no staging copy)?
bool host_is_pinned(const void* p, size_t bytes);

// The context's flow engines (created on first use). which: 0 side / operator-level, 1 pole, 2 pole removal. Without frame
// pipelining the three never compute at the same time (one stream) and share one set of device buffers (flow.hpp FlowBufs);
// with it the finish stage runs on its own stream, and the pole / pole-removal engines get a set of their own.
FlowEngine& flow_engine(s360_ctx* c, int which);
void flow_engines_follow_pipelining(s360_ctx* c);  // after c->pipeline changed (streams idle)

}  // namespace s360
