/*
 * s360.h — C ABI of the MI355X-native stereo-panorama hot path (libs360.so).
 *
 * Drop-in boundary for surround360_render's per-frame render path. Every entry point
 * names the reference interface it replaces (paths relative to the reference repo,
 * SR/ = surround360_render/source/). Plain C types only: pointers, sizes, POD structs.
 * All functions return S360_OK (0) or a negative error code; s360_last_error() gives the
 * message (the reference throws VrCamException / CHECK-aborts instead — the host binary
 * maps a non-zero code back to that behaviour).
 *
 * Device: the library drives gfx950 through HIP. There is NO CPU fallback: if no HIP
 * device is usable every compute entry point fails with S360_ERR_NO_DEVICE.
 *
 * Image layouts are the reference's cv::Mat layouts: row-major, channel-interleaved,
 * 8-bit BGR / BGRA (CV_8UC3 / CV_8UC4), float32 x2 for flow (CV_32FC2: fx, fy).
 * "Host" pointers are ordinary memory; "dev" pointers are HIP device pointers (e.g.
 * torch.Tensor.data_ptr()) and must be on the context's device.
 */
#ifndef S360_H_
#define S360_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define S360_OK 0
#define S360_ERR_INVALID_ARG (-1)
#define S360_ERR_NO_DEVICE (-2)
#define S360_ERR_HIP (-3)
#define S360_ERR_UNKNOWN_ALG (-4) /* makeOpticalFlowByName throws, SR/optical_flow/OpticalFlowFactory.h:63 */
#define S360_ERR_IO (-5)
#define S360_ERR_STATE (-6)

/* OpticalFlowInterface::DirectionHint, SR/optical_flow/OpticalFlowInterface.h:24 */
enum { S360_HINT_UNKNOWN = 0, S360_HINT_RIGHT = 1, S360_HINT_DOWN = 2, S360_HINT_LEFT = 3, S360_HINT_UP = 4 };
/* Camera::Type, SR/render/Camera.h:87 */
enum { S360_CAM_FTHETA = 0, S360_CAM_RECTILINEAR = 1 };

/* One camera of the rig after Camera(const dynamic& json) (SR/render/Camera.cpp:44-83):
 * rotation is the re-unitarised row-major matrix with rows right/up/backward. */
typedef struct s360_camera {
  int32_t type;
  int32_t is_side; /* group contains "side" (SR/render/RigDescription.cpp:20-24) */
  double position[3];
  double rotation[9];
  double resolution[2];
  double principal[2];
  double distortion[2];
  double focal[2];
  double fov_threshold; /* cos(fov)*|cos(fov)|, SR/render/Camera.h:98 */
  char id[32];
} s360_camera;

/* The gflags of SR/test/TestRenderStereoPanorama.cpp:44-70 that influence pixels. */
typedef struct s360_params {
  double interpupilary_dist;       /* 6.4 */
  double zero_parallax_dist;       /* 10000 */
  double sharpening;               /* 0.0 */
  int32_t side_alpha_feather_size; /* 100 */
  int32_t std_alpha_feather_size;  /* 31 */
  int32_t enable_top, enable_bottom;
  int32_t eqr_width, eqr_height;
  int32_t final_eqr_width, final_eqr_height;
  char side_flow_alg[32];  /* "pixflow_low" | "pixflow_search_20" */
  char polar_flow_alg[32];
  int32_t enable_pole_removal;    /* TRSP:58: merge the two bottom cameras to erase the tripod pole (needs the secondary
                                     bottom image and the red pole masks: s360_frame_upload_pole_removal) */
  char poleremoval_flow_alg[32];  /* "" = "pixflow_low" */
} s360_params;

/* Derived sizes (SR/test/TestRenderStereoPanorama.cpp:153-173, 309-348, 656-659). */
typedef struct s360_geometry {
  int32_t cam_image_width, cam_image_height; /* side spherical projection */
  int32_t overlap_image_width, num_novel_views;
  int32_t top_rows, bottom_rows;   /* pole spherical heights */
  int32_t out_width, out_height;   /* stacked stereo equirect */
  float h_radians, v_radians, fov_horizontal_radians;
  float verge_at_infinity_slab_displacement, zero_parallax_novel_view_shift_pixels;
} s360_geometry;

typedef struct s360_ctx s360_ctx;

/* ---- library / device ----------------------------------------------------------------- */
const char* s360_version(void);
int s360_device_count(void);
const char* s360_last_error(const s360_ctx* ctx); /* ctx may be NULL: last error of this thread */

/* ---- rig loading (host; replaces Camera::loadRig + RigDescription ctor,
 *      SR/render/Camera.cpp:243-254, SR/render/RigDescription.cpp:18-31) -------------------- */
/* Parses a rig JSON file (RIG_JSON.md). Writes up to max_cams cameras, returns the count (<0 on error). */
int s360_rig_load_json(const char* path, s360_camera* cams, int max_cams);
/* Camera(const dynamic& json) from already-parsed vectors; principal/distortion/fov may be NULL. */
int s360_camera_init(s360_camera* out, int type, const double origin[3], const double forward[3], const double up[3],
                     const double right[3], const double resolution[2], const double* principal,
                     const double* distortion, const double focal[2], const double* fov, const char* group,
                     const char* id);
/* Camera::pixel (SR/render/Camera.h:133-140), Camera::getFov (Camera.cpp:150-154), host side. */
void s360_camera_pixel(const s360_camera* cam, const double rig_point[3], double pixel_out[2]);
double s360_camera_get_fov(const s360_camera* cam);
/* RigDescription::findCameraByDirection(+Z / -Z) (SR/render/RigDescription.cpp:33-47); index or <0. */
int s360_rig_find_top(const s360_camera* cams, int n);
int s360_rig_find_bottom(const s360_camera* cams, int n);
/* RigDescription::findLargestDistCamAxisToRigCenter (RigDescription.cpp:46-54): the secondary bottom camera. */
int s360_rig_find_bottom2(const s360_camera* cams, int n);
/* Camera::approximateUsablePixelsRadius (SR/render/Camera.h:201-212). */
float s360_camera_usable_pixels_radius(const s360_camera* cam);

/* Derived panorama geometry for a rig + flag set (host only; what s360_create computes and caches). */
int s360_derive_geometry(const s360_camera* cams, int n_cams, const s360_params* params, s360_geometry* out);
/* Pole ramp constants of poleToSideFlowThread (TRSP:454-481): poleCameraRadius, phiRampStart, phiMid, phiRampEnd
 * (degrees). Host only. */
int s360_pole_ramp(const s360_camera* cams, int n_cams, float out4[4]);

/* ---- context: one per device; owns streams, persistent HBM buffers, cached warp maps --------
 * Thread safety: every entry point that takes a context locks it for the duration of the call, so any number of host
 * threads may call into ONE context concurrently — the 14 std::threads of TRSP:320-335, each with "its own" flow
 * operator (NovelView.cpp:281-298), may share it; their calls execute one after the other. Calls on different contexts
 * run in parallel. For throughput hand the pairs over together (s360_compute_optical_flow_batch, s360_frame_render):
 * 14 serialised single-pair calls cost 14 flow latencies. With concurrent callers read a failed call's message with
 * s360_last_error(NULL) (the calling thread's own last error); s360_last_error(ctx) is the context's most recent one. */
/* cams: the whole rig (side cameras in rig order + pole cameras), as RigDescription holds it.
 * Flags whose frame cannot exist do not fail here — the operator-level entry points (flows, remaps, blends) do not
 * depend on the frame geometry — but every s360_frame_* / frame-slot call of such a context fails with
 * S360_ERR_INVALID_ARG and the reason (the reference runs into OpenCV's assertions inside the frame and aborts):
 * eqr_width / eqr_height outside 1..65536, final_eqr_width / final_eqr_height outside 0..65536 (0 = no final resize) or
 * leaving fewer than two output rows, a side projection / overlap / strip of zero pixels (s360_geometry), and flow
 * images — overlap_image_width x cam_image_height for the sides, (eqr_width x 1.2) x pole rows for an enabled pole —
 * below 2 x 2 pixels after the algorithm's entry downscale (4 x 4 for pixflow_low): the same floor
 * s360_compute_optical_flow has. Unknown algorithm names fail here: S360_ERR_UNKNOWN_ALG. */
int s360_create(s360_ctx** out, int device, const s360_camera* cams, int n_cams, const s360_params* params);
void s360_destroy(s360_ctx* ctx);
int s360_get_geometry(const s360_ctx* ctx, s360_geometry* out);
/* The HIP stream all of this context's work is enqueued on (hipStream_t as void*). */
void* s360_stream(s360_ctx* ctx);
int s360_synchronize(s360_ctx* ctx);
/* FLAGS_sharpening (TRSP:56, read at TRSP:901) is an ordinary run-time flag of the reference: the frames rendered after
 * this call are sharpened by `sharpening` (0 = off) instead of s360_params.sharpening. */
int s360_set_sharpening(s360_ctx* ctx, double sharpening);

/* ---- operator level (host pointers in/out; each call uploads, runs on the GPU, downloads) --- */
/* OpticalFlowInterface::computeOpticalFlow via makeOpticalFlowByName(alg)
 * (SR/optical_flow/OpticalFlowInterface.h:34-41, OpticalFlowFactory.h:23-64, PixFlow.h:81-183).
 * prev_* == NULL means "empty Mat" (no temporal regularisation). flow_out: w*h*2 floats. */
int s360_compute_optical_flow(s360_ctx* ctx, const char* alg, const uint8_t* i0_bgra, const uint8_t* i1_bgra, int w,
                              int h, const float* prev_flow, const uint8_t* prev_i0_bgra,
                              const uint8_t* prev_i1_bgra, int hint, float* flow_out);
/* Same, `batch` independent pairs of identical size in one launch sequence (how the 14 side pairs x 2
 * directions and the 4 pole flows are issued). Arrays are contiguous [batch][h][w][c]. */
int s360_compute_optical_flow_batch(s360_ctx* ctx, const char* alg, int batch, const uint8_t* i0_bgra,
                                    const uint8_t* i1_bgra, int w, int h, const float* prev_flow,
                                    const uint8_t* prev_i0_bgra, const uint8_t* prev_i1_bgra, int hint,
                                    float* flow_out);
/* bicubicRemapToSpherical (SR/render/ImageWarper.cpp:143-174): dst_channels 3 or 4 decides BGR2BGRA. */
int s360_bicubic_remap_to_spherical(s360_ctx* ctx, uint8_t* dst, int dst_w, int dst_h, int dst_channels,
                                    const uint8_t* src, int src_w, int src_h, int src_channels,
                                    const s360_camera* camera, float left_angle, float right_angle, float top_angle,
                                    float bottom_angle);
/* The float warp map of the same call (pixel - 0.5 per dst pixel), for parity tests. map_out: dst_w*dst_h*2. */
int s360_spherical_warp_map(s360_ctx* ctx, float* map_out, int dst_w, int dst_h, const s360_camera* camera,
                            float left_angle, float right_angle, float top_angle, float bottom_angle);
/* NovelViewGeneratorLazyFlow::combineLazyNovelViews with the LazyNovelViewBuffer of
 * renderStereoPanoramaChunksThread (SR/optical_flow/NovelView.cpp:226-268, TRSP:259-292).
 * image_l/r: overlap images (overlap_image_width x cam_image_height BGRA), flows same size.
 * chunk_l/r out: (eqr_width/n_side) x cam_image_height BGRA. */
int s360_combine_lazy_novel_views(s360_ctx* ctx, const uint8_t* image_l, const uint8_t* image_r,
                                  const float* flow_l_to_r, const float* flow_r_to_l, uint8_t* chunk_l,
                                  uint8_t* chunk_r);
/* flattenLayersDeghostPreferBase (SR/util/CvUtil.cpp:224-260). BGRA in, BGRA out. */
int s360_flatten_layers_deghost_prefer_base(s360_ctx* ctx, const uint8_t* bottom_layer, const uint8_t* top_layer, int w,
                                            int h, uint8_t* out);
/* offsetHorizontalWrap (SR/util/CvUtil.cpp:93-115). channels 3 (BGR) or 4 (BGRA). */
int s360_offset_horizontal_wrap(s360_ctx* ctx, const uint8_t* src, int w, int h, int channels, float offset,
                                uint8_t* out);
/* featherAlphaChannel (SR/util/CvUtil.cpp:140-157). BGRA. erode_size odd (it is also the GaussianBlur kernel size)
 * and <= 31; the context's std_alpha_feather_size uses the cached taps and the fixed-radius kernels. */
int s360_feather_alpha_channel(s360_ctx* ctx, const uint8_t* src, int w, int h, int erode_size, uint8_t* out);
/* poleToSideFlowThread (SR/test/TestRenderStereoPanorama.cpp:388-561) without temporal state:
 * side: eqr_width x eqr_height BGRA eye panorama, pole: eqr_width x pole_rows BGRA.
 * warped_out: eqr_width x eqr_height BGRA. flow_out (nullable): extendedWidth x pole_rows x 2. */
int s360_pole_to_side_flow(s360_ctx* ctx, const uint8_t* side, const uint8_t* pole, int pole_rows, uint8_t* warped_out,
                           float* flow_out);
/* sharpenThread (TRSP:688-696, SR/util/Filter.h:40-127), in place on BGR. */
int s360_sharpen(s360_ctx* ctx, uint8_t* bgr, int w, int h, float sharpening);

/* ---- frame level: renderStereoPanorama (SR/test/TestRenderStereoPanorama.cpp:716-972) ------- */
/* Inputs stay resident in HBM between upload and render (bench timing starts after upload). */
int s360_frame_upload_side(s360_ctx* ctx, int side_idx, const uint8_t* img, int w, int h, int channels);
int s360_frame_upload_top(s360_ctx* ctx, const uint8_t* bgr, int w, int h);
int s360_frame_upload_bottom(s360_ctx* ctx, const uint8_t* bgr, int w, int h);
/* --enable_pole_removal inputs (combineBottomImagesWithPoleRemoval, SR/render/PoleRemoval.cpp:32-188): the secondary
 * bottom camera's image and the red masks of both bottom cameras (pure BGR red (0,0,255) = pole), all BGR w x h. */
int s360_frame_upload_pole_removal(s360_ctx* ctx, const uint8_t* bottom2_bgr, const uint8_t* mask_bgr,
                                   const uint8_t* mask2_bgr, int w, int h);
/* Enqueue the whole frame on the context stream (asynchronous). use_prev != 0 applies the
 * temporal regularisation against the previous s360_frame_render's device-resident state
 * (the reference's --prev_frame_data_dir, TRSP:215-235, 421-436). The flows of that state are updated in place: a render
 * that FAILS (non-zero return) leaves the slot without temporal state — the next use_prev render of it runs as a first
 * frame until a render has completed, or state is handed in again with s360_frame_set_prev_side / _pole. */
int s360_frame_render(s360_ctx* ctx, int use_prev);
/* Sharded form for multi-GPU: render only side pairs [pair_begin, pair_end) into this
 * context's strip buffers; strips of other pairs are filled in by the caller (RCCL gather) through
 * s360_frame_strip_ptr before s360_frame_finish assembles panoramas, runs the pole units given by
 * pole_mask (bit0 top_left, bit1 top_right, bit2 bottom_left, bit3 bottom_right) and composites. */
int s360_frame_render_pairs(s360_ctx* ctx, int pair_begin, int pair_end, int use_prev);
/* Device pointer + byte size of the strip buffer of one eye: [n_side][cam_image_height][strip_w][4]. */
int s360_frame_strip_ptr(s360_ctx* ctx, int eye, void** dev_ptr, size_t* bytes_per_pair);
int s360_frame_finish(s360_ctx* ctx, int pole_mask, int use_prev);
/* Stacked stereo equirect (left eye over right eye), BGR, out_width x out_height (host / device). */
int s360_frame_download_equirect(s360_ctx* ctx, uint8_t* out_bgr);
int s360_frame_equirect_dev(s360_ctx* ctx, void** dev_ptr, size_t* bytes);
/* Streaming hosts (one video stream, the reference's per-frame loop of scripts/batch_process_video.py:123-210 inside
 * one process): the stacked equirect of the last two frames is kept. age 0 = the frame enqueued last, age 1 = the one
 * before (needs s360_set_frame_pipelining, which is what makes the library alternate between two output buffers); the
 * call waits for THAT frame only and copies on a stream of its own, so frame k can be fetched and encoded while frame
 * k+1 renders:  upload(k+1); render(k+1); download_equirect_of(age 1) -> frame k.
 * Contract for a feeder thread that runs ahead of the fetching thread (the call releases the context's lock while it waits): a
 * frame can be fetched while it is the last or the last-but-one ENQUEUED frame, i.e. the fetch of frame k must be CALLED before
 * frame k+2 is enqueued; from then on the library orders things itself — the frame that reuses k's output buffer (k+2) waits on
 * the device for k's transfer, and k's sweep error words travel with its pixels. One fetching thread per context. */
int s360_frame_download_equirect_of(s360_ctx* ctx, int age, uint8_t* out_bgr);
/* Two output buffers per frame slot WITHOUT frame pipelining: finished frames alternate
 * between them, so a host that renders batches — s360_frame_render_batch / _slots — can enqueue step k+1 first and then fetch
 * step k's frames (age 1) while k+1 renders, instead of leaving the GPU idle for the length of the fetch. Costs one more
 * output image (and PNG file image) per slot. */
int s360_set_output_double_buffer(s360_ctx* ctx, int on);
/* The same fetch with the frame slot NAMED instead of selected (s360_select_frame_slot is state of the context): the one fetching
 * thread of a batch host drains step k's slots while another thread selects slots for step k+2's uploads. */
int s360_frame_download_equirect_slot(s360_ctx* ctx, int slot, int age, uint8_t* out_bgr);
/* ---- the equirect as a PNG FILE, encoded on the device -------------------------------------------------
 * Replaces imwriteExceptionOnFail(FLAGS_output_equirect_path, ...) (TRSP:938-961; cv::imwrite's PngEncoder: 8-bit RGB, Sub filter,
 * zlib Z_BEST_SPEED + Z_RLE) for the output frame: with s360_set_png_encode(ctx, 1) every frame rendered from then on is also
 * filtered (Sub) and deflated (dynamic Huffman over literals + distance-1 matches, one independent segment per band of rows) by
 * HIP kernels behind its last kernel, and s360_frame_download_png copies the compressed file image instead of the 201 MB of
 * pixels of an 8K frame; the host side of the call adds what needs no pixel (signature, IHDR, the chunks' CRC-32 on
 * S360_PNG_CRC_THREADS threads, default 4, the combined Adler-32, IEND). `out` receives a complete PNG file (any reader; the
 * banded layout of host/png_io.hpp, whose reader inflates the bands in parallel) that decodes to exactly the bytes
 * s360_frame_download_equirect returns, as R,G,B. cap >= s360_frame_png_bound(ctx); page-locked memory (s360_host_alloc) makes
 * the copy a direct transfer. age / threading contract: as s360_frame_download_equirect_of (age 0 works without pipelining).
 * S360_ERR_STATE when that frame was rendered with the encoder off. */
int s360_set_png_encode(s360_ctx* ctx, int on);
size_t s360_frame_png_bound(s360_ctx* ctx);
int s360_frame_download_png(s360_ctx* ctx, int age, uint8_t* out, size_t cap, size_t* n_out);
int s360_frame_download_png_slot(s360_ctx* ctx, int slot, int age, uint8_t* out, size_t cap, size_t* n_out);
/* The same encoder as an operator: any 8-bit B,G,R image in host memory (rows contiguous) -> a PNG file in `out`
 * (cap >= s360_png_bound(w, h)); synchronous. */
size_t s360_png_bound(int w, int h);
int s360_encode_png(s360_ctx* ctx, const uint8_t* bgr, int w, int h, uint8_t* out, size_t cap, size_t* n_out);
/* Page-locked host buffers for streaming hosts (what the reference's per-frame loop has no need for: its cv::Mat pixels
 * never leave the host, RigDescription.cpp:80-108 / TRSP:961). An image passed to s360_frame_upload_* from such a buffer is
 * sent straight from it — no staging copy inside the library, the call returns in microseconds — and must then stay
 * untouched until s360_frame_uploads_complete(ctx) has returned; a download into such a buffer is one DMA transfer.
 * s360_frame_download_equirect_of and s360_frame_uploads_complete release the context's lock while they wait for the
 * device, so that a second host thread can upload and enqueue the next frame meanwhile (one fetching thread per context).
 * s360_host_alloc returns NULL on failure (s360_last_error(NULL) says why). */
void* s360_host_alloc(size_t bytes);
void s360_host_free(void* p);
int s360_frame_uploads_complete(s360_ctx* ctx);
/* Stereo cubemap of the last rendered frame (convertSphericalToCubemapBicubicRemap + stackOutputCubemapFaces,
 * SR/render/ImageWarper.cpp:95-141, SR/util/CvUtil.cpp:117-138, TRSP:917-935): format "video" (3 x 2 faces per eye,
 * flipped) or "photo" (6 faces stacked). whc receives width/height/3; out_bgr may be NULL for a size query. */
int s360_frame_cubemap(s360_ctx* ctx, int face_width, int face_height, const char* format, int whc[3], uint8_t* out_bgr);
/* Intermediates for stage-by-stage parity tests and for the reference's on-disk state
 * (overlap_<i>_{L,R}.png, flow{LtoR,RtoL}_<i>.bin, extended*Spherical_<eye>.png, flow_<eye>.bin).
 * Names: "projection"(idx cam) "overlap_l" "overlap_r" "side_pano_l" "side_pano_r" "top_spherical"
 * "bottom_spherical" "pole_warped"(idx 0..3) "extended_side" "extended_fisheye" "eye_l" "eye_r"
 * "bottom_image" "bottom_image2" (pole removal's flow inputs).
 * whc receives width/height/channels; dst may be NULL for a size query. */
int s360_frame_get_u8(s360_ctx* ctx, const char* name, int idx, int whc[3], uint8_t* dst);
/* "flow_l_to_r" "flow_r_to_l" (idx pair) "flow_pole" (idx 0..3) "flow_bottom_secondary". */
int s360_frame_get_f32(s360_ctx* ctx, const char* name, int idx, int whc[3], float* dst);

/* Temporal state from a previous process (the reference's --prev_frame_data_dir files, TRSP:215-235, 421-436):
 * previous flows (readFlowFromFile) and the exact 8UC4 images that went into them (overlap_<i>_{L,R}.png,
 * extendedSideSpherical_<eye>.png, extendedFisheyeSpherical_<eye>.png). Call for every pair / enabled pole unit
 * before s360_frame_render(ctx, use_prev = 1). Sizes: overlap_image_width x cam_image_height for the side pairs;
 * int(eqr_width * 1.2) x pole rows for the pole units (0 top_left, 1 top_right, 2 bottom_left, 3 bottom_right). */
int s360_frame_set_prev_side(s360_ctx* ctx, int pair_idx, const float* flow_l_to_r, const float* flow_r_to_l,
                             const uint8_t* overlap_l_bgra, const uint8_t* overlap_r_bgra);
int s360_frame_set_prev_pole(s360_ctx* ctx, int unit, const float* flow, const uint8_t* extended_side_bgra,
                             const uint8_t* extended_fisheye_bgra);
/* flow_bottom_secondary.bin + flow_images/bottomImage.png, bottomImage2.png (PoleRemoval.cpp:95-110), w x h of the
 * bottom cameras. */
int s360_frame_set_prev_pole_removal(s360_ctx* ctx, const float* flow, const uint8_t* bottom_image_bgra,
                                     const uint8_t* bottom_image2_bgra, int w, int h);

/* ---- frame slots: several independent frames through ONE launch sequence ----------------------------
 * The reference renders one frame per process and scales out by running processes side by side. On one MI355X a single
 * frame cannot fill the chip (PixFlow's sweeps are raster-order recurrences), so independent frames — the streams of a
 * multi-stream job, or the segments of an offline batch — are given to one context as slots: s360_set_frame_slots(n),
 * then per slot s360_select_frame_slot(k) + the usual uploads (and later getters / downloads), and ONE
 * s360_frame_render_batch renders all of them: per-frame kernels slot by slot, the side flows of every slot in one
 * batch of the flow kernels (n x 28 flows per launch), the pole flows of every slot in another (n x 4). Every slot's
 * result equals s360_frame_render on it. use_prev applies each slot's own device-resident temporal state (a batch
 * uses it only if every slot has one). s360_frame_render_slots is the same for a subset of the slots (ascending, distinct): the
 * streams of a job that still have a frame to render when the others have ended, or the one stream whose first frame resumes from
 * state files while the others start without (host/TestRenderStereoPanorama --num_streams on one GPU: the reference's
 * batch_process_video.py workload, frame k of every stream regularised toward its frame k-1, TRSP:215-235, 421-436, as ONE launch
 * sequence per step). */
int s360_set_frame_slots(s360_ctx* ctx, int n);
int s360_select_frame_slot(s360_ctx* ctx, int k);
int s360_frame_render_batch(s360_ctx* ctx, int use_prev);
int s360_frame_render_slots(s360_ctx* ctx, const int* slots, int n, int use_prev);

/* ---- multi-GPU: one frame sharded by side pairs, ONE RCCL exchange ------------------
 * Replaces the per-pair thread fan-out + join + stackHorizontal of TRSP:320-335, 354-384 when the pairs of a frame are
 * rendered on several GPUs: rank r renders the contiguous block of pairs [bounds[r], bounds[r+1]) with
 * s360_frame_render_pairs, s360_frame_gather_strips moves every block into the root's strip buffers (grouped
 * ncclSend/ncclRecv over xGMI, enqueued on s360_stream(): no host synchronisation), the root runs s360_frame_finish.
 * One context per rank and per GPU. Either one process per GPU (rank 0 calls s360_comm_get_unique_id, the host
 * distributes the 128 bytes, every rank calls s360_comm_init_rank) or one process driving all GPUs
 * (s360_comm_init_all over one context per device, then one host thread per context for the collective call). */
#define S360_COMM_ID_BYTES 128
int s360_comm_get_unique_id(void* id_out /* S360_COMM_ID_BYTES */);
int s360_comm_init_rank(s360_ctx* ctx, const void* id, int rank, int nranks);
int s360_comm_init_all(s360_ctx* const* ctxs, int n);
int s360_comm_destroy(s360_ctx* ctx);
/* The file the RCCL entry points were resolved from (librccl is loaded on first use: S360_RCCL_LIB=<path> in the environment
 * names it outright; otherwise by soname — a process that already holds an RCCL, e.g. torch's copy, shares it — then
 * /opt/rocm/lib). S360_RCCL_VERBOSE=1 prints it to stderr once. NULL (and s360_last_error) when no librccl can be loaded. */
const char* s360_comm_library_path(void);
/* What the context's COMMUNICATOR reports — ncclCommCount / ncclCommUserRank — not what the caller asked for: 0 / -1 when the
 * context has none (a context that renders whole frames never makes one); -1 on error (s360_last_error). */
int s360_comm_size(s360_ctx* ctx);
int s360_comm_rank(s360_ctx* ctx);
/* What the two exchanges have moved through this context since its communicator was made: which = 0
 * s360_frame_exchange_strips / _gather_strips, 1 s360_frame_gather_pole_layers; out = {calls, bytes sent, bytes received}.
 * Their device time is the "exchange_strips" / "exchange_pole_layers" families of s360_profile_get (HIP events on
 * s360_stream() around ncclGroupStart .. ncclGroupEnd). */
int s360_comm_stats(s360_ctx* ctx, int which, unsigned long long out[3]);
/* bounds: nranks + 1 non-decreasing pair indices from 0 to n_side. */
int s360_frame_gather_strips(s360_ctx* ctx, const int* bounds, int root);
/* The pole units on several GPUs as well (SURVEY §8e second stage; the reference runs its four poleToSideFlowThread
 * threads concurrently, TRSP:811-860). Per frame, on every rank, all enqueued on s360_stream():
 *   s360_frame_render_pairs(block of the rank)
 *   s360_frame_exchange_strips(bounds, need_mask)   need_mask[r]: eyes (bit 0 left, bit 1 right) whose complete strips
 *                                                   rank r needs: 3 for the root and for a rank that owns units of
 *                                                   both eyes, 1 << eye for the owner of one unit, 0 otherwise
 *   s360_frame_pole_units(pole_mask of the rank, use_prev)   panoramas + flow + warp of the rank's units (bit u:
 *                                                   0 top_left, 1 top_right, 2 bottom_left, 3 bottom_right); the
 *                                                   rank needs the pole image(s) of its units (s360_frame_upload_top /
 *                                                   _bottom) and keeps those units' temporal state
 *   s360_frame_gather_pole_layers(owner, root)      owner[u] = rank of unit u (-1: not enabled): the warped layers
 *                                                   (eqr_width x pole rows) travel to the root, one grouped exchange
 *   s360_frame_composite(mask of all enabled units) root only: flattenLayersDeghostPreferBase x4, sharpen, resize, stack
 * s360_frame_finish(mask) == s360_frame_pole_units(mask) + s360_frame_composite(mask) on one context. */
int s360_frame_exchange_strips(s360_ctx* ctx, const int* bounds, const int* need_mask /* [nranks] */);
int s360_frame_pole_units(s360_ctx* ctx, int pole_mask, int use_prev);
int s360_frame_gather_pole_layers(s360_ctx* ctx, const int owner[4], int root);
int s360_frame_composite(s360_ctx* ctx, int pole_mask);
/* Self-test for single-GPU boxes: one grouped ncclSend + ncclRecv of this rank to itself, eye-0 strip of pair
 * src_pair into the slot of dst_pair, on s360_stream(). */
int s360_comm_loopback(s360_ctx* ctx, int src_pair, int dst_pair);
/* Declares the block of pairs [pair_begin, pair_end) this context renders BEFORE previous-frame state is handed in with
 * s360_frame_set_prev_side (which then only accepts pairs of the block). Default: all pairs. */
int s360_frame_set_partition(s360_ctx* ctx, int pair_begin, int pair_end);

/* Keep copies of the eye panoramas as they are before the pole composite ("side_pano_l/r"); costs two
 * device copies per frame, off by default. */
int s360_set_keep_intermediates(s360_ctx* ctx, int on);
/* Test tap: the flow field after every pyramid level (coarsest first, concatenated) of one pair, i.e. the value
 * of `flow` at PixFlow.h:167 per level. levels_out capacity in floats. */
int s360_debug_flow_levels(s360_ctx* ctx, const char* alg, const uint8_t* i0_bgra, const uint8_t* i1_bgra, int w, int h,
                           int hint, float* levels_out, size_t cap_floats, int* n_levels);

/* Which sweep kernel PixFlow's propagation uses: "latency" (default) finishes ONE flow soonest; "throughput"
 * spends ~4x fewer instructions per pixel and is the right choice when several frames / contexts are in flight on
 * the GPU. Results are bit-identical. */
int s360_set_sweep_mode(s360_ctx* ctx, const char* mode);
/* Frame pipelining for a video stream (BASELINE configs[4]: "temporal-flow reuse and frame pipelining"). With it on,
 * s360_frame_finish (pole units, composite: TRSP:811-960) is enqueued on a second HIP stream and overlaps the side
 * stage (s360_frame_render_pairs: projection, flows, novel views, TRSP:320-384) of the NEXT frame; the three buffers
 * the two stages share are ordered by events inside the library. Results are unchanged. The same holds for batches of frame
 * slots (s360_frame_render_batch / _slots, round 6): batch k's pole stage runs beside batch k+1's side stage — the latency-bound
 * pole sweeps of one batch next to the wide side sweeps of the next. One context alone then renders 22 independent 8K frames per
 * batch in 21.4 instead of 22.7 ms per frame; it costs a second set of flow buffers and a second output buffer per slot (160
 * instead of 127 GB at 22 slots). Calls that return data
 * (download, get_*, cubemap) and s360_synchronize wait for both streams. The sharded (multi-GPU) frame's split phases
 * (s360_frame_exchange_strips / _gather_strips / _pole_units / _gather_pole_layers / _composite) enqueue their exchanges on
 * s360_stream() and are refused with S360_ERR_STATE while pipelining is on: nothing would order them against the second
 * stream (a stream keeps its temporal state on one GPU anyway). */
int s360_set_frame_pipelining(s360_ctx* ctx, int on);

/* ---- measurement -------------------------------------------------------------------------- */
/* Per-kernel-family device time of the last frame/flow call measured with HIP events on the context
 * stream, when enabled. names_out receives a ';'-separated list matching ms_out entries. */
int s360_profile_enable(s360_ctx* ctx, int on);
int s360_profile_get(s360_ctx* ctx, char* names_out, size_t names_cap, float* ms_out, int* launches_out, int cap);

/* ---- flow state file format (saveFlowToFile / readFlowFromFile, SR/util/CvUtil.cpp:159-199) -- */
int s360_save_flow_to_file(const char* path, const float* flow, int w, int h);
int s360_read_flow_from_file(const char* path, float* flow_out, int* w, int* h, size_t cap_floats);

/* ---- ISP: 16-bit Bayer raw -> BGR --------------------------------------
 * Both arithmetics of the reference (s360_isp_config.pipe): the soft CameraIsp described here and, from round 4, the
 * accelerated CameraIspPipe (camera_isp/CameraIspPipe.h, CameraIspGen.cpp).
 * pipe = 0 replaces the non-accelerated path of Raw2Rgb (SR/camera_isp/Raw2Rgb.cpp:441-456 -> CameraIsp.h): black level,
 * anti-vignetting, white balance, clamp + stretch, demosaic (bilinear or edge-aware), composite CCM + tone-curve LUT,
 * IIR unsharp mask, 8- or 16-bit output. Independent of s360_ctx (no rig is involved).
 * Not available: FREQUENCY_DM_FILTER (demosaic_filter 1; cv::dct) -> S360_ERR_INVALID_ARG.
 * Stuck-pixel removal (CameraIsp.h:1024-1104) with a non-zero radius: the reference's loop over the sorted region,
 *     for (int k = region.size() - 1; k <= region.size() - stuckPixelThreshold; k--)        (:1090-1092)
 * compares size_t values, so for 2 <= stuckPixelThreshold <= (stuckPixelRadius + 1)^2 (the population of a red / blue
 * region; the shipped configurations say 5) its condition is false at once and the pass changes nothing. Thresholds outside
 * that range make the pass a serial in-place median filter of the dark regions in boustrophedon order; both are reproduced
 * (pinned against CameraIsp.h compiled: tests/test_cpu_isp.py), the second as the raster-order recurrence it is — exact, and
 * as slow as a recurrence is where most pixels change. stuck_pixel_radius up to 6 (JSON 3: a 13 x 13 window).
 * COST of that second case (measured, 2048 x 2048): ~7.5 us per CHANGED pixel — one thread gathers and sorts a window behind the
 * previous one — i.e. 45 ms with 15 000 changed pixels, 2.2 s with a million. The reference's DEFAULT threshold, 0, is such a
 * case ("always write" wherever the neighbourhood is dark): a dark frame walks every pixel. The pass runs on the stream the
 * object is used on — s360_frame_upload_raw: the context's upload stream, which then stalls for that long. */
#define S360_ISP_MAX_CURVE_POINTS 16
typedef struct s360_isp_config {
  /* the "CameraIsp" JSON object as the constructor stores it (CameraIsp.h:425-607): doubles narrowed to float */
  float black_level[3], clamp_min[3], clamp_max[3], white_balance_gain[3];
  float ccm[9]; /* row-major 3x3 */
  float saturation, contrast;
  float gamma[3], low_key_boost[3], high_key_boost[3], sharpening[3];
  float sharpening_support, noise_core;
  int32_t n_vignette_h, n_vignette_v;
  float vignette_roll_off_h[S360_ISP_MAX_CURVE_POINTS][3], vignette_roll_off_v[S360_ISP_MAX_CURVE_POINTS][3];
  int32_t stuck_pixel_radius; /* 2 x the JSON value (CameraIsp.h:512) */
  int32_t bayer_pattern;      /* 0 RGGB, 1 GRBG, 2 GBRG (default), 3 BGGR */
  /* Raw2Rgb flags (Raw2Rgb.cpp:25-39) */
  int32_t output_bpp;         /* 8 or 16 */
  int32_t demosaic_filter;    /* 0 bilinear, 2 edge-aware (default) */
  int32_t resize;             /* 1, 2, 4, 8 */
  int32_t disable_tone_curve, black_level_offset;
  /* removeStuckPixels (CameraIsp.h:1024-1104), read when stuck_pixel_radius > 0 */
  int32_t stuck_pixel_threshold;
  float stuck_pixel_darkness_threshold;
  /* Which of the reference's two ISP arithmetics runs: 0 = CameraIsp (CameraIsp.h; Raw2Rgb without --accelerate), PINNED bit
   * for bit to the reference compiled here; 1 = CameraIspPipe (CameraIspPipe.h: the Halide pipeline of CameraIspGen.cpp —
   * what Unpacker.cpp:176-183 runs and Raw2Rgb --accelerate), 2 = its `fast` variant (bilinear demosaic, no vignetting, no
   * sharpening; Raw2Rgb --accelerate --fast). 1 and 2 are written from the generator's source and are NOT pinned: Halide
   * cannot be built here, so a real build may differ at rounding level (it may contract a * b + c).
   * With 1 / 2: resize must be 1, demosaic_filter and the stuck-pixel fields are not read (the pipeline has neither), and
   * only GBRG and RGGB exist — any other bayer_pattern runs as GBRG, as CameraIspPipe::runPipe does (CameraIspPipe.h:133-141). */
  int32_t pipe;
} s360_isp_config;
/* CameraIsp(json, output_bpp) defaults (CameraIsp.h:440-462) + Raw2Rgb's flag defaults. */
void s360_isp_config_defaults(s360_isp_config* cfg);
/* The constructor's reading of an ISP configuration (the text of e.g. res/config/isp/cmosis_fujinon.json): defaults,
 * then every key present under "CameraIsp". The Raw2Rgb flag fields keep the values they have in *cfg. */
int s360_isp_config_from_json(const char* json_text, s360_isp_config* cfg);
typedef struct s360_isp s360_isp;
/* Builds the composite CCM and the 4096-entry tone curve on the host (CameraIsp::setup, buildToneCurveLut) and uploads
 * them. One object per configuration; frames of any size can follow. */
int s360_isp_create(s360_isp** out, int device, const s360_isp_config* cfg);
void s360_isp_destroy(s360_isp* isp);
/* CameraIsp::loadImage + getImage(swizzle = true). raw16: h x w uint16 (host). out: (h / resize) x (w / resize) x 3,
 * B,G,R, uint8 or uint16 by output_bpp (host). */
int s360_isp_process(s360_isp* isp, const uint16_t* raw16, int w, int h, void* out_bgr);
/* The same from the sensor's packed bytes as Unpacker reads them from a .bin container (Unpacker.cpp:136-143;
 * RawConverter::convert8Frame / convert12Frame, RawConverter.cpp:15-59): bits 8 (w * h bytes) or 12 (3 * w / 2 bytes
 * per row, even w); widened to 16 bits on the device, then as s360_isp_process.
 * The reference's Unpacker runs CameraIspPipe(json, fast = false, 16 bits) (Unpacker.cpp:176-183): an ISP object created
 * with pipe = 1 runs that pipeline's arithmetic as restated from its generator (not pinned, see s360_isp_config.pipe); with
 * pipe = 0 the frames go through the soft CameraIsp arithmetic, pinned bit for bit to CameraIsp.h compiled from the
 * reference. host/Unpacker takes the first by default like the reference and the second with --soft_isp.
 * Not available with pipe = 0 (S360_ERR_INVALID_ARG): demosaic_filter 1 = FREQUENCY_DM_FILTER (CameraIsp.h:1175-1192, needs
 * cv::dct; the reference transforms planes it never initialised). */
int s360_isp_process_packed(s360_isp* isp, const uint8_t* frame, int bits, int w, int h, void* out_bgr);
/* The functions Halide generates from camera_isp/CameraIspGen.cpp — CameraIspGen8, CameraIspGen16, CameraIspGenFast8,
 * CameraIspGenFast16 (argument list CameraIspGen.cpp:704-712; called by CameraIspPipe::runPipe, CameraIspPipe.h:143-175) — as
 * one entry point: the reference's own FFI boundary of the accelerated ISP. EVERY parameter is the caller's, in the units the
 * generated functions take them (black levels in 16-bit counts, the two vignette tables and the truncated tone table as
 * CameraIspPipe::initPipe builds them, the composite CCM as CameraIsp::setup leaves it); `isp` (created with pipe != 0, any
 * configuration) only lends its device, stream and buffers. INTEGRATION.md section 3 shows the four generated
 * functions written over this call, with which the reference's unmodified CameraIspPipe.h and Unpacker.cpp compile and run.
 * Arithmetic: see s360_isp_config.pipe (not pinned). */
typedef struct s360_camera_isp_gen_args {
  const uint16_t* input;        /* height rows of input_stride uint16 (buffer_t stride[1]) */
  int32_t input_stride, width, height;
  const float* vignette_h;      /* [width][3]  (vignetteTableH(c, x)) */
  const float* vignette_v;      /* [height][3] */
  float black_level[3], white_balance_gain[3], clamp_min[3], clamp_max[3], sharpening[3]; /* R, G, B */
  float sharpening_support, noise_core;
  const float* ccm;             /* 3 x 3 row-major (ccm(i, j) = ccm[i + 3 j]) */
  const void* tone_table;       /* [4096][3] uint8 (output_bpp 8) or uint16 (16) */
  int32_t bgr;                  /* BGR: output channel c takes channel 2 - c */
  int32_t bayer_pattern;        /* 0 GBRG, 1 RGGB */
  int32_t fast, output_bpp;     /* which of the four functions */
  void* output;                 /* [height][width][3] uint8 / uint16, interleaved (set_stride(0, 3)) */
} s360_camera_isp_gen_args;
int s360_isp_pipe_generated(s360_isp* isp, const s360_camera_isp_gen_args* args);
/* A camera's raw Bayer frame through the ISP straight into a frame's source slot, on the context's upload stream and
 * without leaving the device — the reference's chain through files (Unpacker writes the ISP's 16-bit result as a PNG,
 * RigDescription::loadSideCameraImages / imread decodes it to 8 bits = its high byte). camera: side index, or
 * S360_CAMERA_TOP / S360_CAMERA_BOTTOM. The ISP object must have output_bpp 16 and live on the context's device; it may
 * be shared by all cameras of a rig that use one configuration (cameras of different resolutions included), but it
 * feeds ONE context: its kernels run on that context's upload stream over the object's own buffers, so a second context
 * using it is refused with S360_ERR_STATE — create one ISP object per context. Replaces s360_frame_upload_side / _top /
 * _bottom. */
#define S360_CAMERA_TOP (-1)
#define S360_CAMERA_BOTTOM (-2)
int s360_frame_upload_raw(s360_ctx* ctx, s360_isp* isp, int camera, const uint16_t* raw16, int w, int h);
/* The same from the sensor's packed bytes as a capture's .bin container holds them (BinaryFootageFile::getFrame; bits 8 or 12
 * as in s360_isp_process_packed): Unpacker's and the renderer's work on one frame of one camera without a file in between —
 * "the ISP feeding the GPU directly from .bin" (SURVEY.md 8f row 4). host/TestRenderStereoPanorama --bin_list does this for
 * every camera of a frame. */
int s360_frame_upload_packed(s360_ctx* ctx, s360_isp* isp, int camera, const uint8_t* frame, int bits, int w, int h);
/* The host-side tables of a configuration (no device needed; what s360_isp_create uploads): ccm9 = composite CCM x 4095
 * (CameraIsp::setup), lut = 4096 x 3 floats (buildToneCurveLut), curve_h = w x 3 and curve_v = h x 3 vignette gains of
 * a w x h frame (curveHAtPixel / curveVAtPixel; either may be NULL). */
int s360_isp_config_tables(const s360_isp_config* cfg, float* ccm9, float* lut, int w, int h, float* curve_h,
                           float* curve_v);

#ifdef __cplusplus
}
#endif
#endif /* S360_H_ */
