// render_kernels.hpp — launchers of the warp / blend / composite HIP kernels (render_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace s360 {

struct DevCamera {  // Camera (SR/render/Camera.h:87-99) as kernel argument
  int type;
  double pos[3], R[9], principal[2], focal[2], distortion[2];
};
// Lookup tables shared by the warp/blend kernels; built on the host (tables.cpp), resident in HBM.
struct DevTables {
  const short* bicubic_i;   // [1024][16] remap weights, sum == 32768
  // the same table as what it is made of (initInterTab2D): the 1-D cubic taps [32][4] — x's pre-multiplied by 32768 in the second
  // half — and, per entry, the tap (2 bits: (2,2) (2,3) (3,2) (3,3)) that took the entry's rounding residue (upper 14 bits,
  // signed). nullptr when the host's rebuild of all 1024 entries from them did not reproduce bicubic_i.
  const float* bicubic_w1;  // [2][32][4]
  const short* bicubic_res; // [1024]
  const float* bicubic_f;   // [1024][16]
  const float* tanh10;      // [766] tanhf(10 * (s/255)), s = sum |dBGR|   (NovelView.cpp:131-138)
  const float* tanh5;       // [766] tanhf(5 * (s/255))                    (CvUtil.cpp:236-242)
  const float* flat_softmaxL;  // [256] softmaxL of flattenLayersDeghostPreferBase by top alpha (CvUtil.cpp:243-250)
};
struct NovelViewParams {  // TRSP:259-292 + NovelView.cpp:174-268
  int overlapW, camH, stripW, numNovelViews;
  int numPairs;   // P: strip layout [eye][P][camH][stripW]
  int numLocal;   // pairs held by the local overlap/flow arrays: L at j, R at numLocal + j
  float camImageWidthHalf;  // float(camImageWidth) * 0.5f
  float disp;               // vergeAtInfinitySlabDisplacement
};
struct PoleWarpParams {  // TRSP:483-536
  int cols, rows, extW, maxBlendX;
  float poleCameraRadius, phiRampStart, phiMid, phiRampEnd;
};

// generic
void launch_bgr_to_bgra(hipStream_t st, const uint8_t* src, int channels, uchar4* dst, size_t n);
void launch_u16_high_byte(hipStream_t st, const unsigned short* src, uint8_t* dst, size_t n);
// projectSideToSpherical's source preparation: BGR(A) -> BGRA with the top/bottom alpha ramp (TRSP:108-125)
void launch_prepare_side_src(hipStream_t st, const uint8_t* src, int channels, uchar4* dst, int w, int h, int feather);
// bicubicRemapToSpherical's warp map (ImageWarper.cpp:151-167); trig tables per column / row are host-built.
void launch_spherical_map(hipStream_t st, float2* map, int dw, int dh, const DevCamera& cam, const float* cosX,
                          const float* sinX, const float* cosY, const float* sinY);
// remap INTER_CUBIC / BORDER_CONSTANT(0) of a BGRA image through a float2 map. alpha_mode 0: keep interpolated
// alpha; 1: pole rule (alpha = 255 above yFeatherStart, 255*ramp below; TRSP:669-678, 629-637).
void launch_remap_cubic_u8c4(hipStream_t st, const uchar4* src, int sw, int sh, const float2* map, uchar4* dst, int dw,
                             int dh, const DevTables& T, int alpha_mode, int yFeatherStart, int featherSize,
                             int batch = 1 /* images of identical geometry: sources sw*sh, maps and outputs dw*dh apart */);
// The same through coordinates prepared once per map (render_kernels.hip "packed bicubic remap"): launch_remap_pack_map
// fills `packed` (one dword per destination pixel) and `tiles` (remap_packed_tiles(dw, dh) x 16 bytes per image) from a
// float map; the remap then reads those instead of the map (the map is only touched by tiles whose source box does not
// fit in LDS).
size_t remap_packed_tiles(int dw, int dh);
void launch_remap_pack_map(hipStream_t st, const float2* map, int sw, int sh, int dw, int dh, unsigned* packed, void* tiles,
                           int batch = 1);
void launch_remap_cubic_u8c4_packed(hipStream_t st, const uchar4* src, int sw, int sh, const float2* map, const unsigned* packed,
                                    const void* tiles, uchar4* dst, int dw, int dh, const DevTables& T, int alpha_mode,
                                    int yFeatherStart, int featherSize, int batch = 1);
// overlap crops (TRSP:196-198) for pairs [p0,p1): out[j] = right part of proj p0+j, out[n+j] = left part of
// proj (p0+j+1)%P, n = p1-p0
void launch_crop_overlaps(hipStream_t st, const uchar4* proj, int camW, int camH, int P, int overlapW, uchar4* out,
                          int p0, int p1);
// fused renderLazyNovelView x4 + combineLazyViews x2 for pairs [p0,p1): strips[eye][pair][camH][stripW]
void launch_novel_view(hipStream_t st, const uchar4* overlaps, const float2* flows /*[2n]: LtoR[n], RtoL[n]*/,
                       uchar4* strips, const NovelViewParams& nv, int p0, int p1, const DevTables& T);
// stackHorizontal + offsetHorizontalWrap + padToheight (TRSP:380-384, 806-807) for one eye
void launch_assemble_pano(hipStream_t st, const uchar4* strips_eye, int P, int camH, int stripW, float offset,
                          uchar4* pano, int eqrW, int eqrH);
// first `rows` rows of the image flipped around both axes (flip(img, img, -1))
void launch_flip_both(hipStream_t st, const uchar4* src, uchar4* dst, int w, int h, int rows);
// featherAlphaChannel pieces (CvUtil.cpp:140-157) on the top `rows` rows of a pano
void launch_erode_alpha(hipStream_t st, const uchar4* img, uint8_t* out, int w, int h, int e);
void launch_gauss_u8(hipStream_t st, const uint8_t* a, uint8_t* out, int w, int h, const int* ik, int r);
// extended (x % cols wrapped) flow inputs (TRSP:399-411): side image takes the feathered alpha plane
void launch_extend_wrap(hipStream_t st, const uchar4* img, const uint8_t* alpha /*nullable*/, int cols, int rows,
                        uchar4* ext, int extW);
// pole warp map + remap (TRSP:487-503)
// as two kernels: coordinates + tile boxes of this frame's warp (k_remap_pack), then the packed remap
void launch_pole_warp_packed(hipStream_t st, const uchar4* extFisheye, const float2* flow, uchar4* warpedExt,
                             const PoleWarpParams& pw, const DevTables& T, unsigned* packed /* extW*rows */,
                             void* tiles /* remap_packed_tiles(extW, rows) * 16 bytes */);
// seam blend + alpha ramp + bottom padding (TRSP:505-546): out is eqrW x eqrH
void launch_pole_finish(hipStream_t st, const uchar4* warpedExt, uchar4* out, int eqrH, const PoleWarpParams& pw);
// flattenLayersDeghostPreferBase (CvUtil.cpp:224-260); flip_top: top layer is indexed (W-1-x, H-1-y)
void launch_flatten(hipStream_t st, const uchar4* base, const uchar4* top, uchar4* out, int w, int h, int flip_top,
                    const DevTables& T);
// BGRA rows -> packed BGR at row offset (stackVertical + BGRA2BGR, TRSP:890-894, 960)
// stereo cubemap from the two eye panoramas through cached face warp maps (TRSP:917-935)
void launch_cubemap(hipStream_t st, const uchar4* eyeL, const uchar4* eyeR, int sw, int sh, const float2* maps, int fw,
                    int fh, int video, uint8_t* out, const DevTables& T);
// pole removal pieces (SR/render/PoleRemoval.cpp:32-188, SR/util/CvUtil.cpp:201-222)
void launch_remap_by_flow(hipStream_t st, const uchar4* src, int w, int h, const float2* flow, uchar4* dst,
                          const DevTables& T);
void launch_red_mask(hipStream_t st, const uint8_t* bgr, uint8_t* red, size_t n);
void launch_circle_alpha(hipStream_t st, const uchar4* src, const uint8_t* red /*nullable*/, uchar4* dst, int w, int h,
                         float radius);
void launch_pole_removal_combine(hipStream_t st, uchar4* bottom, const uchar4* warped2, size_t n);
void launch_pack_bgr(hipStream_t st, const uchar4* src, int w, int h, uint8_t* dst);
// sharpen (Filter.h:40-127) on BGRA in place, lp scratch same size
int sharpen_max_images();  // images one launch_sharpen_many call takes
size_t sharpen_scratch_bytes(int w, int h);
void launch_sharpen_many(hipStream_t st, uchar4* const* imgs, uchar4* const* lps, float* const* scratch, int n, int w,
                         int h, float amount);
void launch_sharpen(hipStream_t st, uchar4* img, uchar4* lp, float* scratch, int w, int h, float amount);

}  // namespace s360
