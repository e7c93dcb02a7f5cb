// flow_kernels.hip — HIP kernels of PixFlow::computeOpticalFlow for gfx950 (CDNA4).
//
// Reference: surround360_render/source/optical_flow/PixFlow.h:81-534. Every kernel is
// batched: blockIdx.z selects one of B independent flows (the 28 side flows, or the 4 pole
// flows, of one frame share their dimensions), buffers are [plane][B][h][w].
// All arithmetic is float32 in the reference's operation order (no FMA contraction, IEEE
// sqrt/div) so results are bit-identical to the CPU restatement.
//
// Layouts: BGRA images uchar4; grey/alpha planes float; gradients packed float2 (Ix,Iy) so a
// bilinear gather is two 16-byte loads; flow float2 (fx,fy).
#include "flow_kernels.hpp"

#include <stdexcept>

#include "devmath.hpp"

namespace s360 {

typedef float f4a8 __attribute__((ext_vector_type(4), aligned(8)));

// ------------------------------------------------------------------------------------------
// resize INTER_CUBIC 8UC4 (PixFlow.h:98-107 entry downscale; TRSP:938-957 final resize).
// OpenCV fixed point: short taps = round(w*2048); H pass int (HResizeCubic). V pass as the reference's x86-64 build
// runs it: VResizeCubicVec_32s8u (SSE2) covers whole groups of 8 row elements = 2 BGRA pixels and works in float —
// taps b*2^-22, float(int row sum) * tap added left to right, cvtps2dq (round-half-even), saturate; the scalar tail
// (the last pixel of an odd-width row) is FixedPtCast: (sum + 2^21) >> 22.
__global__ __launch_bounds__(256) void k_resize_cubic_u8c4(const uchar4* __restrict__ src, int sw, int sh,
                                                           size_t sbs, uchar4* __restrict__ dst, int dw, int dh,
                                                           size_t dbs, double scx, double scy,
                                                           const uchar4* const* __restrict__ src_tab) {
  // the double-precision source coordinate and the four 11-bit taps depend on the column (row) only: computed once
  // per block column / row instead of once per pixel
  __shared__ int s_sx[32], s_sy[8];
  __shared__ short s_ax[32][4], s_ay[8][4];
  const int tid = threadIdx.y * blockDim.x + threadIdx.x;
  if (tid < 40) {
    const bool col = tid < 32;
    const int k = col ? tid : tid - 32;
    const int d = col ? min((int)(blockIdx.x * 32 + k), dw - 1) : min((int)(blockIdx.y * 8 + k), dh - 1);
    int s0;
    float f, cb[4];
    resize_coord(d, col ? scx : scy, &s0, &f);
    cubic_coeffs(f, cb);
    if (col) s_sx[k] = s0; else s_sy[k] = s0;
#pragma unroll
    for (int q = 0; q < 4; ++q) (col ? s_ax[k] : s_ay[k])[q] = (short)sat_s16(cv_round(cb[q] * 2048.f));
  }
  __syncthreads();
  const int dx = blockIdx.x * blockDim.x + threadIdx.x;
  const int dy = blockIdx.y * blockDim.y + threadIdx.y;
  if (dx >= dw || dy >= dh) return;
  src = src_tab ? src_tab[blockIdx.z] : src + sbs * blockIdx.z;
  dst += dbs * blockIdx.z;
  const int sx = s_sx[threadIdx.x], sy = s_sy[threadIdx.y];
  int ax[4], ay[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { ax[k] = s_ax[threadIdx.x][k]; ay[k] = s_ay[threadIdx.y][k]; }
  // Horizontal pass (round 5): the four taps of a row and channel as two v_perm_b32 (byte -> 16-bit lanes of two pixels) + two
  // v_dot2_i32_i16 with the packed short taps, like the packed remap — 64 operations per pixel instead of 64 byte extractions + 64
  // 24-bit multiply-adds. Integer sums: same bits. (Measured: entry downscale 0.67 and finish 1.82 ms per frame before and after —
  // the compiler had folded the byte extractions into the multiply-adds' operands; kept for its shape, not for a gain.)
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  const s16x2 w01 = __builtin_bit_cast(s16x2, (unsigned)(ax[0] & 0xffff) | ((unsigned)ax[1] << 16));
  const s16x2 w23 = __builtin_bit_cast(s16x2, (unsigned)(ax[2] & 0xffff) | ((unsigned)ax[3] << 16));
  const int c0 = clip_idx(sx - 1, sw), c1 = clip_idx(sx, sw), c2 = clip_idx(sx + 1, sw), c3 = clip_idx(sx + 2, sw);
  int hx[4], hy[4], hz[4], hw[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const unsigned* S = reinterpret_cast<const unsigned*>(src + (size_t)clip_idx(sy - 1 + r, sh) * sw);
    const unsigned p0 = S[c0], p1 = S[c1], p2 = S[c2], p3 = S[c3];
    int h[4];
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
      const unsigned sel = 0x0c040c00u + ch * 0x00010001u;
      const s16x2 lo = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(p1, p0, sel));
      const s16x2 hi = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(p3, p2, sel));
      h[ch] = __builtin_amdgcn_sdot2(lo, w01, 0, false);
      h[ch] = __builtin_amdgcn_sdot2(hi, w23, h[ch], false);
    }
    hx[r] = h[0]; hy[r] = h[1]; hz[r] = h[2]; hw[r] = h[3];
  }
  uchar4 o;
  if (dx < (dw & ~1)) {  // SSE2-covered elements
    const float scale = 1.f / (2048 * 2048);
    const float b0 = (float)ay[0] * scale, b1 = (float)ay[1] * scale, b2 = (float)ay[2] * scale, b3 = (float)ay[3] * scale;
    auto vf = [&](const int* hh) {
      float s = (float)hh[0] * b0;
      s = s + (float)hh[1] * b1;
      s = s + (float)hh[2] * b2;
      s = s + (float)hh[3] * b3;
      return (unsigned char)sat_u8(cv_round(s));
    };
    o.x = vf(hx); o.y = vf(hy); o.z = vf(hz); o.w = vf(hw);
  } else {
    auto vi = [&](const int* hh) {  // |h| < 2^20, |a| < 2^12
      const int v = __mul24(hh[0], ay[0]) + __mul24(hh[1], ay[1]) + __mul24(hh[2], ay[2]) + __mul24(hh[3], ay[3]);
      return (unsigned char)sat_u8((v + (1 << 21)) >> 22);
    };
    o.x = vi(hx); o.y = vi(hy); o.z = vi(hz); o.w = vi(hw);
  }
  dst[(size_t)dy * dw + dx] = o;
}

// BGRA -> grey float in [0,1] (pre-blur) and alpha float (PixFlow.h:121-135).
// cvtColor BGRA2GRAY 8-bit: (B*1868 + G*9617 + R*4899 + 2^13) >> 14; "/= 255.0f" == * float(1/255.).
__global__ __launch_bounds__(256) void k_gray_alpha(const uchar4* __restrict__ src, size_t n, size_t sbs,
                                                    float* __restrict__ gray, float* __restrict__ alpha,
                                                    size_t pbs) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uchar4 p = src[sbs * blockIdx.z + i];
  const float inv255 = (float)(1.0 / 255.0);
  const int g = (p.x * 1868 + p.y * 9617 + p.z * 4899 + (1 << 13)) >> 14;
  gray[pbs * blockIdx.z + i] = (float)g * inv255;
  alpha[pbs * blockIdx.z + i] = (float)p.w * inv255;
}

// motion map for temporal regularisation (PixFlow.h:109-117): sum|dBGR| / (255*3), double then float.
__global__ __launch_bounds__(256) void k_motion(const uchar4* __restrict__ cur, const uchar4* __restrict__ prev,
                                                size_t n, size_t sbs, float* __restrict__ motion, size_t pbs) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uchar4 a = cur[sbs * blockIdx.z + i], b = prev[sbs * blockIdx.z + i];
  const double s = fabs((double)((int)a.x - (int)b.x)) + fabs((double)((int)a.y - (int)b.y)) +
                   fabs((double)((int)a.z - (int)b.z));
  motion[pbs * blockIdx.z + i] = (float)(s / (double)(255.0f * 3.0f));
}

// ------------------------------------------------------------------------------------------
// Separable symmetric Gaussian on CV_32F, BORDER_REFLECT_101, row pass then column pass (GaussianBlur;
// PixFlow.h:137-138, 363-366, 379-383, 439-443, 178-182). Evaluation order as OpenCV 3.1's filter engine has it:
// row pass of the 3- and 5-tap kernels k[c]*x[c] + sum_j k[c+j]*(x[c+j] + x[c-j]) (SymmRowSmallFilter), row pass of
// the 15-tap kernel left to right (generic RowFilter), column pass k[c]*x[c] + 0, then the symmetric pairs. One LDS tile with halo per workgroup; the row-pass result is
// rounded to float in LDS exactly like the intermediate image of the two-pass reference.
// Both passes are register-blocked: a thread produces 4 consecutive outputs along the filter axis from one
// sliding window of 4 + 2R inputs (the wide 15x15 kernels are LDS-bandwidth bound otherwise).
// SRC 0: the source is the image itself. SRC 1 (CN = 2): the source is Sobel(ksize 1, BORDER_REPLICATE) of a
//        single-channel plane, computed while the tile is loaded (PixFlow.h:356-366 without the intermediate image).
// SRC 2: the source is the INTER_LINEAR upscale (resize + scalar multiply, PixFlow.h:176-177) of a smaller image,
//        evaluated while the tile is loaded: the full-resolution intermediate of the final flow blur never exists.
// EPI 0: store. EPI 1: lowAlphaFlowDiffusion blend (PixFlow.h:444-453). EPI 3: store the sweep record
//        {I0x | NaN when the pixel is not updated, I0y, blurred.x, blurred.y} instead of the blurred flow (latency sweep kernel:
//        its bulk service wave shares a SIMD with a compute wave, and one 16-byte load per pixel is what it can afford —
//        with half-records a single 8K frame took 134.8 instead of 132.9 ms, profiles/r04_v3_frame_time_halfrec.txt).
//        EPI 2 (throughput sweep kernel): store the sweeps' half-record
//        {blurred.x | NaN when the pixel is not updated (PixFlow.h:390), blurred.y}; the other half of what a sweep reads per
//        pixel is I0's gradient, which the gradient kernel has already written (round 3 wrote a 16-byte record {I0x | NaN, I0y,
//        blurred} here and re-read the gradient to do so: 40 bytes of traffic per pixel-level for an 8-byte result, now 24).
// EPI 4 (round 5; temporally chained frames): EPI 1 followed by adjustFlowTowardPrevious (PixFlow.h:185-193) on the same
//        pixel — flow = flow * (1 - w) + prev * w with w = 1 - motion — and the previous flow's per-level rescale
//        (PixFlow.h:147-153) applied to the value as it is read. As passes of their own (k_scale_f32 over the previous flow's
//        pyramid, then k_adjust_toward_prev reading and rewriting the diffused flow at every level) these were 44 bytes of traffic
//        per pixel-level and 71 launches per flow batch for what is two multiply-adds in this kernel's epilogue; the same float
//        operations in the same order (Gp = the previous flow's level, recv = the motion level, up.post_scale = the level's factor).
// Tile: 64x16 outputs for the 3- and 5-tap kernels, 32x32 for the 15-tap ones (29 KB of LDS instead of 35 KB and 1.44x
// instead of 1.9x row-pass halo work: blur15 + diffusion 100 -> 69 ms of summed kernel time per frame with 16 frames in flight).
struct UpSrc {  // SRC 2: geometry of the small source image and the scalar applied after the resize
  int sw, sh;
  size_t sbs;
  double scx, scy;
  float post_scale;
};
template <int R, int CN, int EPI, int SRC, int SB_TW = 64, int SB_TH = 16, int NT = 256>
__global__ __launch_bounds__(NT) void k_sepblur(const float* __restrict__ src, float* __restrict__ dst, int w, int h,
                                                 size_t bs /*elements of CN floats per batch*/, BlurTaps taps,
                                                 const float* __restrict__ A, FlowIdx idx,
                                                 const float2* __restrict__ Gp, void* __restrict__ recv,
                                                 float* const* __restrict__ dst_tab, UpSrc up,
                                                 unsigned* __restrict__ rowflags) {
  constexpr int IW = SB_TW + 2 * R, IH = SB_TH + 2 * R;
  constexpr int IWP = IW | 1;  // odd row stride: the 4-wide row tasks of consecutive rows fall into different banks
  __shared__ __attribute__((aligned(8))) float s_in[IH][IWP][CN];
  __shared__ __attribute__((aligned(8))) float s_mid[IH][SB_TW + 1][CN];
  const int tid = threadIdx.y * blockDim.x + threadIdx.x;
  const TileId tile = xcd_tile();  // neighbouring tiles (shared halo) on the same XCD's L2
  const int tx0 = tile.x * SB_TW, ty0 = tile.y * SB_TH;
  const int vecEnd = ((w * CN) / 8) * 8;
  // EPI 1 / 2 / 3: the epilogue's operands (the two alphas, EPI 3 also I0's gradient) of this thread's column task are
  // requested NOW, in front of the tile load, instead of after the column pass: the kernel spent 80 % of its wave time
  // waiting (profiles/r03_v5_pmc_sq.txt) and this was the second of its two exposed memory round trips per tile.
  constexpr int kColTasks = SB_TW * (SB_TH / 4);
  static_assert(EPI == 0 || kColTasks <= NT, "one column task per thread when the epilogue is prefetched");
  float pa0[4] = {0.f, 0.f, 0.f, 0.f}, pa1[4] = {0.f, 0.f, 0.f, 0.f}, pm[4] = {0.f, 0.f, 0.f, 0.f};
  float2 pg[4];
  if (EPI != 0 && tid < kColTasks) {
    const int lx = tid % SB_TW, ly0 = (tid / SB_TW) * 4, gx = tx0 + lx;
    const size_t b0 = bs * idx.i0[tile.z], b1 = bs * idx.i1[tile.z];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const int gy = ty0 + ly0 + o;
      pg[o] = make_float2(0.f, 0.f);
      if (gx < w && gy < h) {
        const size_t off = (size_t)gy * w + gx;
        pa0[o] = A[b0 + off];
        pa1[o] = A[b1 + off];
        if (EPI == 3) pg[o] = Gp[b0 + off];
        if (EPI == 4) {
          pg[o] = Gp[bs * tile.z + off];                         // previous flow (this level, unscaled)
          pm[o] = static_cast<const float*>(recv)[b1 + off];     // motion of image i1 (PixFlow.h:186)
        }
      }
    }
  }
  src += (SRC == 2 ? up.sbs * CN : bs * (SRC == 1 ? 1 : CN)) * tile.z;
  // tile load: when the thread count is a multiple of the (padded) tile width a thread keeps its column and walks
  // down the rows — no division, one reflected column index per thread
  constexpr int LW = (IW + 7) & ~7;
  constexpr bool kColumnWalk = (NT % LW) == 0;
  const int lxc = tid % LW, lyc = tid / LW;
  const int gxc = reflect101(tx0 - R + (lxc < IW ? lxc : 0), w);
  // Two phases: every element this thread brings in is REQUESTED first (into registers), then all of them go to LDS. As
  // one loop (load, store, next) the compiler waited for each element before it asked for the next one: 5-6 serialised
  // memory round trips per tile.
  constexpr int kLdStep = kColumnWalk ? NT / LW : NT, kLdEnd = kColumnWalk ? IH : IH * IW;
  constexpr int kLdIters = (kLdEnd + kLdStep - 1) / kLdStep;
  float ld[kLdIters][CN];
#pragma unroll
  for (int it = 0; it < kLdIters; ++it) {
    // (no branch here: the index is clamped instead, so that nothing separates this element's loads from the next one's —
    // with an early-out per element the compiler waited for every element's loads before it issued the next ones; the
    // elements that do not exist are loaded from a valid address and never stored)
    const int i = min((kColumnWalk ? lyc : tid) + it * kLdStep, kLdEnd - 1);
    const int ly = kColumnWalk ? i : i / IW, lx = kColumnWalk ? min(lxc, IW - 1) : i - ly * IW;
    const int gy = reflect101(ty0 - R + ly, h), gx = kColumnWalk ? gxc : reflect101(tx0 - R + lx, w);
    if (SRC == 2) {  // k_resize_linear_f32's arithmetic at (gx, gy), then the scalar multiply
      int sx, sy;
      float fx, fy;
      resize_coord(gx, up.scx, &sx, &fx);
      if (sx < 0) { fx = 0; sx = 0; }
      if (sx >= up.sw - 1) { fx = 0; sx = up.sw - 1; }
      resize_coord(gy, up.scy, &sy, &fy);
      const float* S0 = src + (size_t)clip_idx(sy, up.sh) * up.sw * CN;
      const float* S1 = src + (size_t)clip_idx(sy + 1, up.sh) * up.sw * CN;
      const float b0 = 1.f - fy, b1 = fy;
#pragma unroll
      for (int k = 0; k < CN; ++k) {
        float h0, h1;
        if (sx >= up.sw - 1) {
          h0 = S0[sx * CN + k] * 1.0f;
          h1 = S1[sx * CN + k] * 1.0f;
        } else {
          const float a0 = 1.f - fx, a1 = fx;
          h0 = S0[sx * CN + k] * a0 + S0[(sx + 1) * CN + k] * a1;
          h1 = S1[sx * CN + k] * a0 + S1[(sx + 1) * CN + k] * a1;
        }
        float v = h0 * b0 + h1 * b1;
        v *= up.post_scale;
        ld[it][k] = v;
      }
    } else if (SRC == 1) {  // Sobel ksize=1: [-1 0 +1], BORDER_REPLICATE, no scale
      const float* r0 = src + (size_t)gy * w;
      ld[it][0] = r0[min(gx + 1, w - 1)] - r0[max(gx - 1, 0)];
      ld[it][CN - 1] = src[(size_t)min(gy + 1, h - 1) * w + gx] - src[(size_t)max(gy - 1, 0) * w + gx];
    } else if (CN == 2) {
      const float2 v = *reinterpret_cast<const float2*>(src + ((size_t)gy * w + gx) * CN);
      ld[it][0] = v.x;
      ld[it][CN - 1] = v.y;
    } else {
      ld[it][0] = src[(size_t)gy * w + gx];
    }
  }
#pragma unroll
  for (int it = 0; it < kLdIters; ++it) {
    const int i = (kColumnWalk ? lyc : tid) + it * kLdStep;
    if (i >= kLdEnd || (kColumnWalk && lxc >= IW)) continue;
    const int ly = kColumnWalk ? i : i / IW, lx = kColumnWalk ? lxc : i - ly * IW;
    if (CN == 2) *reinterpret_cast<float2*>(&s_in[ly][lx][0]) = make_float2(ld[it][0], ld[it][CN - 1]);
    else s_in[ly][lx][0] = ld[it][0];
  }
  __syncthreads();
  // row pass: task = (row ly, group of 4 consecutive x). Two-channel images move through LDS as 8-byte pairs (one
  // ds_read_b64 per tap for both channels: half the LDS instructions and all banks in use).
  for (int t = tid; t < IH * (SB_TW / 4); t += NT) {
    const int ly = t % IH, lx0 = (t / IH) * 4;
    float v[CN][4 + 2 * R];
#pragma unroll
    for (int j = 0; j < 4 + 2 * R; ++j) {
      if (CN == 2) {
        const float2 p = *reinterpret_cast<const float2*>(&s_in[ly][lx0 + j][0]);
        v[0][j] = p.x;
        v[CN - 1][j] = p.y;
      } else {
        v[0][j] = s_in[ly][lx0 + j][0];
      }
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      float acc[CN];
#pragma unroll
      for (int k = 0; k < CN; ++k) {
        if (R <= 2) {  // SymmRowSmallFilter: centre, then symmetric pairs
          acc[k] = taps.k[0] * v[k][o + R];
#pragma unroll
          for (int j = 1; j <= R; ++j) acc[k] += taps.k[j] * (v[k][o + R + j] + v[k][o + R - j]);
        } else {  // generic RowFilter: left to right; its SSE2 loop (whole groups of 8 row elements) starts from +0
          acc[k] = ((tx0 + lx0 + o) * CN + k < vecEnd) ? 0.0f : -0.0f;
#pragma unroll
          for (int j = 0; j <= 2 * R; ++j) acc[k] += taps.k[j < R ? R - j : j - R] * v[k][o + j];
        }
      }
      if (CN == 2) *reinterpret_cast<float2*>(&s_mid[ly][lx0 + o][0]) = make_float2(acc[0], acc[CN - 1]);
      else s_mid[ly][lx0 + o][0] = acc[0];
    }
  }
  __syncthreads();
  // column pass: task = (column lx, group of 4 consecutive y)
  dst = dst_tab ? dst_tab[tile.z] : dst + bs * CN * tile.z;
  for (int t = tid; t < SB_TW * (SB_TH / 4); t += NT) {
    const int lx = t % SB_TW, ly0 = (t / SB_TW) * 4;
    const int gx = tx0 + lx;
    float outv[4][CN];
    {
      float v[CN][4 + 2 * R];
#pragma unroll
      for (int j = 0; j < 4 + 2 * R; ++j) {
        if (CN == 2) {
          const float2 p = *reinterpret_cast<const float2*>(&s_mid[ly0 + j][lx][0]);
          v[0][j] = p.x;
          v[CN - 1][j] = p.y;
        } else {
          v[0][j] = s_mid[ly0 + j][lx][0];
        }
      }
      if (CN == 2) {
        // both channels of a pixel as one two-float vector: v_pk_add_f32 / v_pk_mul_f32, each half rounded like the scalar
        // operation (no fused multiply-add). The compiler found this form by itself only where both channels meet the same
        // epilogue; EPI 2 (channel 0 selected against the NaN mark) ran the column pass as 186 scalar operations.
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 p[4 + 2 * R];
#pragma unroll
        for (int j = 0; j < 4 + 2 * R; ++j) p[j] = f32x2{v[0][j], v[CN - 1][j]};
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          f32x2 acc = taps.k[0] * p[o + R] + 0.0f;  // SymmColumnFilter: centre + delta (= +0), then symmetric pairs
#pragma unroll
          for (int j = 1; j <= R; ++j) acc += taps.k[j] * (p[o + R + j] + p[o + R - j]);
          outv[o][0] = acc.x;
          outv[o][CN - 1] = acc.y;
        }
      } else {
#pragma unroll
        for (int k = 0; k < CN; ++k)
#pragma unroll
          for (int o = 0; o < 4; ++o) {
            float acc = taps.k[0] * v[k][o + R] + 0.0f;  // SymmColumnFilter: centre + delta (= +0), then symmetric pairs
#pragma unroll
            for (int j = 1; j <= R; ++j) acc += taps.k[j] * (v[k][o + R + j] + v[k][o + R - j]);
            outv[o][k] = acc;
          }
      }
    }
    if (gx >= w) continue;
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const int gy = ty0 + ly0 + o;
      if (gy >= h) continue;
      const size_t off = (size_t)gy * w + gx;
      if (EPI == 1 || EPI == 4) {
        const float cc = 1.0f - pa0[o] * pa1[o];
#pragma unroll
        for (int k = 0; k < CN; ++k) outv[o][k] = cc * outv[o][k] + (1.0f - cc) * s_in[ly0 + o + R][lx + R][k];
      }
      if (EPI == 4) {
        const float wgt = 1.0f - pm[o];
        const float px = pg[o].x * up.post_scale, py = pg[o].y * up.post_scale;
        outv[o][0] = outv[o][0] * (1.0f - wgt) + px * wgt;
        outv[o][CN - 1] = outv[o][CN - 1] * (1.0f - wgt) + py * wgt;
      }
      if (EPI == 2 || EPI == 3) {
        const bool upd = pa0[o] > 0.9f && pa1[o] > 0.9f;
        if (EPI == 2)
          static_cast<float2*>(recv)[bs * tile.z + off] = make_float2(upd ? outv[o][0] : __int_as_float(0x7fc00000), outv[o][CN - 1]);
        else
          static_cast<float4*>(recv)[bs * tile.z + off] = make_float4(upd ? pg[o].x : __int_as_float(0x7fc00000), pg[o].y, outv[o][0], outv[o][CN - 1]);
        // rows with at least one updated pixel (all-ones = none: the sweeps let bands without any leave at once)
        if (rowflags && upd) rowflags[(size_t)tile.z * h + gy] = 0u;
      } else {
#pragma unroll
        for (int k = 0; k < CN; ++k) dst[off * CN + k] = outv[o][k];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// resize INTER_LINEAR float (pyramid x0.9, PixFlow.h:487; final upscale :176).
// Horizontal: s<0 => (0, f=0); s>=sw-1 => S[sw-1]*1; vertical rows clipped, f kept.
template <int CN, int PPT>
__global__ __launch_bounds__(256) void k_resize_linear_f32(const float* __restrict__ src, int sw, int sh, size_t sbs,
                                                           float* __restrict__ dst, int dw, int dh, size_t dbs,
                                                           double scx, double scy, float post_scale, int do_scale) {
  const int dx = blockIdx.x * blockDim.x + threadIdx.x;
  const int dy = blockIdx.y * blockDim.y + threadIdx.y;
  if (dx >= dw || dy >= dh) return;
  int sx, sy;
  float fx, fy;
  resize_coord(dx, scx, &sx, &fx);
  if (sx < 0) { fx = 0; sx = 0; }
  if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
  resize_coord(dy, scy, &sy, &fy);
  const size_t r0 = (size_t)clip_idx(sy, sh) * sw * CN, r1 = (size_t)clip_idx(sy + 1, sh) * sw * CN;
  const float b0 = 1.f - fy, b1 = fy;
  const bool edge = sx >= sw - 1;
  const int sx1 = edge ? sx : sx + 1;  // (the edge case multiplies the one tap by 1.0f: the second address is a dummy)
  const float a0 = 1.f - fx, a1 = fx;
  // the planes of a launch share their geometry: one thread produces this pixel of PPT of them. All 4 x PPT taps are
  // requested before the first is used — the kernel is pure memory latency (92 % of its wave time waiting when the
  // planes were walked one after the other, profiles/r03_v5_pmc_sq.txt).
  float t[PPT][CN][4];
#pragma unroll
  for (int pl = 0; pl < PPT; ++pl) {
    const size_t z = (size_t)blockIdx.z * PPT + pl;
    const float* S0 = src + sbs * CN * z + r0;
    const float* S1 = src + sbs * CN * z + r1;
#pragma unroll
    for (int k = 0; k < CN; ++k) {
      t[pl][k][0] = S0[sx * CN + k];
      t[pl][k][1] = S0[sx1 * CN + k];
      t[pl][k][2] = S1[sx * CN + k];
      t[pl][k][3] = S1[sx1 * CN + k];
    }
  }
#pragma unroll
  for (int pl = 0; pl < PPT; ++pl) {
    const size_t z = (size_t)blockIdx.z * PPT + pl;
    float* D = dst + dbs * CN * z;
#pragma unroll
    for (int k = 0; k < CN; ++k) {
      float h0, h1;
      if (edge) {
        h0 = t[pl][k][0] * 1.0f;
        h1 = t[pl][k][2] * 1.0f;
      } else {
        h0 = t[pl][k][0] * a0 + t[pl][k][1] * a1;
        h1 = t[pl][k][2] * a0 + t[pl][k][3] * a1;
      }
      float v = h0 * b0 + h1 * b1;
      if (do_scale) v *= post_scale;
      D[((size_t)dy * dw + dx) * CN + k] = v;
    }
  }
}

// The same resize for the image pyramids (one channel, x0.9 per level), tiled: with one thread per output pixel the kernel
// is bound by its memory INSTRUCTIONS — four scattered 4-byte taps per output and plane; 68 % of the wave time ready
// but not issued behind the address unit (profiles/r03_v8_pmc_sq.txt). Here a 64x16 tile's source box (<= 76x20 for
// scales up to 1.13) is read once as 16-byte pieces into LDS for PPT planes, the FP64 coordinates are computed once per
// tile column / row (80 per tile instead of 2 per pixel), and the taps are LDS reads. Arithmetic as in
// k_resize_linear_f32, term by term.
constexpr int RL_TW = 64, RL_TH = 16, RL_BW = 76, RL_BH = 20;
template <int PPT>
__global__ __launch_bounds__(256) void k_resize_linear_f32c1_tiled(const float* __restrict__ src, int sw, int sh, size_t sbs,
                                                                   float* __restrict__ dst, int dw, int dh, size_t dbs,
                                                                   double scx, double scy, float post_scale, int do_scale) {
  __shared__ __attribute__((aligned(16))) float s_box[PPT][RL_BH][RL_BW];
  __shared__ int s_sx[RL_TW], s_r0[RL_TH], s_r1[RL_TH];
  __shared__ float s_fx[RL_TW], s_fy[RL_TH];
  const TileId tile = xcd_tile();
  const int tid = threadIdx.x, tx0 = tile.x * RL_TW, ty0 = tile.y * RL_TH;
  if (tid < RL_TW) {  // column coordinates (absolute; columns beyond the image repeat the last one)
    int sx;
    float fx;
    resize_coord(min(tx0 + tid, dw - 1), scx, &sx, &fx);
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    s_sx[tid] = sx;
    s_fx[tid] = fx;
  } else if (tid < RL_TW + RL_TH) {
    int sy;
    float fy;
    resize_coord(min(ty0 + tid - RL_TW, dh - 1), scy, &sy, &fy);
    s_r0[tid - RL_TW] = clip_idx(sy, sh);
    s_r1[tid - RL_TW] = clip_idx(sy + 1, sh);
    s_fy[tid - RL_TW] = fy;
  }
  __syncthreads();
  // the box: columns bx0 .. (last column's sx) + 1, rows by0 .. (last row's r1); the launcher guarantees that it fits
  const int bx0 = s_sx[0], by0 = s_r0[0];
  const int bw4 = (min(s_sx[RL_TW - 1] + 1, sw - 1) - bx0 + 4) >> 2, bh = s_r1[RL_TH - 1] - by0 + 1;
  const int npieces = bh * bw4;
  constexpr int kIters = (RL_BH * (RL_BW / 4) + 255) / 256;
  float4 ld[kIters][PPT];
  // all pieces requested before the first is stored (index clamped, no early-out: see k_sepblur)
#pragma unroll
  for (int it = 0; it < kIters; ++it) {
    const int i = min(tid + it * 256, npieces - 1);
    const int row = i / bw4, gx = bx0 + 4 * (i - row * bw4);
    const size_t off = (size_t)(by0 + row) * sw;
#pragma unroll
    for (int pl = 0; pl < PPT; ++pl) {
      const float* S = src + sbs * ((size_t)tile.z * PPT + pl) + off;
      if (gx + 3 < sw) {
        typedef float f4a4 __attribute__((ext_vector_type(4), aligned(4)));  // (rows start at any 4-byte address)
        const f4a4 q = *reinterpret_cast<const f4a4*>(S + gx);
        ld[it][pl] = make_float4(q.x, q.y, q.z, q.w);
      } else {  // the piece crosses the end of the row: the columns behind it are never tapped
        ld[it][pl] = make_float4(S[min(gx, sw - 1)], S[min(gx + 1, sw - 1)], S[min(gx + 2, sw - 1)], S[sw - 1]);
      }
    }
  }
#pragma unroll
  for (int it = 0; it < kIters; ++it) {
    const int i = tid + it * 256;
    if (i >= npieces) continue;
    const int row = i / bw4, c4 = i - row * bw4;
#pragma unroll
    for (int pl = 0; pl < PPT; ++pl) *reinterpret_cast<float4*>(&s_box[pl][row][4 * c4]) = ld[it][pl];
  }
  __syncthreads();
  const int cx = tid & (RL_TW - 1), dx = tx0 + cx;
  if (dx >= dw) return;
  const int sxa = s_sx[cx], sxr = sxa - bx0;
  const bool edge = sxa >= sw - 1;  // (then the second tap is a dummy: the one tap is multiplied by 1.0f)
  const int sx1 = edge ? sxr : sxr + 1;
  const float fx = s_fx[cx], a0 = 1.f - fx, a1 = fx;
#pragma unroll
  for (int j = 0; j < RL_TH / 4; ++j) {
    const int ry = (tid >> 6) + 4 * j, dy = ty0 + ry;
    if (dy >= dh) continue;
    const int r0 = s_r0[ry] - by0, r1 = s_r1[ry] - by0;
    const float fy = s_fy[ry], b0 = 1.f - fy, b1 = fy;
#pragma unroll
    for (int pl = 0; pl < PPT; ++pl) {
      const float t0 = s_box[pl][r0][sxr], t1 = s_box[pl][r0][sx1], t2 = s_box[pl][r1][sxr], t3 = s_box[pl][r1][sx1];
      float h0, h1;
      if (edge) {
        h0 = t0 * 1.0f;
        h1 = t2 * 1.0f;
      } else {
        h0 = t0 * a0 + t1 * a1;
        h1 = t2 * a0 + t3 * a1;
      }
      float v = h0 * b0 + h1 * b1;
      if (do_scale) v *= post_scale;
      dst[dbs * ((size_t)tile.z * PPT + pl) + (size_t)dy * dw + dx] = v;
    }
  }
}

// resize INTER_CUBIC float2 (flow upscale between levels, PixFlow.h:170-171; prevFlow :103-104),
// followed by the scalar multiply.
__global__ __launch_bounds__(256) void k_resize_cubic_f32c2(const float2* __restrict__ src, int sw, int sh,
                                                            size_t sbs, float2* __restrict__ dst, int dw, int dh,
                                                            size_t dbs, double scx, double scy, float post_scale,
                                                            const float2* const* __restrict__ src_tab) {
  const int dx = blockIdx.x * blockDim.x + threadIdx.x;
  const int dy = blockIdx.y * blockDim.y + threadIdx.y;
  if (dx >= dw || dy >= dh) return;
  src = src_tab ? src_tab[blockIdx.z] : src + sbs * blockIdx.z;
  dst += dbs * blockIdx.z;
  int sx, sy;
  float fx, fy, ax[4], ay[4];
  resize_coord(dx, scx, &sx, &fx);
  cubic_coeffs(fx, ax);
  resize_coord(dy, scy, &sy, &fy);
  cubic_coeffs(fy, ay);
  const int x0 = clip_idx(sx - 1, sw), x1 = clip_idx(sx, sw), x2 = clip_idx(sx + 1, sw), x3 = clip_idx(sx + 2, sw);
  float hx[4], hy[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float2* S = src + (size_t)clip_idx(sy - 1 + r, sh) * sw;
    const float2 p0 = S[x0], p1 = S[x1], p2 = S[x2], p3 = S[x3];
    hx[r] = p0.x * ax[0] + p1.x * ax[1] + p2.x * ax[2] + p3.x * ax[3];
    hy[r] = p0.y * ax[0] + p1.y * ax[1] + p2.y * ax[2] + p3.y * ax[3];
  }
  float2 o;
  o.x = hx[0] * ay[0] + hx[1] * ay[1] + hx[2] * ay[2] + hx[3] * ay[3];
  o.y = hy[0] * ay[0] + hy[1] * ay[1] + hy[2] * ay[2] + hy[3] * ay[3];
  o.x *= post_scale;
  o.y *= post_scale;
  dst[(size_t)dy * dw + dx] = o;
}

// The same resize for ratios <= 1 (the x1/0.9 upscale between pyramid levels, 190 M pixel-levels per frame) with the
// source window of a 64x16 output tile staged in LDS: the per-column / per-row coordinates and cubic weights (double
// precision coordinate arithmetic) are computed once per tile instead of once per pixel, the horizontal pass runs once
// per source row of the window instead of four times per output, and global memory is read in coalesced rows.
constexpr int UC_TW = 64, UC_TH = 16, UC_SW = 72, UC_SH = 24;
__global__ __launch_bounds__(256) void k_resize_cubic_f32c2_tiled(const float2* __restrict__ src, int sw, int sh,
                                                                  size_t sbs, float2* __restrict__ dst, int dw, int dh,
                                                                  size_t dbs, double scx, double scy, float post_scale) {
  __shared__ float2 s_src[UC_SH][UC_SW];
  __shared__ float2 s_h[UC_SH][UC_TW];
  __shared__ float s_ax[UC_TW][4], s_ay[UC_TH][4];
  __shared__ int s_sx[UC_TW], s_sy[UC_TH];
  const int tid = threadIdx.x;
  const TileId tile = xcd_tile();
  const int dx0 = tile.x * UC_TW, dy0 = tile.y * UC_TH;
  src += sbs * tile.z;
  dst += dbs * tile.z;
  if (tid < UC_TW) {
    float f;
    resize_coord(min(dx0 + tid, dw - 1), scx, &s_sx[tid], &f);
    cubic_coeffs(f, s_ax[tid]);
  } else if (tid < UC_TW + UC_TH) {
    float f;
    resize_coord(min(dy0 + tid - UC_TW, dh - 1), scy, &s_sy[tid - UC_TW], &f);
    cubic_coeffs(f, s_ay[tid - UC_TW]);
  }
  __syncthreads();
  const int X0 = clip_idx(s_sx[0] - 1, sw), X1 = clip_idx(s_sx[UC_TW - 1] + 2, sw);
  const int Y0 = clip_idx(s_sy[0] - 1, sh), Y1 = clip_idx(s_sy[UC_TH - 1] + 2, sh);
  const int W = X1 - X0 + 1, H = Y1 - Y0 + 1;  // <= UC_SW x UC_SH for ratios <= 1 (checked by the launcher)
  // A thread keeps its column c = tid & 63 through all three phases (256 is a multiple of the tile width): no
  // division in the load, and its source offsets and horizontal coefficients are read once, not once per row.
  const int c = tid & (UC_TW - 1), q = tid >> 6;
  {  // (all rows of the thread's column requested before the first is stored: not one round trip per row)
    constexpr int kIt = (UC_SH + 3) / 4;
    float2 v[kIt];
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
      const int ly = q + 4 * it;
      v[it] = (ly < H && c < W) ? src[(size_t)(Y0 + ly) * sw + X0 + c] : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
      const int ly = q + 4 * it;
      if (ly < H && c < W) s_src[ly][c] = v[it];
    }
  }
  if (W > UC_TW) {  // the up to 8 remaining columns of the window
    const int lx = UC_TW + (tid & 7);
    for (int ly = tid >> 3; ly < H; ly += 32)
      if (lx < W) s_src[ly][lx] = src[(size_t)(Y0 + ly) * sw + X0 + lx];
  }
  __syncthreads();
  {  // horizontal pass of every source row of the window
    const int sx = s_sx[c];
    const int o0 = clip_idx(sx - 1, sw) - X0, o1 = clip_idx(sx, sw) - X0, o2 = clip_idx(sx + 1, sw) - X0,
              o3 = clip_idx(sx + 2, sw) - X0;
    const float a0 = s_ax[c][0], a1 = s_ax[c][1], a2 = s_ax[c][2], a3 = s_ax[c][3];
    for (int ly = q; ly < H; ly += 4) {
      const float2 p0 = s_src[ly][o0], p1 = s_src[ly][o1], p2 = s_src[ly][o2], p3 = s_src[ly][o3];
      float2 hv;
      hv.x = p0.x * a0 + p1.x * a1 + p2.x * a2 + p3.x * a3;
      hv.y = p0.y * a0 + p1.y * a1 + p2.y * a2 + p3.y * a3;
      s_h[ly][c] = hv;
    }
  }
  __syncthreads();
  const int r0 = q * 4;
  const int dx = dx0 + c;
  if (dx >= dw) return;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = r0 + j, dy = dy0 + r;
    if (dy >= dh) break;
    const int sy = s_sy[r];
    const float2 h0 = s_h[clip_idx(sy - 1, sh) - Y0][c], h1 = s_h[clip_idx(sy, sh) - Y0][c],
                 h2 = s_h[clip_idx(sy + 1, sh) - Y0][c], h3 = s_h[clip_idx(sy + 2, sh) - Y0][c];
    const float b0 = s_ay[r][0], b1 = s_ay[r][1], b2 = s_ay[r][2], b3 = s_ay[r][3];
    float2 o;
    o.x = h0.x * b0 + h1.x * b1 + h2.x * b2 + h3.x * b3;
    o.y = h0.y * b0 + h1.y * b1 + h2.y * b2 + h3.y * b3;
    o.x *= post_scale;
    o.y *= post_scale;
    dst[(size_t)dy * dw + dx] = o;
  }
}




// ==========================================================================================
// launchers
// ---- pixflow_search_20 only: adjustInitialFlow at the coarsest level (PixFlow.h:219-342) ----
__global__ void k_search_init(const float* __restrict__ I, const float* __restrict__ A, int w, int h, size_t pbs,
                              FlowIdx idx, float2* __restrict__ flow, int hint, int dist, float* __restrict__ I1eq) {
  // one workgroup per flow; thread 0 computes the global ratio sequentially, then all threads search.
  const int b = blockIdx.x;
  const float* __restrict__ I0 = I + pbs * idx.i0[b];
  const float* __restrict__ I1 = I + pbs * idx.i1[b];
  const float* __restrict__ A0 = A + pbs * idx.i0[b];
  const float* __restrict__ A1 = A + pbs * idx.i1[b];
  flow += pbs * b;
  I1eq += pbs * b;
  __shared__ float s_ratio;
  if (threadIdx.x == 0) {
    float sumL = 0, sumR = 0;
    for (int i = 0; i < w * h; ++i) {
      const float a = A0[i] * A1[i];
      sumL += a * I0[i];
      sumR += a * I1[i];
    }
    s_ratio = sumL / sumR;
  }
  __syncthreads();
  const float ratio = s_ratio;
  for (int i = threadIdx.x; i < w * h; i += blockDim.x) I1eq[i] = I1[i] * ratio;
  __syncthreads();
  const int kRatio = 8;
  const int ortho = (dist + kRatio / 2) / kRatio;
  const int thickness = 2 * ortho + 1;
  int bx, by, bw, bh;
  if (hint == 1) { bx = 0; by = -ortho; bw = dist + 1; bh = thickness; }
  else if (hint == 2) { bx = -ortho; by = 0; bw = thickness; bh = dist + 1; }
  else if (hint == 3) { bx = -dist; by = -ortho; bw = dist + 1; bh = thickness; }
  else { bx = -ortho; by = -dist; bw = thickness; bh = dist + 1; }
  auto patchErr = [&](int i0x, int i0y, int i1x, int i1y) -> float {
    float sad = 0, alpha = 0;
    for (int dy = -2; dy <= 2; ++dy) {
      const int d0y = i0y + dy;
      if (0 <= d0y && d0y < h) {
        const int d1y = min(max(i1y + dy, 0), h - 1);
        for (int dx = -2; dx <= 2; ++dx) {
          const int d0x = i0x + dx;
          if (0 <= d0x && d0x < w) {
            const int d1x = min(max(i1x + dx, 0), w - 1);
            const float difference = I0[d0y * w + d0x] - I1eq[d1y * w + d1x];
            sad += fabsf(difference);
            alpha += A0[d0y * w + d0x] * A1[d1y * w + d1x];
          }
        }
      }
    }
    sad /= alpha;
    const float ddx = (float)(i1x - i0x), ddy = (float)(i1y - i0y);
    const float length = (float)sqrt((double)ddx * ddx + (double)ddy * ddy);
    sad *= 1 + length / dist;
    return sad;
  };
  for (int i = threadIdx.x; i < w * h; i += blockDim.x) {
    const int i0y = i / w, i0x = i - i0y * w;
    if (A0[i] > 0.9f) {
      float errorBest = 0.8f * patchErr(i0x, i0y, i0x, i0y);
      int bxb = i0x, byb = i0y;
      for (int dy = by; dy < by + bh; ++dy)
        for (int dx = bx; dx < bx + bw; ++dx) {
          const int i1x = i0x + dx, i1y = i0y + dy;
          if (0 <= i1x && i1x < w && 0 <= i1y && i1y < h) {
            const float e = patchErr(i0x, i0y, i1x, i1y);
            if (errorBest > e) { errorBest = e; bxb = i1x; byb = i1y; }
          }
        }
      flow[i] = make_float2((float)(bxb - i0x), (float)(byb - i0y));
    }
  }
}

static inline dim3 grid2d(int w, int h, int B, dim3 blk) { return dim3((w + blk.x - 1) / blk.x, (h + blk.y - 1) / blk.y, B); }

void launch_resize_cubic_u8c4(hipStream_t st, const uchar4* src, int sw, int sh, size_t sbs, uchar4* dst, int dw,
                              int dh, size_t dbs, int B, const uchar4* const* src_tab) {
  const double scx = 1.0 / ((double)dw / (double)sw), scy = 1.0 / ((double)dh / (double)sh);
  dim3 blk(32, 8);
  hipLaunchKernelGGL(k_resize_cubic_u8c4, grid2d(dw, dh, B, blk), blk, 0, st, src, sw, sh, sbs, dst, dw, dh, dbs, scx,
                     scy, src_tab);
}
void launch_gray_alpha(hipStream_t st, const uchar4* src, size_t n, size_t sbs, float* gray, float* alpha, size_t pbs,
                       int B) {
  hipLaunchKernelGGL(k_gray_alpha, dim3((unsigned)((n + 255) / 256), 1, B), dim3(256), 0, st, src, n, sbs, gray, alpha,
                     pbs);
}
void launch_motion(hipStream_t st, const uchar4* cur, const uchar4* prev, size_t n, size_t sbs, float* motion,
                   size_t pbs, int B) {
  hipLaunchKernelGGL(k_motion, dim3((unsigned)((n + 255) / 256), 1, B), dim3(256), 0, st, cur, prev, n, sbs, motion,
                     pbs);
}
template <int R, int CN, int EPI, int SRC>
static void launch_sepblur_t(hipStream_t st, const float* src, float* dst, int w, int h, size_t bs, int B,
                             const BlurTaps& t, const float* A, const FlowIdx& idx, const float2* Gp, void* rec,
                             float* const* dst_tab = nullptr, const UpSrc& up = UpSrc{}, unsigned* rowflags = nullptr) {
  dim3 blk(64, 4);
  if constexpr (R == 7) {  // 15x15: 32x32 tile — 29 KB of LDS, 1.44x row-pass halo work (64x16: 35 KB, 1.9x; measured 30 % slower);
    // 384 threads: the row pass has 46 rows x 8 groups = 368 tasks (two rounds on 256 threads, the second 44 % full)
    dim3 grd((w + 31) / 32, (h + 31) / 32, B);
    hipLaunchKernelGGL((k_sepblur<R, CN, EPI, SRC, 32, 32, 384>), grd, dim3(64, 6), 0, st, src, dst, w, h, bs, t, A, idx, Gp, rec, dst_tab, up, rowflags);
  } else {
    dim3 grd((w + 63) / 64, (h + 15) / 16, B);
    hipLaunchKernelGGL((k_sepblur<R, CN, EPI, SRC>), grd, blk, 0, st, src, dst, w, h, bs, t, A, idx, Gp, rec, dst_tab, up, rowflags);
  }
}
void launch_sepblur(hipStream_t st, const float* src, float* dst, int w, int h, int cn, size_t bs, int B,
                    const BlurTaps& t, float* const* dst_tab) {
  static const FlowIdx none = {nullptr, nullptr};
  if (t.r == 1 && cn == 1) launch_sepblur_t<1, 1, 0, 0>(st, src, dst, w, h, bs, B, t, nullptr, none, nullptr, nullptr);
  else if (t.r == 1 && cn == 2) launch_sepblur_t<1, 2, 0, 0>(st, src, dst, w, h, bs, B, t, nullptr, none, nullptr, nullptr, dst_tab);
  else if (t.r == 2 && cn == 1) launch_sepblur_t<2, 1, 0, 0>(st, src, dst, w, h, bs, B, t, nullptr, none, nullptr, nullptr);
  else if (t.r == 7 && cn == 2) launch_sepblur_t<7, 2, 0, 0>(st, src, dst, w, h, bs, B, t, nullptr, none, nullptr, nullptr);
  else throw std::runtime_error("launch_sepblur: unsupported radius/channels");
}
void launch_diffusion(hipStream_t st, const float2* flow, float2* dst, int w, int h, size_t bs, int B,
                      const BlurTaps& t, const float* A, const FlowIdx& idx) {
  launch_sepblur_t<7, 2, 1, 0>(st, (const float*)flow, (float*)dst, w, h, bs, B, t, A, idx, nullptr, nullptr);
}
// ... followed, on the same pixel, by adjustFlowTowardPrevious with the previous flow's level rescaled as it is read
void launch_diffusion_adjust(hipStream_t st, const float2* flow, float2* dst, int w, int h, size_t bs, int B, const BlurTaps& t,
                             const float* A, const FlowIdx& idx, const float2* prev, const float* motion, float prev_scale) {
  UpSrc up{};
  up.post_scale = prev_scale;
  launch_sepblur_t<7, 2, 4, 0>(st, (const float*)flow, (float*)dst, w, h, bs, B, t, A, idx, prev, const_cast<float*>(motion), nullptr, up);
}
// resize(flow, originalSize, INTER_LINEAR); flow *= s; GaussianBlur(flow, 3x3) in one pass (PixFlow.h:175-182)
void launch_upscale_blur(hipStream_t st, const float2* src, int sw, int sh, size_t sbs, float2* dst, int dw, int dh,
                         size_t dbs, int B, float post_scale, const BlurTaps& t, float* const* dst_tab) {
  static const FlowIdx none = {nullptr, nullptr};
  if (t.r != 1) throw std::runtime_error("launch_upscale_blur: 3x3 kernel expected");
  UpSrc up;
  up.sw = sw; up.sh = sh; up.sbs = sbs;
  up.scx = 1.0 / ((double)dw / (double)sw);
  up.scy = 1.0 / ((double)dh / (double)sh);
  up.post_scale = post_scale;
  launch_sepblur_t<1, 2, 0, 2>(st, (const float*)src, (float*)dst, dw, dh, dbs, B, t, nullptr, none, nullptr, nullptr, dst_tab, up);
}
// Sobel + 3x3 Gaussian of a float plane in one pass -> packed (Ix, Iy) (PixFlow.h:353-366)
void launch_gradients(hipStream_t st, const float* I, float2* G, int w, int h, size_t bs, int B, const BlurTaps& t) {
  static const FlowIdx none = {nullptr, nullptr};
  if (t.r != 1) throw std::runtime_error("launch_gradients: 3x3 kernel expected");
  launch_sepblur_t<1, 2, 0, 1>(st, I, (float*)G, w, h, bs, B, t, nullptr, none, nullptr, nullptr);
}
// Measured and NOT adopted (round 6, commit ecbd4d5's k_upblur_rec; profiles/r06_v3_upblur_fusion_ab.txt): the inter-level upscale
// (launch_resize_cubic_f32c2) fused into this blur — one kernel reading the coarser level, writing the upscaled flow's 32x32 tile and
// the records, the resize's source window and horizontal pass staged in LDS that the blur's tile then reuses (36 KB). Bit-exact (the
// emulated flow tests and the GPU flow / frame / operator tests passed with it), 8 bytes per pixel-level and one launch per level
// less — and slower: upscale + blur 3.07 ms per frame against 1.90 as two kernels (22-slot batch alone), 46.8 against 49.0 frames/s
// in the headline. The blur needs the upscaled flow on 46x46 positions per 32x32 outputs, so the fused kernel evaluates the bicubic
// resize 2.07 times per pixel where the resize kernel's 64x16 tiles do it once, with two more LDS passes and four more barriers per
// tile; neither kernel is near the HBM roof (0.2 / 0.36 of it), so the bytes saved buy nothing.
// 15x15 Gaussian of the flow written straight into what the sweeps read (blurredFlow is only read by them): half-records
// {blurred.x | NaN = not updated, blurred.y} for the throughput kernel — the other half of a pixel's record is I0's gradient in
// the gradient planes — or, with G given, full records {I0x | NaN, I0y, blurred.x, blurred.y} for the latency kernel
void launch_blur_to_records(hipStream_t st, const float2* flow, void* rec, int w, int h, size_t bs, int B,
                            const BlurTaps& t, const float2* G, const float* A, const FlowIdx& idx, unsigned* rowflags) {
  if (t.r != 7) throw std::runtime_error("launch_blur_to_records: 15x15 kernel expected");
  if (G) launch_sepblur_t<7, 2, 3, 0>(st, (const float*)flow, nullptr, w, h, bs, B, t, A, idx, G, rec, nullptr, UpSrc{}, rowflags);
  else launch_sepblur_t<7, 2, 2, 0>(st, (const float*)flow, nullptr, w, h, bs, B, t, A, idx, nullptr, rec, nullptr, UpSrc{}, rowflags);
}
void launch_resize_linear_f32(hipStream_t st, const float* src, int sw, int sh, size_t sbs, float* dst, int dw, int dh,
                              size_t dbs, int cn, int B, float post_scale, int do_scale) {
  const double scx = 1.0 / ((double)dw / (double)sw), scy = 1.0 / ((double)dh / (double)sh);
  const int ppt = (B % 4 == 0) ? 4 : (B % 2 == 0) ? 2 : 1;  // planes per thread (the coordinates are computed once)
  // one-channel planes (the image pyramids): tiled, if a tile's source box fits — x0.9 levels do, the smallest levels'
  // rounded sizes may not
  const int tw = std::min(dw, RL_TW), th = std::min(dh, RL_TH);
  const bool fits = std::min(sw, (int)std::floor((tw - 1) * scx) + 3) + 3 <= RL_BW &&
                    std::min(sh, (int)std::floor((th - 1) * scy) + 3) <= RL_BH;
  if (cn == 1 && fits) {
    const dim3 grid((dw + RL_TW - 1) / RL_TW, (dh + RL_TH - 1) / RL_TH, B / ppt);
#define S360_RLT(P)                                                                                                   \
  hipLaunchKernelGGL((k_resize_linear_f32c1_tiled<P>), grid, dim3(256), 0, st, src, sw, sh, sbs, dst, dw, dh, dbs, scx, \
                     scy, post_scale, do_scale)
    if (ppt == 4) S360_RLT(4); else if (ppt == 2) S360_RLT(2); else S360_RLT(1);
#undef S360_RLT
    return;
  }
  dim3 blk(32, 8);
#define S360_RL(C, P)                                                                                                   \
  hipLaunchKernelGGL((k_resize_linear_f32<C, P>), grid2d(dw, dh, B / P, blk), blk, 0, st, src, sw, sh, sbs, dst, dw, dh, \
                     dbs, scx, scy, post_scale, do_scale)
  if (cn == 1) { if (ppt == 4) S360_RL(1, 4); else if (ppt == 2) S360_RL(1, 2); else S360_RL(1, 1); }
  else { if (ppt == 4) S360_RL(2, 4); else if (ppt == 2) S360_RL(2, 2); else S360_RL(2, 1); }
#undef S360_RL
}
void launch_resize_cubic_f32c2(hipStream_t st, const float2* src, int sw, int sh, size_t sbs, float2* dst, int dw,
                               int dh, size_t dbs, int B, float post_scale, const float2* const* src_tab) {
  const double scx = 1.0 / ((double)dw / (double)sw), scy = 1.0 / ((double)dh / (double)sh);
  if (!src_tab && scx <= 1.0 && scy <= 1.0) {  // upscale: a 64x16 tile reads at most (64 + 4) x (16 + 4) source pixels
    hipLaunchKernelGGL(k_resize_cubic_f32c2_tiled, dim3((dw + UC_TW - 1) / UC_TW, (dh + UC_TH - 1) / UC_TH, B), dim3(256),
                       0, st, src, sw, sh, sbs, dst, dw, dh, dbs, scx, scy, post_scale);
    return;
  }
  dim3 blk(32, 8);
  hipLaunchKernelGGL(k_resize_cubic_f32c2, grid2d(dw, dh, B, blk), blk, 0, st, src, sw, sh, sbs, dst, dw, dh, dbs, scx,
                     scy, post_scale, src_tab);
}
void launch_search_init(hipStream_t st, const float* I, const float* A, int w, int h, size_t pbs, int B,
                        const FlowIdx& idx, float2* flow, int hint, int dist, float* I1eq) {
  hipLaunchKernelGGL(k_search_init, dim3(B), dim3(256), 0, st, I, A, w, h, pbs, idx, flow, hint, dist, I1eq);
}

}  // namespace s360
