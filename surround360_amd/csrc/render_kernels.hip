// render_kernels.hip — reprojection, novel-view warp, strip blend and pole composite kernels for
// gfx950. All of these are gather/stream kernels bounded by HBM/L2 bandwidth (no dense
// contraction, no MFMA). 8-bit arithmetic follows OpenCV's fixed-point remap;
// float arithmetic follows the reference's operation order (compiled with -ffp-contract=off).
//
// References: SR/render/ImageWarper.cpp:143-174, SR/optical_flow/NovelView.cpp:101-268,
// SR/test/TestRenderStereoPanorama.cpp:99-135, 259-292, 380-384, 388-561, 701-713,
// SR/util/CvUtil.cpp:93-115, 140-157, 224-260, SR/util/Filter.h:40-127.
#include "render_kernels.hpp"

#include <cstdint>
#include <cstdlib>

#include <algorithm>

#include <stdexcept>

#include "devmath.hpp"

namespace s360 {

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bgr_to_bgra(const uint8_t* __restrict__ src, int cn, uchar4* __restrict__ dst,
                                                     size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t* p = src + i * cn;
  dst[i] = make_uchar4(p[0], p[1], p[2], cn == 4 ? p[3] : 255);
}
__global__ __launch_bounds__(256) void k_prepare_side_src(const uint8_t* __restrict__ src, int cn,
                                                          uchar4* __restrict__ dst, int w, int h, int feather) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  const uint8_t* p = src + ((size_t)y * w + x) * cn;
  int a = cn == 4 ? p[3] : 255;
  // rows [0,feather) and [h-feather,h): alpha = uint8(255.0f * float(yy + 0.5f) / float(feather)), yy being
  // the loop iteration of TRSP:117-124 that touches this row.
  int yy = -1;
  if (y < feather) yy = y;
  if (h - 1 - y < feather) yy = max(yy, h - 1 - y);  // the later loop iteration wins
  if (yy >= 0) a = (int)(unsigned char)(255.0f * (float)(yy + 0.5f) / (float)feather);
  dst[(size_t)y * w + x] = make_uchar4(p[0], p[1], p[2], (unsigned char)a);
}

// ------------------------------------------------------------------------------------------
// Warp map of bicubicRemapToSpherical: unit vector products in float (cosf/sinf tables from the host's
// libm, separable in x and y), Camera::pixel in double, stored as float (pixel - 0.5).
__global__ __launch_bounds__(256) void k_spherical_map(float2* __restrict__ map, int dw, int dh, DevCamera cam,
                                                       const float* __restrict__ cosX, const float* __restrict__ sinX,
                                                       const float* __restrict__ cosY,
                                                       const float* __restrict__ sinY) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= dw || y >= dh) return;
  const float ux = cosY[y] * cosX[x];
  const float uy = cosY[y] * sinX[x];
  const float uz = sinY[y];
  const int kNear = 1000000;  // int(Camera::kNearInfinity)
  const double dx = (double)ux * kNear - cam.pos[0], dy = (double)uy * kNear - cam.pos[1],
               dz = (double)uz * kNear - cam.pos[2];
  const double cx = cam.R[0] * dx + cam.R[1] * dy + cam.R[2] * dz;
  const double cy = cam.R[3] * dx + cam.R[4] * dy + cam.R[5] * dz;
  const double cz = cam.R[6] * dx + cam.R[7] * dy + cam.R[8] * dz;
  double sx, sy;
  if (cam.type == 0) {  // FTHETA
    const double norm = sqrt(cx * cx + cy * cy);
    const double r = atan2(norm, -cz);
    const double r2 = r * r;
    const double f = ((1 + r2 * (cam.distortion[0] + r2 * cam.distortion[1])) * r) / norm;
    sx = f * cx;
    sy = f * cy;
  } else {  // RECTILINEAR
    const double px = cx / -cz, py = cy / -cz;
    const double r2 = px * px + py * py;
    const double f = 1 + r2 * (cam.distortion[0] + r2 * cam.distortion[1]);
    sx = f * px;
    sy = f * py;
  }
  const double pxx = cam.focal[0] * sx + cam.principal[0];
  const double pyy = cam.focal[1] * sy + cam.principal[1];
  map[(size_t)y * dw + x] = make_float2((float)(pxx - 0.5), (float)(pyy - 0.5));
}

// ------------------------------------------------------------------------------------------
// cv::remap INTER_CUBIC, BORDER_CONSTANT(0), 8UC4: integer weights (sum 32768), (sum + 2^14) >> 15.
__device__ __forceinline__ uchar4 remap_cubic_u8c4_at(const uchar4* __restrict__ src, int sw, int sh, float mx,
                                                      float my, const short* __restrict__ tab) {
  int sx, sy, fxy;
  remap_coord(mx, my, &sx, &sy, &fxy);
  if (sx >= sw || sx + 4 <= 0 || sy >= sh || sy + 4 <= 0) return make_uchar4(0, 0, 0, 0);
  if (sx >= 0 && sx + 4 <= sw && sy >= 0 && sy + 4 <= sh) {
    // interior: four 16-byte row loads (4-byte aligned) and the 64 multiply-adds as v_perm_b32 + v_dot2_i32_i16 — the
    // table holds the weights of two neighbouring taps per dword (integer sums: the order does not matter)
    typedef unsigned u4a4 __attribute__((ext_vector_type(4), aligned(4)));
    typedef short s16x2_ __attribute__((ext_vector_type(2)));
    const uint4* w4 = reinterpret_cast<const uint4*>(tab + fxy * 16);
    const uint4 wa = w4[0], wb = w4[1];
    const unsigned wq[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
    int acc[4] = {0, 0, 0, 0};
    u4a4 prow[4];  // (the four row loads go out together, behind the two weight loads above)
#pragma unroll
    for (int r = 0; r < 4; ++r) prow[r] = *reinterpret_cast<const u4a4*>(src + (size_t)(sy + r) * sw + sx);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const u4a4 p = prow[r];
      const s16x2_ w01 = __builtin_bit_cast(s16x2_, wq[2 * r]), w23 = __builtin_bit_cast(s16x2_, wq[2 * r + 1]);
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) {
        const unsigned sel = 0x0c040c00u + ch * 0x00010001u;
        const s16x2_ lo = __builtin_bit_cast(s16x2_, __builtin_amdgcn_perm(p.y, p.x, sel));
        const s16x2_ hi = __builtin_bit_cast(s16x2_, __builtin_amdgcn_perm(p.w, p.z, sel));
        acc[ch] = __builtin_amdgcn_sdot2(lo, w01, acc[ch], false);
        acc[ch] = __builtin_amdgcn_sdot2(hi, w23, acc[ch], false);
      }
    }
    return make_uchar4((unsigned char)sat_u8((acc[0] + (1 << 14)) >> 15), (unsigned char)sat_u8((acc[1] + (1 << 14)) >> 15),
                       (unsigned char)sat_u8((acc[2] + (1 << 14)) >> 15), (unsigned char)sat_u8((acc[3] + (1 << 14)) >> 15));
  }
  const short* w = tab + fxy * 16;
  int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int yi = sy + r;
    if (yi < 0 || yi >= sh) continue;
    const uchar4* S = src + (size_t)yi * sw;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int xi = sx + q;
      if (xi < 0 || xi >= sw) continue;
      const uchar4 p = S[xi];
      const int ww = w[r * 4 + q];
      s0 += __mul24((int)p.x, ww); s1 += __mul24((int)p.y, ww); s2 += __mul24((int)p.z, ww); s3 += __mul24((int)p.w, ww);  // 8-bit x 16-bit
    }
  }
  return make_uchar4((unsigned char)sat_u8((s0 + (1 << 14)) >> 15), (unsigned char)sat_u8((s1 + (1 << 14)) >> 15),
                     (unsigned char)sat_u8((s2 + (1 << 14)) >> 15), (unsigned char)sat_u8((s3 + (1 << 14)) >> 15));
}
// cv::remap INTER_CUBIC, BORDER_CONSTANT(0), 32FC2 (NovelView.cpp:191): float weights; interior sums row by
// row (4-term left-associated, then sum += row), border taps are added one by one.
__device__ __forceinline__ float2 remap_cubic_f32c2_at(const float2* __restrict__ src, int sw, int sh, float mx,
                                                       float my, const float* __restrict__ tab) {
  int sx, sy, fxy;
  remap_coord(mx, my, &sx, &sy, &fxy);
  const float* w = tab + fxy * 16;
  const unsigned width1 = sw - 3 > 0 ? sw - 3 : 0, height1 = sh - 3 > 0 ? sh - 3 : 0;
  float2 o;
  if ((unsigned)sx < width1 && (unsigned)sy < height1) {
    typedef float f4a8_ __attribute__((ext_vector_type(4), aligned(8)));
    // all twelve loads (4 rows x two 16-byte pixel pairs, 4 x four weights) are requested before the first is used: as
    // load-use pairs row by row they were twelve serialised memory round trips per sample
    const float2* S = src + (size_t)sy * sw + sx;
    f4a8_ ab[4], cd[4];
    float4 wr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const f4a8_* V = reinterpret_cast<const f4a8_*>(S + (size_t)r * sw);  // two pixels per 16-byte load
      ab[r] = V[0];
      cd[r] = V[1];
      wr[r] = *reinterpret_cast<const float4*>(w + 4 * r);
    }
    o.x = ab[0].x * wr[0].x + ab[0].z * wr[0].y + cd[0].x * wr[0].z + cd[0].z * wr[0].w;
    o.y = ab[0].y * wr[0].x + ab[0].w * wr[0].y + cd[0].y * wr[0].z + cd[0].w * wr[0].w;
#pragma unroll
    for (int r = 1; r < 4; ++r) {
      o.x += ab[r].x * wr[r].x + ab[r].z * wr[r].y + cd[r].x * wr[r].z + cd[r].z * wr[r].w;
      o.y += ab[r].y * wr[r].x + ab[r].w * wr[r].y + cd[r].y * wr[r].z + cd[r].w * wr[r].w;
    }
  } else if (sx >= sw || sx + 4 <= 0 || sy >= sh || sy + 4 <= 0) {
    o = make_float2(0.f, 0.f);
  } else {
    o = make_float2(0.f, 0.f);
    for (int r = 0; r < 4; ++r) {
      const int yi = sy + r;
      if (yi < 0 || yi >= sh) continue;
      for (int q = 0; q < 4; ++q) {
        const int xi = sx + q;
        if (xi < 0 || xi >= sw) continue;
        const float2 p = src[(size_t)yi * sw + xi];
        o.x += p.x * w[r * 4 + q];
        o.y += p.y * w[r * 4 + q];
      }
    }
  }
  return o;
}

// ------------------------------------------------------------------------------------------
// Tiled bicubic remap: one workgroup renders a 64x8 tile of the destination. The source taps of the tile lie in a
// small box (the maps on this path are smooth); the box is found with wavefront min/max reductions of the tap
// origins, loaded ONCE into LDS with coalesced row reads (zero outside the image = BORDER_CONSTANT(0), so the taps
// need no bounds checks), and the 16 taps per pixel are LDS reads. Tiles whose box does not fit fall back to the
// per-tap global gather. Integer arithmetic: the result does not depend on the summation order.
constexpr int RT_W = 64, RT_H = 8, RT_CAP = 4608;
typedef short s16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = min(v, __shfl_xor(v, m, 64));
  return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m, 64));
  return v;
}

struct MapFromBuffer {  // bicubicRemapToSpherical: cached warp map (ImageWarper.cpp:151-173)
  const float2* map;
  int dw;
  __device__ __forceinline__ float2 operator()(int x, int y) const { return map[(size_t)y * dw + x]; }
  // the same in two steps, so that a thread can request the values of several pixels before it uses the first
  __device__ __forceinline__ float2 load(int x, int y) const { return map[(size_t)y * dw + x]; }
  __device__ __forceinline__ float2 apply(int, int, float2 v) const { return v; }
  __device__ __forceinline__ void advance(size_t n) { map += n; }
};
struct MapFromPoleFlow {  // poleToSideFlowThread's ramped warp (TRSP:483-503)
  const float2* flow;
  PoleWarpParams pw;
  __device__ __forceinline__ float2 operator()(int x, int y) const {
    const float phi = pw.poleCameraRadius * (float)(y + 0.5f) / (float)pw.rows;
    const float alpha = 1.0f - rampf(phi, pw.phiRampStart, pw.phiMid);
    const float2 f = flow[(size_t)y * pw.extW + x];
    return make_float2((float)x + (1.0f - alpha) * f.x, (float)y + (1.0f - alpha) * f.y);
  }
  __device__ __forceinline__ float2 load(int x, int y) const { return flow[(size_t)y * pw.extW + x]; }
  __device__ __forceinline__ float2 apply(int x, int y, float2 f) const {
    const float phi = pw.poleCameraRadius * (float)(y + 0.5f) / (float)pw.rows;
    const float alpha = 1.0f - rampf(phi, pw.phiRampStart, pw.phiMid);
    return make_float2((float)x + (1.0f - alpha) * f.x, (float)y + (1.0f - alpha) * f.y);
  }
  __device__ __forceinline__ void advance(size_t n) { flow += n; }
};

struct MapFromFlowAdd {  // PoleRemoval.cpp:128-133: warp = (x, y) + flow
  const float2* flow;
  int w;
  __device__ __forceinline__ float2 operator()(int x, int y) const {
    const float2 f = flow[(size_t)y * w + x];
    return make_float2((float)x + f.x, (float)y + f.y);
  }
  __device__ __forceinline__ void advance(size_t n) { flow += n; }
};

template <class MapFn>
__global__ __launch_bounds__(RT_W* RT_H) void k_remap_cubic_u8c4_tiled(const uchar4* __restrict__ src, int sw, int sh,
                                                                       MapFn mapfn, uchar4* __restrict__ dst, int dw,
                                                                       int dh, const short* __restrict__ tab,
                                                                       int alpha_mode, int yFeatherStart,
                                                                       int featherSize, size_t sbs, size_t dbs) {
  // blockIdx.z = image of a batch with identical geometry (the side cameras): sources sbs, maps / outputs dbs apart
  src += sbs * blockIdx.z;
  dst += dbs * blockIdx.z;
  mapfn.advance(dbs * blockIdx.z);
  __shared__ uchar4 s_tile[RT_CAP];
  __shared__ int s_box[4];
  const int x = blockIdx.x * RT_W + threadIdx.x, y = blockIdx.y * RT_H + threadIdx.y;
  const int tid = threadIdx.y * RT_W + threadIdx.x;
  const bool inside = x < dw && y < dh;
  int sx = 0, sy = 0, fxy = 0;
  bool live = false;
  float2 m = make_float2(0.f, 0.f);
  if (inside) {
    m = mapfn(x, y);
    remap_coord(m.x, m.y, &sx, &sy, &fxy);
    live = !(sx >= sw || sx + 4 <= 0 || sy >= sh || sy + 4 <= 0);
  }
  if (tid == 0) { s_box[0] = INT_MAX; s_box[1] = INT_MIN; s_box[2] = INT_MAX; s_box[3] = INT_MIN; }
  const int mnx = wave_min_i(live ? sx : INT_MAX), mxx = wave_max_i(live ? sx : INT_MIN);
  const int mny = wave_min_i(live ? sy : INT_MAX), mxy = wave_max_i(live ? sy : INT_MIN);
  __syncthreads();
  if ((tid & 63) == 0 && mnx <= mxx) {
    atomicMin(&s_box[0], mnx); atomicMax(&s_box[1], mxx);
    atomicMin(&s_box[2], mny); atomicMax(&s_box[3], mxy);
  }
  __syncthreads();
  const int bx0 = s_box[0], by0 = s_box[2];
  const bool any = bx0 <= s_box[1];
  const int bw = any ? s_box[1] + 4 - bx0 : 0, bh = any ? s_box[3] + 4 - by0 : 0;
  uchar4 o = make_uchar4(0, 0, 0, 0);
  if (any && (long long)bw * bh <= RT_CAP) {
    for (int ly = threadIdx.y; ly < bh; ly += RT_H) {
      const int gy = by0 + ly;
      const bool rowIn = gy >= 0 && gy < sh;
      const uchar4* S = src + (size_t)(rowIn ? gy : 0) * sw;
      for (int lx = threadIdx.x; lx < bw; lx += RT_W) {
        const int gx = bx0 + lx;
        s_tile[ly * bw + lx] = (rowIn && gx >= 0 && gx < sw) ? S[gx] : make_uchar4(0, 0, 0, 0);
      }
    }
    __syncthreads();
    if (live) {
      // 16 taps x 4 channels as v_dot2_i32_i16: the table already holds the weights of two neighbouring taps in one
      // dword; v_perm_b32 puts one channel of two neighbouring pixels into the halves of the other operand.
      const uint4* w4 = reinterpret_cast<const uint4*>(tab + fxy * 16);
      const uint4 wa = w4[0], wb = w4[1];
      const unsigned wq[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
      const unsigned* T = reinterpret_cast<const unsigned*>(s_tile) + (sy - by0) * bw + (sx - bx0);
      int acc[4] = {0, 0, 0, 0};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned p0 = T[r * bw], p1 = T[r * bw + 1], p2 = T[r * bw + 2], p3 = T[r * bw + 3];
        const s16x2 w01 = __builtin_bit_cast(s16x2, wq[2 * r]), w23 = __builtin_bit_cast(s16x2, wq[2 * r + 1]);
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
          const unsigned sel = 0x0c040c00u + ch * 0x00010001u;
          const s16x2 lo = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(p1, p0, sel));
          const s16x2 hi = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(p3, p2, sel));
          acc[ch] = __builtin_amdgcn_sdot2(lo, w01, acc[ch], false);
          acc[ch] = __builtin_amdgcn_sdot2(hi, w23, acc[ch], false);
        }
      }
      o = make_uchar4((unsigned char)sat_u8((acc[0] + (1 << 14)) >> 15), (unsigned char)sat_u8((acc[1] + (1 << 14)) >> 15),
                      (unsigned char)sat_u8((acc[2] + (1 << 14)) >> 15), (unsigned char)sat_u8((acc[3] + (1 << 14)) >> 15));
    }
  } else if (live) {
    o = remap_cubic_u8c4_at(src, sw, sh, m.x, m.y, tab);
  }
  if (!inside) return;
  if (alpha_mode == 1) {
    // remap ran on 3 channels, cvtColor BGR2BGRA sets 255, the feather loop overwrites the last rows
    int a = 255;
    if (y >= yFeatherStart) {
      const float alpha = 1.0f - (float)(y - yFeatherStart) / (float)featherSize;
      a = (int)(unsigned char)(255.0f * alpha);
    }
    o.w = (unsigned char)a;
  } else if (alpha_mode == 2) {
    // 4-channel source (pole removal result): the interpolated alpha is kept, the feather rows take the minimum
    // (TRSP:625-634)
    if (y >= yFeatherStart) {
      const float alpha = 1.0f - (float)(y - yFeatherStart) / (float)featherSize;
      const unsigned char a = (unsigned char)(255.0f * alpha);
      o.w = o.w < a ? o.w : a;
    }
  }
  dst[(size_t)y * dw + x] = o;
}

// ------------------------------------------------------------------------------------------
// Packed bicubic remap: the same arithmetic as k_remap_cubic_u8c4_tiled with everything that depends only on the MAP
// prepared by a kernel of its own (k_remap_pack) — once per rig for the cached spherical maps of the side and pole
// cameras, once per frame for the pole warp: per destination tile of 64 x 16 pixels the box of source pixels its taps
// touch, and per pixel ONE dword {live | tap origin relative to the box (11 + 10 bits) | 1/32-pixel fraction index
// (10 bits)} instead of the 8-byte float map. The remap kernel then has one memory phase: the tile's packed coordinates
// and its source box are requested together (no coordinate arithmetic, wave reductions, LDS atomics or barrier in front
// of the box load), one barrier, 16 LDS taps per pixel folded with v_perm_b32 + v_dot2_i32_i16. 256 threads render 4
// pixels each. Tiles whose box does not fit (map singularities) take the per-tap global gather from the float map.
constexpr int PT_W = 64, PT_H = 16, PT_TY = 4, PT_CAP = 4096;

template <class MapFn>
__global__ __launch_bounds__(PT_W* PT_TY) void k_remap_pack(MapFn mapfn, int sw, int sh, int dw, int dh,
                                                            unsigned* __restrict__ packed, int4* __restrict__ tiles,
                                                            size_t dbs, int tilesPerImage) {
  mapfn.advance(dbs * blockIdx.z);
  packed += dbs * blockIdx.z;
  tiles += (size_t)tilesPerImage * blockIdx.z;
  __shared__ int s_box[4];
  const int tid = threadIdx.y * PT_W + threadIdx.x;
  const int x = blockIdx.x * PT_W + threadIdx.x;
  int sx[4], sy[4], fxy[4];
  bool live[4];
  int mnx = INT_MAX, mxx = INT_MIN, mny = INT_MAX, mxy = INT_MIN;
  float2 raw[4];  // (the four map / flow values are requested together)
#pragma unroll
  for (int k = 0; k < 4; ++k) raw[k] = mapfn.load(min(x, dw - 1), min((int)(blockIdx.y * PT_H + threadIdx.y + PT_TY * k), dh - 1));
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int y = blockIdx.y * PT_H + threadIdx.y + PT_TY * k;
    live[k] = false;
    sx[k] = sy[k] = fxy[k] = 0;
    if (x < dw && y < dh) {
      const float2 m = mapfn.apply(x, y, raw[k]);
      remap_coord(m.x, m.y, &sx[k], &sy[k], &fxy[k]);
      live[k] = !(sx[k] >= sw || sx[k] + 4 <= 0 || sy[k] >= sh || sy[k] + 4 <= 0);
      if (live[k]) {
        mnx = min(mnx, sx[k]); mxx = max(mxx, sx[k]);
        mny = min(mny, sy[k]); mxy = max(mxy, sy[k]);
      }
    }
  }
  if (tid == 0) { s_box[0] = INT_MAX; s_box[1] = INT_MIN; s_box[2] = INT_MAX; s_box[3] = INT_MIN; }
  mnx = wave_min_i(mnx); mxx = wave_max_i(mxx);
  mny = wave_min_i(mny); mxy = wave_max_i(mxy);
  __syncthreads();
  if ((tid & 63) == 0 && mnx <= mxx) {
    atomicMin(&s_box[0], mnx); atomicMax(&s_box[1], mxx);
    atomicMin(&s_box[2], mny); atomicMax(&s_box[3], mxy);
  }
  __syncthreads();
  const int bx0 = s_box[0], by0 = s_box[2];
  const bool any = bx0 <= s_box[1];
  int bw = any ? s_box[1] + 4 - bx0 : 0, bh = any ? s_box[3] + 4 - by0 : 0;
  if (any && ((long long)bw * bh > PT_CAP || bw > 2047 || bh > 1023)) bh = -1;  // too large for the LDS tile / the packed fields
  if (tid == 0) tiles[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = make_int4(bx0, by0, bw, bh);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int y = blockIdx.y * PT_H + threadIdx.y + PT_TY * k;
    if (x < dw && y < dh)
      packed[(size_t)y * dw + x] = (live[k] && bh > 0) ? (0x80000000u | ((unsigned)(sy[k] - by0) << 21) | ((unsigned)(sx[k] - bx0) << 10) | (unsigned)fxy[k]) : 0u;
  }
}

// Round 5: the alpha mode is a template parameter — mode 1 (3-channel source: the pole projections) evaluates three channels; a
// quarter of the tap arithmetic of a 17.7 Mpx image was computed and then overwritten with 255 — and the rounding constant is
// the accumulators' start value: side projections 0.332 -> 0.300 ms per 8K frame, pole projections 0.269 -> 0.261, pole warp
// 0.813 -> 0.793.
// Measured and NOT adopted: 64 x 64 destination tiles for the pole projections (their 64 x 16 tiles' boxes are ~20 x 20 source
// pixels; a tall tile's ~31 x 31 for four times the taps per workgroup and per chain of dependent round trips): 0.262 against
// 0.261 ms per frame (profiles/r05_v9_remap_tall_tiles.json) — that chain is not what the kernel waits for either.
// Measured twice and NOT adopted: the 32 KB weight table in LDS. It has to be loaded once per workgroup, not once per 64 x 16 tile
// (that would be the same 32 bytes per pixel again), i.e. PERSISTENT workgroups — 3 per CU at 48 KB of LDS, each XCD's
// workgroups striding through its contiguous run of tiles. (1) As it stands, load -> barrier -> taps -> barrier per tile: 0.422 /
// 0.332 / 1.051 ms (profiles/r05_v2_persistent_remap_warp_blend.json). (2) With the next tile's packed coordinates and source
// box in flight into registers while this tile's taps run (NI x NJ dwords per thread, the record of the tile after that on its
// way; rows / columns beyond the pattern loaded when the tile is stored): 0.384 / 0.286 / 0.979 ms against 0.300 / 0.261 /
// 0.793 of this form on the same box (profiles/r05_v4_remap_persistent_prefetch_ab.json). Twelve waves per CU whose phases line
// up leave the VALUs idle more than the weight rows' trips to L1 / L2 cost the 32 resident waves of this form (VALU 63 % / 53 %
// busy, profiles/r05_v3_valu_busy.txt); both variants were bit-exact on the emulation and on the GPU and are gone.
// WT (round 6): the 16 weights of a pixel are REBUILT from what initInterTab2D makes them of — the 1-D cubic
// taps of the pixel's two fractions (two 16-byte LDS reads), eight packed multiplies, the round-to-even as eight packed adds of
// 1.5 * 2^23 (the integer is then the low 16 bits of the sum), eight v_perm_b32 to pack them, and the entry's rounding residue
// (one LDS read, a packed 16-bit add on its tap) — instead of fetched as 32 bytes from the 32 KB table through L1 / L2. The
// host has rebuilt all 1024 entries the same way and compared them (Tables::build); bit-identical by construction.
// Measured (round 6, alternating on one box, 8K frame, profiles/r06_v5_remap_rebuilt_weights_ab.txt): side projections 0.297 ->
// 0.272 ms per frame in a batch (0.302 -> 0.287 alone), pole projections 0.263 -> 0.236; the pole warp 0.787 -> 0.865 (it computes
// more than it waits) and keeps the table. The default for the MapFromBuffer instantiations; S360_REMAP_REBUILD_WEIGHTS=0 = table.
typedef float f2v __attribute__((ext_vector_type(2)));
typedef unsigned short us2v __attribute__((ext_vector_type(2)));
template <class MapFn, int ALPHA, bool WT = false>
__global__ __launch_bounds__(PT_W* PT_TY) void k_remap_cubic_u8c4_packed(const uchar4* __restrict__ src, int sw, int sh,
                                                                         const unsigned* __restrict__ packed,
                                                                         const int4* __restrict__ tiles, MapFn mapfn,
                                                                         uchar4* __restrict__ dst, int dw, int dh,
                                                                         const short* __restrict__ tab,
                                                                         int yFeatherStart, int featherSize, size_t sbs,
                                                                         size_t dbs, int tilesPerImage,
                                                                         const float* __restrict__ w1, const short* __restrict__ wres) {
  __shared__ __attribute__((aligned(16))) float s_w1[WT ? 256 : 4];
  __shared__ short s_res[WT ? 1024 : 2];
  const TileId tile = xcd_tile();  // neighbouring tiles (overlapping source boxes) on the same XCD's L2
  src += sbs * tile.z;
  dst += dbs * tile.z;
  packed += dbs * tile.z;
  mapfn.advance(dbs * tile.z);
  __shared__ uchar4 s_tile[PT_CAP];
  const int4 box = tiles[(size_t)tilesPerImage * tile.z + (size_t)tile.y * gridDim.x + tile.x];  // (uniform)
  const int bx0 = box.x, by0 = box.y, bw = box.z, bh = box.w;
  const int x = tile.x * PT_W + threadIdx.x;
  unsigned pk[4] = {0u, 0u, 0u, 0u};
  if (bh > 0 && x < dw) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int y = tile.y * PT_H + threadIdx.y + PT_TY * k;
      if (y < dh) pk[k] = packed[(size_t)y * dw + x];
    }
  }
  if (bh > 0) {  // the tile's source box, zero outside the image (BORDER_CONSTANT): requested together with the coordinates
    // eight pixels (4 rows x 2 column groups) are requested before the first goes to LDS: as load-store pairs in a
    // runtime loop the ~7 pixels of a thread were as many serialised memory round trips
    const unsigned* S32 = reinterpret_cast<const unsigned*>(src);
    unsigned* T32 = reinterpret_cast<unsigned*>(s_tile);
    if (WT) {  // 1 KB of taps + 2 KB of residues per workgroup (against 32 bytes per pixel = 32 KB per tile from the table)
      const int tid = threadIdx.y * PT_W + threadIdx.x;
      s_w1[tid] = w1[tid];
      reinterpret_cast<uint2*>(s_res)[tid] = reinterpret_cast<const uint2*>(wres)[tid];
    }
    for (int ly0 = threadIdx.y; ly0 < bh; ly0 += 4 * PT_TY)
      for (int lx0 = threadIdx.x; lx0 < bw; lx0 += 2 * PT_W) {
        unsigned v[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int ly = ly0 + j * PT_TY, gy = by0 + ly;
          const bool rowIn = ly < bh && gy >= 0 && gy < sh;
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int lx = lx0 + i * PT_W, gx = bx0 + lx;
            v[j][i] = (rowIn && lx < bw && gx >= 0 && gx < sw) ? S32[(size_t)gy * sw + gx] : 0u;
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int ly = ly0 + j * PT_TY;
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int lx = lx0 + i * PT_W;
            if (ly < bh && lx < bw) T32[ly * bw + lx] = v[j][i];
          }
        }
      }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int y = tile.y * PT_H + threadIdx.y + PT_TY * k;
    if (x >= dw || y >= dh) continue;
    uchar4 o = make_uchar4(0, 0, 0, 0);
    if (bh > 0) {
      if (pk[k] & 0x80000000u) {
        const int rx = (pk[k] >> 10) & 2047, ry = (pk[k] >> 21) & 1023;
        // (the weight rows are fetched here, per pixel: requesting all four in front of the barrier cost 30 VGPRs and
        // measured 10 % slower, profiles/r03_v7 vs r3i)
        unsigned wq[8];
        if (WT) {
          const unsigned fxy = pk[k] & 1023u;
          const float4 wy = *reinterpret_cast<const float4*>(&s_w1[(fxy >> 5) * 4]);
          const float4 wx = *reinterpret_cast<const float4*>(&s_w1[128 + (fxy & 31u) * 4]);  // (x's taps carry the 2^15)
          const int rr = s_res[fxy];
          const f2v x01 = {wx.x, wx.y}, x23 = {wx.z, wx.w};
          const float wyr[4] = {wy.x, wy.y, wy.z, wy.w};
          const f2v magic = {12582912.0f, 12582912.0f};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            f2v a = x01 * wyr[r], b = x23 * wyr[r];
            if (r == 1) a.y = fminf(a.y, 32767.0f);  // saturate_cast<short>: the one product that reaches 32768 (both fractions 0)
            a += magic;
            b += magic;
            wq[2 * r] = __builtin_amdgcn_perm(__float_as_uint(a.y), __float_as_uint(a.x), 0x05040100u);
            wq[2 * r + 1] = __builtin_amdgcn_perm(__float_as_uint(b.y), __float_as_uint(b.x), 0x05040100u);
          }
          // the entry's residue on tap (2,2) (2,3) (3,2) or (3,3), added in 16 bits like the table's `(short)(itab - diff)`
          const unsigned add = ((unsigned)(rr >> 2) & 0xffffu) << (16 * (rr & 1));
          const unsigned a5 = (rr & 2) ? 0u : add, a7 = (rr & 2) ? add : 0u;
          wq[5] = __builtin_bit_cast(unsigned, (us2v)(__builtin_bit_cast(us2v, wq[5]) + __builtin_bit_cast(us2v, a5)));
          wq[7] = __builtin_bit_cast(unsigned, (us2v)(__builtin_bit_cast(us2v, wq[7]) + __builtin_bit_cast(us2v, a7)));
        } else {
          const uint4* w4 = reinterpret_cast<const uint4*>(tab + (pk[k] & 1023u) * 16);
          const uint4 wa = w4[0], wb = w4[1];
          wq[0] = wa.x; wq[1] = wa.y; wq[2] = wa.z; wq[3] = wa.w; wq[4] = wb.x; wq[5] = wb.y; wq[6] = wb.z; wq[7] = wb.w;
        }
        const unsigned* T = reinterpret_cast<const unsigned*>(s_tile) + ry * bw + rx;
        int acc[4] = {1 << 14, 1 << 14, 1 << 14, 1 << 14};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const unsigned p0 = T[r * bw], p1 = T[r * bw + 1], p2 = T[r * bw + 2], p3 = T[r * bw + 3];
          const s16x2 w01 = __builtin_bit_cast(s16x2, wq[2 * r]), w23 = __builtin_bit_cast(s16x2, wq[2 * r + 1]);
#pragma unroll
          for (int ch = 0; ch < (ALPHA == 1 ? 3 : 4); ++ch) {
            const unsigned sel = 0x0c040c00u + ch * 0x00010001u;
            const s16x2 lo = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(p1, p0, sel));
            const s16x2 hi = __builtin_bit_cast(s16x2, __builtin_amdgcn_perm(p3, p2, sel));
            acc[ch] = __builtin_amdgcn_sdot2(lo, w01, acc[ch], false);
            acc[ch] = __builtin_amdgcn_sdot2(hi, w23, acc[ch], false);
          }
        }
        o = make_uchar4((unsigned char)sat_u8(acc[0] >> 15), (unsigned char)sat_u8(acc[1] >> 15), (unsigned char)sat_u8(acc[2] >> 15),
                        ALPHA == 1 ? (unsigned char)0 : (unsigned char)sat_u8(acc[3] >> 15));
      }
    } else if (bh < 0) {  // box too large for LDS: per-tap gather through the map itself
      const float2 m = mapfn(x, y);
      int sx, sy, fxy;
      remap_coord(m.x, m.y, &sx, &sy, &fxy);
      if (!(sx >= sw || sx + 4 <= 0 || sy >= sh || sy + 4 <= 0)) o = remap_cubic_u8c4_at(src, sw, sh, m.x, m.y, tab);
    }
    if (ALPHA == 1) {
      // remap ran on 3 channels, cvtColor BGR2BGRA sets 255, the feather loop overwrites the last rows
      int a = 255;
      if (y >= yFeatherStart) {
        const float alpha = 1.0f - (float)(y - yFeatherStart) / (float)featherSize;
        a = (int)(unsigned char)(255.0f * alpha);
      }
      o.w = (unsigned char)a;
    } else if (ALPHA == 2) {
      // 4-channel source (pole removal result): the interpolated alpha is kept, the feather rows take the minimum
      // (TRSP:625-634)
      if (y >= yFeatherStart) {
        const float alpha = 1.0f - (float)(y - yFeatherStart) / (float)featherSize;
        const unsigned char a = (unsigned char)(255.0f * alpha);
        o.w = o.w < a ? o.w : a;
      }
    }
    dst[(size_t)y * dw + x] = o;
  }
}

// ---- pole removal (PoleRemoval.cpp:32-188) -------------------------------------------------------------------
// "is pure red" plane of a BGR mask image (cutRedMaskOutOfAlphaChannel's test, CvUtil.cpp:213-222)
__global__ __launch_bounds__(256) void k_red_mask(const uint8_t* __restrict__ bgr, uint8_t* __restrict__ red, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t* p = bgr + i * 3;
  red[i] = (p[0] == 0 && p[1] == 0 && p[2] == 255) ? 1 : 0;
}
// circleAlphaCut (CvUtil.cpp:201-211) [+ cutRedMaskOutOfAlphaChannel when red != nullptr]; colours copied from src
__global__ __launch_bounds__(256) void k_circle_alpha(const uchar4* __restrict__ src, const uint8_t* __restrict__ red,
                                                      uchar4* __restrict__ dst, int w, int h, float radius) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  const size_t i = (size_t)y * w + x;
  const float dx = (float)x - (float)w / 2.0f;
  const float dy = (float)y - (float)h / 2.0f;
  const float r = sqrtf(dx * dx + dy * dy);
  const float alpha = r < radius ? 1.0f : 0.0f;
  uchar4 p = src[i];
  p.w = (unsigned char)(alpha * 255.0f);
  if (red && red[i]) p.w = 0;
  dst[i] = p;
}
// the alpha-weighted merge of the primary bottom image with the warped secondary one (PoleRemoval.cpp:153-180)
__global__ __launch_bounds__(256) void k_pole_removal_combine(uchar4* __restrict__ bottom, const uchar4* __restrict__ warped2,
                                                              size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uchar4 p1 = bottom[i];
  const uchar4 p2 = warped2[i];
  const float alpha = (float)p1.w / 255.0f, alpha2 = (float)p2.w / 255.0f;
  if (alpha < 1.0f && alpha2 > 0.0f) {
    const float a1 = alpha, a2 = 1.0f - alpha;
    p1.x = (unsigned char)(a1 * (float)p1.x + a2 * (float)p2.x);
    p1.y = (unsigned char)(a1 * (float)p1.y + a2 * (float)p2.y);
    p1.z = (unsigned char)(a1 * (float)p1.z + a2 * (float)p2.z);
    p1.w = 255;
    bottom[i] = p1;
  }
}

// two pixels per thread (camW and overlapW even: both sides of the copy are 8-byte aligned)
__global__ __launch_bounds__(256) void k_crop_overlaps_v2(const uint2* __restrict__ proj, int camW2, int camH, int P,
                                                          int overlapW2, uint2* __restrict__ out, int p0, int n) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, j = blockIdx.z;
  if (x >= overlapW2) return;
  const int cam = j < n ? p0 + j : (p0 + (j - n) + 1) % P;
  const int x0 = j < n ? camW2 - overlapW2 : 0;
  out[((size_t)j * camH + y) * overlapW2 + x] = proj[((size_t)cam * camH + y) * camW2 + x0 + x];
}
__global__ __launch_bounds__(256) void k_crop_overlaps(const uchar4* __restrict__ proj, int camW, int camH, int P,
                                                       int overlapW, uchar4* __restrict__ out, int p0, int n) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, j = blockIdx.z;
  if (x >= overlapW) return;
  const int cam = j < n ? p0 + j : (p0 + (j - n) + 1) % P;
  const int x0 = j < n ? camW - overlapW : 0;
  out[((size_t)j * camH + y) * overlapW + x] = proj[((size_t)cam * camH + y) * camW + x0 + x];
}

// remap_cubic_f32c2_at for a sample whose y coordinate is an INTEGER — every sample of renderLazyNovelView's flow remap:
// the lazy buffer's rows are (float)v (TRSP:279-283), so the fraction index of y is 0 and the bicubic weights of rows 0, 2
// and 3 are (+-)0 * cx[q], the weights of row 1 are 1 * cx[q] (cubic_coeffs(0) = {0, 1, 0, 0} exactly; the 2-D table is
// cy[r] * cx[q]). The reference still forms all sixteen products: twelve of them are +-0 (flows are finite), its sum
// o = row0; o += row1; o += row2; o += row3 therefore equals row 1's four-term sum — also when that is a zero, whose sign
// nobody downstream can see (x + 0*t, sqrt(f.x^2 + f.y^2)). One row of taps is loaded instead of four: 2 of the 8 16-byte
// loads and 8 of the 32 multiply-adds per sample. Samples whose taps touch the border, or with a fractional y, take
// the general function.
__device__ __forceinline__ float2 remap_cubic_f32c2_introw_at(const float2* __restrict__ src, int sw, int sh, float mx,
                                                              float my, const float* __restrict__ tab) {
  int sx, sy, fxy;
  remap_coord(mx, my, &sx, &sy, &fxy);
  const unsigned width1 = sw - 3 > 0 ? sw - 3 : 0, height1 = sh - 3 > 0 ? sh - 3 : 0;
  if ((fxy >> 5) == 0 && (unsigned)sx < width1 && (unsigned)sy < height1) {
    typedef float f4a8_ __attribute__((ext_vector_type(4), aligned(8)));
    const f4a8_* V = reinterpret_cast<const f4a8_*>(src + (size_t)(sy + 1) * sw + sx);
    const f4a8_ ab = V[0], cd = V[1];
    const float4 wr = *reinterpret_cast<const float4*>(tab + fxy * 16 + 4);  // row 1 of the 4x4 weights: 1 * cx[q]
    float2 o;
    o.x = ab.x * wr.x + ab.z * wr.y + cd.x * wr.z + cd.z * wr.w;
    o.y = ab.y * wr.x + ab.w * wr.y + cd.y * wr.z + cd.w * wr.w;
    return o;
  }
  return remap_cubic_f32c2_at(src, sw, sh, mx, my, tab);
}

// ------------------------------------------------------------------------------------------
// One lazily rendered novel-view sample (renderLazyNovelView, NovelView.cpp:174-224).
struct LazySample { uchar4 c; float mag; };
__device__ __forceinline__ LazySample lazy_sample(const uchar4* __restrict__ img, const float2* __restrict__ flow,
                                                  int ow, int oh, float xs, float ys, float t,
                                                  const DevTables& T) {
  const float2 f = remap_cubic_f32c2_introw_at(flow, ow, oh, xs, ys, T.bicubic_f);
  const float wx = xs + f.x * t, wy = ys + f.y * t;
  LazySample s;
  s.c = remap_cubic_u8c4_at(img, ow, oh, wx, wy, T.bicubic_i);
  s.c.w = (unsigned char)(int)((1.0f - t) * (float)s.c.w);
  s.mag = sqrtf(f.x * f.x + f.y * f.y);
  return s;
}
// combineLazyViews (NovelView.cpp:101-154)
__device__ __forceinline__ uchar4 combine_lazy(uchar4 cL, uchar4 cR, float flowMagL, float flowMagR, float fcols,
                                               const DevTables& T) {
  const unsigned char mx = cL.w > cR.w ? cL.w : cR.w;
  const unsigned char outAlpha = ((double)((float)mx / 255.0f) > 0.1) ? 255 : 0;
  if (cL.w == 0 && cR.w == 0) return make_uchar4(0, 0, 0, outAlpha);
  if (cL.w == 0) return make_uchar4(cR.x, cR.y, cR.z, outAlpha);
  if (cR.w == 0) return make_uchar4(cL.x, cL.y, cL.z, outAlpha);
  const float magL = flowMagL / fcols, magR = flowMagR / fcols;
  float blendL = (float)cL.w, blendR = (float)cR.w;
  const float norm = blendL + blendR;
  blendL /= norm;
  blendR /= norm;
  const int sdiff = abs((int)cL.x - (int)cR.x) + abs((int)cL.y - (int)cR.y) + abs((int)cL.z - (int)cR.z);
  const float deghostCoef = T.tanh10[sdiff];
  const double expL = exp((double)(10.0f * blendL) * (1.0 + (double)(20.0f * magL)));
  const double expR = exp((double)(10.0f * blendR) * (1.0 + (double)(20.0f * magR)));
  const double sumExp = expL + expR + 0.00001;
  const float softmaxL = (float)(expL / sumExp);
  const float softmaxR = (float)(expR / sumExp);
  const float wL = lerpf(blendL, softmaxL, deghostCoef), wR = lerpf(blendR, softmaxR, deghostCoef);
  return make_uchar4((unsigned char)trunc_u8((float)cL.x * wL + (float)cR.x * wR),
                     (unsigned char)trunc_u8((float)cL.y * wL + (float)cR.y * wR),
                     (unsigned char)trunc_u8((float)cL.z * wL + (float)cR.z * wR), 255);
}
// A workgroup renders a 64 x 4 tile of one eye's strip of one pair (grid: (ceil(stripW/64), ceil(camH/4), 2*(p1-p0)),
// z = 2*pairLocal + eye), the tiles taken in XCD-aware order: the bicubic footprints of vertically adjacent rows overlap
// by three rows, and a tile's neighbours meet in the same L2 (one wave per 64-pixel row segment in launch order fetched
// 3-5 x the images' and flows' bytes through the fabric: profiles/r03_v10_pmc_fetch_write.txt).
constexpr int NV_TW = 64, NV_TH = 4;
__global__ __launch_bounds__(NV_TW* NV_TH) void k_novel_view(const uchar4* __restrict__ overlaps,
                                                             const float2* __restrict__ flows, uchar4* __restrict__ strips,
                                                             NovelViewParams nv, int p0, DevTables T) {
  const TileId tile = xcd_tile();
  const int u = tile.x * NV_TW + (threadIdx.x & (NV_TW - 1)), v = tile.y * NV_TH + (threadIdx.x >> 6);
  if (u >= nv.stripW || v >= nv.camH) return;
  const int j = (int)(tile.z >> 1), pair = p0 + j, eye = tile.z & 1;
  const size_t isz = (size_t)nv.overlapW * nv.camH;
  const uchar4* imgL = overlaps + isz * j;
  const uchar4* imgR = overlaps + isz * (nv.numLocal + j);
  const float2* flowLtoR = flows + isz * j;
  const float2* flowRtoL = flows + isz * (nv.numLocal + j);
  uchar4 out = make_uchar4(0, 0, 0, 0);
  if (u < nv.numNovelViews) {
    // LazyNovelViewBuffer column u (TRSP:273-285)
    const float shift = (float)u / (float)nv.numNovelViews;
    const float slabShift = nv.camImageWidthHalf - (float)(nv.numNovelViews - u);
    const float xs = eye == 0 ? slabShift + nv.disp : slabShift - nv.disp;
    const float ys = (float)v;
    // from-left: (imageL, flowRtoL, t); from-right: (imageR, flowLtoR, 1 - t)   (NovelView.cpp:230-255)
    const LazySample a = lazy_sample(imgL, flowRtoL, nv.overlapW, nv.camH, xs, ys, shift, T);
    const LazySample b = lazy_sample(imgR, flowLtoR, nv.overlapW, nv.camH, xs, ys, 1.0f - shift, T);
    out = combine_lazy(a.c, b.c, a.mag, b.mag, (float)nv.stripW, T);
  } else {
    // unfilled LazyNovelViewBuffer columns are (0,0,0) warps; never reached when eqr_width % numCams == 0
    const LazySample a = lazy_sample(imgL, flowRtoL, nv.overlapW, nv.camH, 0.f, 0.f, 0.f, T);
    const LazySample b = lazy_sample(imgR, flowLtoR, nv.overlapW, nv.camH, 0.f, 0.f, 1.0f, T);
    out = combine_lazy(a.c, b.c, a.mag, b.mag, (float)nv.stripW, T);
  }
  strips[(((size_t)eye * nv.numPairs + pair) * nv.camH + v) * nv.stripW + u] = out;
}

// ------------------------------------------------------------------------------------------
// offsetHorizontalWrap's source column (CvUtil.cpp:93-115): nearest remap with BORDER_WRAP
__device__ __forceinline__ int wrap_src_col(int x, float offset, int W) {
  float srcX = (float)x - offset;
  if (srcX < 0) srcX += (float)W;
  if (srcX >= (float)W) srcX -= (float)W;
  int sx = sat_s16(cv_round(srcX));
  if (sx < 0) sx -= ((sx - W + 1) / W) * W;
  if (sx >= W) sx %= W;
  return sx;
}
// 4 output pixels per thread: 4 source loads (adjacent except at strip seams / the wrap point), one 16-byte store.
__global__ __launch_bounds__(256) void k_assemble_pano(const uchar4* __restrict__ strips, int P, int camH, int stripW,
                                                       float offset, uchar4* __restrict__ pano, int W, int H,
                                                       int padAbove) {
  const int x = (blockIdx.x * blockDim.x + threadIdx.x) * 4, y = blockIdx.y;
  if (x >= W) return;
  const int ys = y - padAbove;
  const bool rowIn = ys >= 0 && ys < camH;
  uchar4 o[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = make_uchar4(0, 0, 0, 0);
  const int nvalid = min(4, W - x);
  if (rowIn) {
    int sx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) sx[k] = wrap_src_col(min(x + k, W - 1), offset, W);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int pk = sx[k] / stripW, uk = sx[k] - pk * stripW;
      o[k] = strips[((size_t)pk * camH + ys) * stripW + uk];
    }
  }
  uchar4* D = pano + (size_t)y * W + x;
  if (nvalid == 4 && (W & 3) == 0) {
    *reinterpret_cast<uint4*>(D) = make_uint4(__builtin_bit_cast(unsigned, o[0]), __builtin_bit_cast(unsigned, o[1]),
                                              __builtin_bit_cast(unsigned, o[2]), __builtin_bit_cast(unsigned, o[3]));
  } else {
    for (int k = 0; k < nvalid; ++k) D[k] = o[k];
  }
}
__global__ __launch_bounds__(256) void k_flip_both(const uchar4* __restrict__ src, uchar4* __restrict__ dst, int w,
                                                   int h) {  // grid.y = number of leading rows of the flipped image
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  dst[(size_t)y * w + x] = src[(size_t)(h - 1 - y) * w + (w - 1 - x)];
}

// ---- featherAlphaChannel (CvUtil.cpp:140-157) ---------------------------------------------------
// erode with MORPH_CROSS (2e+1)^2 of the alpha channel (outside treated as +inf = 255), then an 8-bit Gaussian.
// Both are LDS-tiled: the erosion runs the two arms of the cross as log-step (doubling) window minima, the Gaussian
// does its row and column passes inside one tile.
constexpr int ER_TW = 128, ER_TH = 32, ER_MAXE = 32;
// Window minimum of length len = 2e+1 by van Herk / Gil-Werman: per segment of len elements a running prefix
// minimum g and a running suffix minimum hh; min over [i, i+len) = min(hh[i], g[i+len-1]). ~3 operations per
// element whatever the window length. Words (not bytes) in LDS: no sub-dword traffic.
__global__ __launch_bounds__(256) void k_erode_cross_tiled(const uchar4* __restrict__ img, uint8_t* __restrict__ out,
                                                           int w, int h, int e) {
  constexpr int HWMAX = ER_TW + 2 * ER_MAXE, VHMAX = ER_TH + 2 * ER_MAXE;
  __shared__ unsigned char s_src_h[ER_TH][HWMAX + 4];
  __shared__ unsigned char s_g_h[ER_TH][HWMAX + 4];
  __shared__ unsigned char s_s_h[ER_TH][HWMAX + 4];
  __shared__ unsigned char s_src_v[VHMAX][ER_TW];
  __shared__ unsigned char s_g_v[VHMAX][ER_TW];
  __shared__ unsigned char s_s_v[VHMAX][ER_TW];
  const int tx = threadIdx.x & (ER_TW - 1), ty = threadIdx.x >> 7;  // 128 x 2 threads
  const int x0 = blockIdx.x * ER_TW, y0 = blockIdx.y * ER_TH;
  const int len = 2 * e + 1;
  const int HW = ER_TW + 2 * e, VH = ER_TH + 2 * e;
  for (int ly = ty; ly < ER_TH; ly += 2)
    for (int lx = tx; lx < HW; lx += ER_TW) {
      const int gx = x0 - e + lx, gy = y0 + ly;
      s_src_h[ly][lx] = (gx >= 0 && gx < w && gy < h) ? img[(size_t)gy * w + gx].w : 255;
    }
  for (int ly = ty; ly < VH; ly += 2) {
    const int gx = x0 + tx, gy = y0 - e + ly;
    s_src_v[ly][tx] = (gx < w && gy >= 0 && gy < h) ? img[(size_t)gy * w + gx].w : 255;
  }
  __syncthreads();
  // horizontal arm: task = (row, segment); segments of `len` elements tile the row [0, HW)
  const int nsegH = (HW + len - 1) / len;
  for (int t = threadIdx.x; t < ER_TH * nsegH; t += 256) {
    const int ly = t % ER_TH, seg = t / ER_TH;
    const int b0 = seg * len, b1 = min(b0 + len, HW);
    unsigned char m = 255;
    for (int i = b0; i < b1; ++i) { m = min(m, s_src_h[ly][i]); s_g_h[ly][i] = m; }
    m = 255;
    for (int i = b1 - 1; i >= b0; --i) { m = min(m, s_src_h[ly][i]); s_s_h[ly][i] = m; }
  }
  // vertical arm: task = (column, segment)
  const int nsegV = (VH + len - 1) / len;
  for (int t = threadIdx.x; t < ER_TW * nsegV; t += 256) {
    const int lx = t & (ER_TW - 1), seg = t >> 7;
    const int b0 = seg * len, b1 = min(b0 + len, VH);
    unsigned char m = 255;
    for (int i = b0; i < b1; ++i) { m = min(m, s_src_v[i][lx]); s_g_v[i][lx] = m; }
    m = 255;
    for (int i = b1 - 1; i >= b0; --i) { m = min(m, s_src_v[i][lx]); s_s_v[i][lx] = m; }
  }
  __syncthreads();
  // output (x, y): horizontal window [lx, lx + len) of the haloed row, vertical window [ly, ly + len) of the column
  const int gx = x0 + tx;
  if (gx < w)
    for (int ly = ty; ly < ER_TH; ly += 2) {
      const int gy = y0 + ly;
      if (gy >= h) continue;
      const unsigned char mh = min(s_s_h[ly][tx], s_g_h[ly][tx + len - 1]);
      const unsigned char mv = min(s_s_v[ly][tx], s_g_v[ly + len - 1][tx]);
      out[(size_t)gy * w + gx] = min(mh, mv);
    }
}

// The same erosion for a compile-time radius E, all window minima in registers. Window minimum of length 2E+1 by
// doubling: m2[i] = min(a[i], a[i+1]), m4[i] = min(m2[i], m2[i+2]) ... up to the largest power of two P <= 2E+1, then
// min(mP[i], mP[i + 2E+1 - P]). A thread owns a strip of 32 outputs (32 + 2E inputs) of TWO rows (horizontal arm) or
// TWO columns (vertical arm) at once, the two packed as the 16-bit halves of a register (v_pk_min_u16): ~7 VALU
// operations per pixel and arm, no serial dependence on the window length, no LDS traffic beyond one read of the tile.
constexpr int ERF_TW = 128, ERF_TH = 64, ERF_OUT = 32;
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
template <int N, int LEN>
__device__ __forceinline__ void window_min_strip(us2 (&a)[N]) {  // a[i] <- min(a[i .. i+LEN-1]) for i < N - LEN + 1
  int span = 1;
#pragma unroll
  for (int d = 1; 2 * d <= LEN; d *= 2) {
#pragma unroll
    for (int i = 0; i + d < N; ++i) a[i] = __builtin_elementwise_min(a[i], a[i + d]);
    span = 2 * d;
  }
  if (span < LEN) {
#pragma unroll
    for (int i = 0; i + LEN - span < N; ++i) a[i] = __builtin_elementwise_min(a[i], a[i + LEN - span]);
  }
}
template <int E>
__global__ __launch_bounds__(256) void k_erode_cross_fixed(const uchar4* __restrict__ img, uint8_t* __restrict__ out,
                                                           int w, int h) {
  constexpr int LEN = 2 * E + 1, NIN = ERF_OUT + 2 * E, HW = ERF_TW + 2 * E, VH = ERF_TH + 2 * E;
  constexpr int HWP = (HW + 3) & ~3;
  __shared__ unsigned s_h[ERF_TH / 2][HWP];       // row pairs: alpha(2j, c) | alpha(2j+1, c) << 16
  __shared__ unsigned char s_v[VH][ERF_TW];       // alpha with the vertical halo
  __shared__ unsigned char s_o[ERF_TH][ERF_TW];   // horizontal-arm result
  const int x0 = blockIdx.x * ERF_TW, y0 = blockIdx.y * ERF_TH;
  const int tid = threadIdx.x;
  const unsigned ref = img[(size_t)y0 * w + x0].w;  // uniform tiles (alpha 0 or 255 almost everywhere) leave after the load
  bool same = true;
  if ((w & 3) == 0 && (E & 3) == 3) {
    // Four pixels per load (16 bytes; x0 and w are multiples of 4, so a group is entirely inside or outside the image).
    // The horizontal tile starts one column early (x0 - E - 1) to keep the groups aligned: s_h index = lx + 1 below.
    const uint4* __restrict__ img4 = reinterpret_cast<const uint4*>(img);
    constexpr int GH = (HW + 1 + 3) / 4;  // groups per row pair: columns [x0 - E - 1, x0 - E - 1 + 4 GH)
    static_assert(4 * GH <= HWP + 4, "s_h row too short");
    // Two phases per tile part: every group this thread brings in is REQUESTED first, then all of them go to LDS. As one
    // loop (load, use, store, next) the 7 + 16 iterations were as many serialised memory round trips per tile — and
    // most tiles are constant and leave right after the load (the same change the flow stencils got in round 3).
    constexpr int N1 = (ERF_TH / 2) * GH, IT1 = (N1 + 255) / 256;
    uint4 q0[IT1], q1[IT1];
#pragma unroll
    for (int it = 0; it < IT1; ++it) {
      const int t = tid + 256 * it;
      const int j = min(t, N1 - 1) / GH, g = min(t, N1 - 1) - j * GH;
      const int gx = x0 - E - 1 + 4 * g, gy = y0 + 2 * j;
      q0[it] = make_uint4(~0u, ~0u, ~0u, ~0u);
      q1[it] = q0[it];
      if (t < N1 && gx >= 0 && gx < w) {
        if (gy < h) q0[it] = img4[((size_t)gy * w + gx) >> 2];
        if (gy + 1 < h) q1[it] = img4[((size_t)(gy + 1) * w + gx) >> 2];
      }
    }
    constexpr int N2 = VH * (ERF_TW / 4), IT2 = (N2 + 255) / 256;
    uint4 qv[IT2];
#pragma unroll
    for (int it = 0; it < IT2; ++it) {
      const int t = tid + 256 * it;
      const int ly = min(t, N2 - 1) / (ERF_TW / 4), g = min(t, N2 - 1) - ly * (ERF_TW / 4);
      const int gx = x0 + 4 * g, gy = y0 - E + ly;
      qv[it] = make_uint4(~0u, ~0u, ~0u, ~0u);
      if (t < N2 && gx < w && gy >= 0 && gy < h) qv[it] = img4[((size_t)gy * w + gx) >> 2];
    }
#pragma unroll
    for (int it = 0; it < IT1; ++it) {
      const int t = tid + 256 * it;
      if (t >= N1) continue;
      const int j = t / GH, g = t - j * GH;
      const uint4 p0 = q0[it], p1 = q1[it];
      const unsigned a0[4] = {p0.x >> 24, p0.y >> 24, p0.z >> 24, p0.w >> 24};
      const unsigned a1[4] = {p1.x >> 24, p1.y >> 24, p1.z >> 24, p1.w >> 24};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int lx = 4 * g + k - 1;  // index in the tile that starts at x0 - E
        if (lx >= 0 && lx < HW) {
          same = same && a0[k] == ref && a1[k] == ref;
          s_h[j][lx] = a0[k] | (a1[k] << 16);
        }
      }
    }
#pragma unroll
    for (int it = 0; it < IT2; ++it) {
      const int t = tid + 256 * it;
      if (t >= N2) continue;
      const int ly = t / (ERF_TW / 4), g = t - ly * (ERF_TW / 4);
      const uint4 p = qv[it];
      const unsigned v = (p.x >> 24) | ((p.y >> 24) << 8) | ((p.z >> 24) << 16) | (p.w & 0xff000000u);
      same = same && v == ref * 0x01010101u;
      *reinterpret_cast<unsigned*>(&s_v[ly][4 * g]) = v;
    }
  } else {
    for (int t = tid; t < (ERF_TH / 2) * HW; t += 256) {
      const int j = t / HW, lx = t - j * HW;
      const int gx = x0 - E + lx, gy = y0 + 2 * j;
      const bool cx = gx >= 0 && gx < w;
      const unsigned a0 = (cx && gy < h) ? img[(size_t)gy * w + gx].w : 255u;
      const unsigned a1 = (cx && gy + 1 < h) ? img[(size_t)(gy + 1) * w + gx].w : 255u;
      same = same && a0 == ref && a1 == ref;
      s_h[j][lx] = a0 | (a1 << 16);
    }
    for (int t = tid; t < VH * ERF_TW; t += 256) {
      const int ly = t >> 7, lx = t & (ERF_TW - 1);
      const int gx = x0 + lx, gy = y0 - E + ly;
      const unsigned v = (gx < w && gy >= 0 && gy < h) ? img[(size_t)gy * w + gx].w : 255u;
      same = same && v == ref;
      s_v[ly][lx] = (unsigned char)v;
    }
  }
  if (__syncthreads_and(same)) {  // the minimum over any window of a constant tile (out-of-image = 255 included) is the constant
    const int ly = tid >> 2, gy = y0 + ly, gx0 = x0 + (tid & 3) * 32;
    if (gy < h)
      for (int i = 0; i < 32; i += 2) {
        const int gx = gx0 + i;
        if (gx + 1 < w) *reinterpret_cast<unsigned short*>(out + (size_t)gy * w + gx) = (unsigned short)(ref | (ref << 8));
        else if (gx < w) out[(size_t)gy * w + gx] = (uint8_t)ref;
      }
    return;
  }
  us2 a[NIN];
  if (tid < 128) {  // horizontal arm: row pair j, strip k of 32 columns
    const int j = tid >> 2, k = tid & 3;
    const unsigned* row = &s_h[j][k * ERF_OUT];
#pragma unroll
    for (int i = 0; i < NIN; ++i) a[i] = __builtin_bit_cast(us2, row[i]);
    window_min_strip<NIN, LEN>(a);
#pragma unroll
    for (int i = 0; i < ERF_OUT; ++i) {
      s_o[2 * j][k * ERF_OUT + i] = (unsigned char)a[i].x;
      s_o[2 * j + 1][k * ERF_OUT + i] = (unsigned char)a[i].y;
    }
  } else {  // vertical arm: column pair c, strip k of 32 rows
    const int t = tid - 128, c = t & 63, k = t >> 6;
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      const unsigned short two = *reinterpret_cast<const unsigned short*>(&s_v[k * ERF_OUT + i][2 * c]);
      a[i] = __builtin_bit_cast(us2, __builtin_amdgcn_perm(0u, (unsigned)two, 0x0c010c00u));
    }
    window_min_strip<NIN, LEN>(a);
  }
  __syncthreads();
  if (tid >= 128) {
    const int t = tid - 128, c = t & 63, k = t >> 6;
    const int gx = x0 + 2 * c;
#pragma unroll
    for (int i = 0; i < ERF_OUT; ++i) {
      const int ly = k * ERF_OUT + i, gy = y0 + ly;
      const unsigned short hh = *reinterpret_cast<const unsigned short*>(&s_o[ly][2 * c]);
      const unsigned m0 = min((unsigned)a[i].x, (unsigned)(hh & 0xff)), m1 = min((unsigned)a[i].y, (unsigned)(hh >> 8));
      if (gy < h) {
        if (gx + 1 < w) *reinterpret_cast<unsigned short*>(out + (size_t)gy * w + gx) = (unsigned short)(m0 | (m1 << 8));
        else if (gx < w) out[(size_t)gy * w + gx] = (uint8_t)m0;
      }
    }
  }
}

// GaussianBlur on CV_8U (ksize x ksize, fixed-point taps ik scaled by 256, BORDER_REFLECT_101): row pass to int32
// (exact), column pass — both inside one LDS tile, 4 outputs per thread along the filter axis. The column pass of the
// reference's x86-64 build is SymmColumnVec_32s8u (SSE2) on whole groups of 4 columns: float taps ik*2^-16,
// float(centre)*k0 + 0, += float(pair sum)*kj outwards, cvtps2dq (round-half-even), saturate; the last w % 4 columns
// take the scalar FixedPtCastEx, (sum + 2^15) >> 16. When sum(ik) <= 256 every float partial sum is an exact multiple
// of 2^-16 below 2^8 (24 bits), so the SSE2 result is the integer sum rounded half-even; otherwise the float
// accumulation is replayed literally.
__device__ __forceinline__ int gauss_u8_round(int acc, bool vec) {
  if (!vec) return sat_u8((acc + (1 << 15)) >> 16);
  const int q = acc >> 16, rem = acc & 0xFFFF;
  return sat_u8(q + ((rem > 0x8000 || (rem == 0x8000 && (q & 1))) ? 1 : 0));
}
// literal SymmColumnVec_32s8u accumulation for one output: rows[c - r .. c + r] of the int32 row pass
template <typename RowAt, typename TapAt>
__device__ __forceinline__ int gauss_u8_float_column(RowAt rowAt, TapAt tapAt, int r) {
  const float sc = (float)(1. / 65536);
  float s = (float)rowAt(0) * ((float)tapAt(0) * sc) + 0.0f;
  for (int j = 1; j <= r; ++j) s = s + (float)(rowAt(j) + rowAt(-j)) * ((float)tapAt(j) * sc);
  return sat_u8(cv_round(s));
}
constexpr int GU_TW = 64, GU_TH = 32, GU_MAXR = 16;
__global__ __launch_bounds__(256) void k_gauss_u8_tiled(const uint8_t* __restrict__ a, uint8_t* __restrict__ out, int w,
                                                        int h, const int* __restrict__ ik, int r) {
  constexpr int IWMAX = GU_TW + 2 * GU_MAXR, IHMAX = GU_TH + 2 * GU_MAXR;
  __shared__ int s_a[IHMAX][IWMAX + 1];
  __shared__ int s_row[IHMAX][GU_TW + 1];
  __shared__ int s_k[2 * GU_MAXR + 1];
  const int tx = threadIdx.x & (GU_TW - 1), ty = threadIdx.x >> 6;  // 64 x 4 threads
  const int x0 = blockIdx.x * GU_TW, y0 = blockIdx.y * GU_TH;
  const int IW = GU_TW + 2 * r, IH = GU_TH + 2 * r;
  if ((int)threadIdx.x <= 2 * r) s_k[threadIdx.x] = ik[threadIdx.x];
  int ksum = 0;
  for (int j = 0; j <= 2 * r; ++j) ksum += ik[j];
  const bool exact = ksum <= 256;
  for (int ly = ty; ly < IH; ly += 4) {
    const uint8_t* row = a + (size_t)reflect101(y0 - r + ly, h) * w;
    for (int lx = tx; lx < IW; lx += GU_TW) s_a[ly][lx] = row[reflect101(x0 - r + lx, w)];
  }
  __syncthreads();
  // row pass: task = (row, group of 4 consecutive x); sums are integers: any order gives the same result
  for (int t = threadIdx.x; t < IH * (GU_TW / 4); t += 256) {
    const int ly = t % IH, lx0 = (t / IH) * 4;
    int acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    for (int j = 0; j < 2 * r + 4; ++j) {
      const int v = s_a[ly][lx0 + j];
      // tap index of input j for output o is j - o (valid 0..2r)
      if (j <= 2 * r) acc0 += (int)__umul24((unsigned)s_k[j], (unsigned)v);
      if (j >= 1 && j - 1 <= 2 * r) acc1 += (int)__umul24((unsigned)s_k[j - 1], (unsigned)v);
      if (j >= 2 && j - 2 <= 2 * r) acc2 += (int)__umul24((unsigned)s_k[j - 2], (unsigned)v);
      if (j >= 3) acc3 += (int)__umul24((unsigned)s_k[j - 3], (unsigned)v);
    }
    s_row[ly][lx0] = acc0; s_row[ly][lx0 + 1] = acc1; s_row[ly][lx0 + 2] = acc2; s_row[ly][lx0 + 3] = acc3;
  }
  __syncthreads();
  // column pass: task = (column, group of 4 consecutive y)
  for (int t = threadIdx.x; t < GU_TW * (GU_TH / 4); t += 256) {
    const int lx = t & (GU_TW - 1), ly0 = (t >> 6) * 4;
    int acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    for (int j = 0; j < 2 * r + 4; ++j) {
      const int v = s_row[ly0 + j][lx];
      if (j <= 2 * r) acc0 += (int)__umul24((unsigned)s_k[j], (unsigned)v);
      if (j >= 1 && j - 1 <= 2 * r) acc1 += (int)__umul24((unsigned)s_k[j - 1], (unsigned)v);
      if (j >= 2 && j - 2 <= 2 * r) acc2 += (int)__umul24((unsigned)s_k[j - 2], (unsigned)v);
      if (j >= 3) acc3 += (int)__umul24((unsigned)s_k[j - 3], (unsigned)v);
    }
    const int gx = x0 + lx;
    if (gx >= w) continue;
    const int acc[4] = {acc0, acc1, acc2, acc3};
    const bool vec = gx < (w & ~3);
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const int gy = y0 + ly0 + o;
      if (gy >= h) continue;
      int v;
      if (exact || !vec) v = gauss_u8_round(acc[o], vec);
      else v = gauss_u8_float_column([&](int j) { return s_row[ly0 + o + r + j][lx]; }, [&](int j) { return s_k[r + j]; }, r);
      out[(size_t)gy * w + gx] = (uint8_t)v;
    }
  }
}
// The same filter for a compile-time radius R: every loop unrolled, tap index known at compile time.
template <int R>
__global__ __launch_bounds__(256) void k_gauss_u8_fixed(const uint8_t* __restrict__ a, uint8_t* __restrict__ out, int w,
                                                        int h, const int* __restrict__ ik) {
  constexpr int IW = GU_TW + 2 * R, IH = GU_TH + 2 * R, NT = 2 * R + 1;
  __shared__ int s_a[IH][IW + 1];
  __shared__ int s_row[IH][GU_TW + 1];
  const int tx = threadIdx.x & (GU_TW - 1), ty = threadIdx.x >> 6;  // 64 x 4 threads
  const int x0 = blockIdx.x * GU_TW, y0 = blockIdx.y * GU_TH;
  int k[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) k[j] = ik[j];  // uniform: scalar registers
  int ksum = 0;
#pragma unroll
  for (int j = 0; j < NT; ++j) ksum += k[j];
  const bool exact = ksum <= 256;
  // Constant tiles (alpha 0 or 255 almost everywhere) leave after the load: every window sums to ksum^2 * v, which both
  // roundings of the column pass return as v when the taps sum to 256, and as 0 for v = 0 whatever the taps.
  const int ref = a[(size_t)y0 * w + x0];
  bool same = ref == 0 || ksum == 256;
  if ((w & 3) == 0 && (R & 3) == 3) {
    // four bytes per load: groups start at x0 - R - 1 (a multiple of 4) and lie entirely inside or outside the row
    constexpr int GW = (IW + 1 + 3) / 4;
    // (requested together, then stored: see k_erode_cross_fixed)
    constexpr int NG = IH * GW, ITG = (NG + 255) / 256;
    unsigned q[ITG];
#pragma unroll
    for (int it = 0; it < ITG; ++it) {
      const int t = threadIdx.x + 256 * it;
      const int ly = min(t, NG - 1) / GW, g = min(t, NG - 1) - ly * GW;
      const uint8_t* row = a + (size_t)reflect101(y0 - R + ly, h) * w;
      const int gx = x0 - R - 1 + 4 * g;
      if (gx >= 0 && gx < w) {
        q[it] = *reinterpret_cast<const unsigned*>(row + gx);
      } else {
        unsigned v4 = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) v4 |= (unsigned)row[reflect101(gx + k, w)] << (8 * k);
        q[it] = v4;
      }
    }
#pragma unroll
    for (int it = 0; it < ITG; ++it) {
      const int t = threadIdx.x + 256 * it;
      if (t >= NG) continue;
      const int ly = t / GW, g = t - ly * GW;
      const unsigned v4 = q[it];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int lx = 4 * g + k - 1, v = (int)((v4 >> (8 * k)) & 0xffu);
        if (lx >= 0 && lx < IW) {
          same = same && v == ref;
          s_a[ly][lx] = v;
        }
      }
    }
  } else {
    for (int ly = ty; ly < IH; ly += 4) {
      const uint8_t* row = a + (size_t)reflect101(y0 - R + ly, h) * w;
      for (int lx = tx; lx < IW; lx += GU_TW) {
        const int v = row[reflect101(x0 - R + lx, w)];
        same = same && v == ref;
        s_a[ly][lx] = v;
      }
    }
  }
  if (__syncthreads_and(same)) {
    const int ly = threadIdx.x >> 3, gy = y0 + ly, gx0 = x0 + (threadIdx.x & 7) * 8;
    if (gy < h)
      for (int i = 0; i < 8; ++i)
        if (gx0 + i < w) out[(size_t)gy * w + gx0 + i] = (uint8_t)ref;
    return;
  }
  // row pass: task = (row, group of 4 consecutive x)
  for (int t = threadIdx.x; t < IH * (GU_TW / 4); t += 256) {
    const int g = t / IH, ly = t - g * IH, lx0 = g * 4;
    int acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
#pragma unroll
    for (int j = 0; j < NT + 3; ++j) {
      const int v = s_a[ly][lx0 + j];
      if (j < NT) acc0 += (int)__umul24((unsigned)k[j], (unsigned)v);
      if (j >= 1 && j - 1 < NT) acc1 += (int)__umul24((unsigned)k[j - 1], (unsigned)v);
      if (j >= 2 && j - 2 < NT) acc2 += (int)__umul24((unsigned)k[j - 2], (unsigned)v);
      if (j >= 3) acc3 += (int)__umul24((unsigned)k[j - 3], (unsigned)v);
    }
    s_row[ly][lx0] = acc0; s_row[ly][lx0 + 1] = acc1; s_row[ly][lx0 + 2] = acc2; s_row[ly][lx0 + 3] = acc3;
  }
  __syncthreads();
  // column pass: task = (column, group of 4 consecutive y)
  for (int t = threadIdx.x; t < GU_TW * (GU_TH / 4); t += 256) {
    const int lx = t & (GU_TW - 1), ly0 = (t >> 6) * 4;
    int acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < NT + 3; ++j) {
      const int v = s_row[ly0 + j][lx];
      if (j < NT) acc[0] += (int)__umul24((unsigned)k[j], (unsigned)v);
      if (j >= 1 && j - 1 < NT) acc[1] += (int)__umul24((unsigned)k[j - 1], (unsigned)v);
      if (j >= 2 && j - 2 < NT) acc[2] += (int)__umul24((unsigned)k[j - 2], (unsigned)v);
      if (j >= 3) acc[3] += (int)__umul24((unsigned)k[j - 3], (unsigned)v);
    }
    const int gx = x0 + lx;
    if (gx >= w) continue;
    const bool vec = gx < (w & ~3);
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const int gy = y0 + ly0 + o;
      if (gy >= h) continue;
      int v;
      if (exact || !vec) v = gauss_u8_round(acc[o], vec);
      else v = gauss_u8_float_column([&](int j) { return s_row[ly0 + o + R + j][lx]; }, [&](int j) { return ik[R + j]; }, R);
      out[(size_t)gy * w + gx] = (uint8_t)v;
    }
  }
}
__global__ __launch_bounds__(256) void k_extend_wrap(const uchar4* __restrict__ img, const uint8_t* __restrict__ alpha,
                                                     int cols, int rows, uchar4* __restrict__ ext, int extW) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= extW) return;
  const int sx = x % cols;
  uchar4 p = img[(size_t)y * cols + sx];
  if (alpha) p.w = alpha[(size_t)y * cols + sx];
  ext[(size_t)y * extW + x] = p;
}

// ---- pole warp / finish ------------------------------------------------------------------------
// the same, four pixels per thread (cols and extW multiples of 4: a group never straddles the wrap point)
__global__ __launch_bounds__(256) void k_extend_wrap_v4(const uint4* __restrict__ img, const unsigned* __restrict__ alpha,
                                                        int cols4, int rows, uint4* __restrict__ ext, int extW4) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= extW4) return;
  const int sx = x % cols4;
  uint4 p = img[(size_t)y * cols4 + sx];
  if (alpha) {
    const unsigned a = alpha[(size_t)y * cols4 + sx];
    p.x = (p.x & 0xffffffu) | (a << 24);
    p.y = (p.y & 0xffffffu) | ((a >> 8) << 24);
    p.z = (p.z & 0xffffffu) | ((a >> 16) << 24);
    p.w = (p.w & 0xffffffu) | (a & 0xff000000u);
  }
  ext[(size_t)y * extW4 + x] = p;
}
// one pixel of poleToSideFlowThread's seam blend + alpha ramp (TRSP:505-536): o = warped pixel, wr = the pixel `cols`
// further right in the extended image (read only where x < maxBlendX)
__device__ __forceinline__ uchar4 pole_finish_px(uchar4 o, uchar4 wr, int x, int y, const PoleWarpParams& pw) {
  if (x < pw.maxBlendX) {
    const float alpha = 1.0f - rampf((float)x, (float)pw.maxBlendX * 0.333f, (float)pw.maxBlendX * 0.667f);
    const float srcB = o.x, srcG = o.y, srcR = o.z, srcA = o.w;
    o.x = (unsigned char)trunc_u8((float)wr.x * alpha + srcB * (1.0f - alpha));
    o.y = (unsigned char)trunc_u8((float)wr.y * alpha + srcG * (1.0f - alpha));
    o.z = (unsigned char)trunc_u8((float)wr.z * alpha + srcR * (1.0f - alpha));
    o.w = (unsigned char)trunc_u8(srcA);
  }
  const float phi = pw.poleCameraRadius * (float)(y + 0.5f) / (float)pw.rows;
  const float a2 = 1.0f - rampf(phi, pw.phiMid, pw.phiRampEnd);
  o.w = (unsigned char)trunc_u8((float)o.w * a2);
  return o;
}
__global__ __launch_bounds__(256) void k_pole_finish(const uchar4* __restrict__ warpedExt, uchar4* __restrict__ out,
                                                     int eqrH, PoleWarpParams pw) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= pw.cols) return;
  uchar4 o = make_uchar4(0, 0, 0, 0);
  if (y < pw.rows) {
    o = warpedExt[(size_t)y * pw.extW + x];
    uchar4 wr = o;
    if (x < pw.maxBlendX) wr = warpedExt[(size_t)y * pw.extW + min(x + pw.cols, pw.extW - 1)];
    o = pole_finish_px(o, wr, x, y, pw);
  }
  out[(size_t)y * pw.cols + x] = o;
}
// four pixels per thread (cols and extW multiples of 4, extW >= cols + maxBlendX: the seam partner of a group is a group)
__global__ __launch_bounds__(256) void k_pole_finish_v4(const uint4* __restrict__ warpedExt, uint4* __restrict__ out, int eqrH,
                                                        PoleWarpParams pw) {
  const int x4 = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, x = 4 * x4;
  if (x >= pw.cols) return;
  uint4 o = make_uint4(0u, 0u, 0u, 0u);
  if (y < pw.rows) {
    const size_t row = (size_t)y * (pw.extW / 4);
    o = warpedExt[row + x4];
    uint4 wr = o;
    if (x < pw.maxBlendX) wr = warpedExt[row + x4 + pw.cols / 4];
    o.x = __builtin_bit_cast(unsigned, pole_finish_px(__builtin_bit_cast(uchar4, o.x), __builtin_bit_cast(uchar4, wr.x), x, y, pw));
    o.y = __builtin_bit_cast(unsigned, pole_finish_px(__builtin_bit_cast(uchar4, o.y), __builtin_bit_cast(uchar4, wr.y), x + 1, y, pw));
    o.z = __builtin_bit_cast(unsigned, pole_finish_px(__builtin_bit_cast(uchar4, o.z), __builtin_bit_cast(uchar4, wr.z), x + 2, y, pw));
    o.w = __builtin_bit_cast(unsigned, pole_finish_px(__builtin_bit_cast(uchar4, o.w), __builtin_bit_cast(uchar4, wr.w), x + 3, y, pw));
  }
  out[(size_t)y * (pw.cols / 4) + x4] = o;
}

// flattenLayersDeghostPreferBase (CvUtil.cpp:224-260)
__device__ __forceinline__ uchar4 flatten_px(uchar4 b, uchar4 t, const DevTables& T) {
  const int sdiff = abs((int)b.x - (int)t.x) + abs((int)b.y - (int)t.y) + abs((int)b.z - (int)t.z);
  const float deghostCoef = T.tanh5[sdiff];
  const float alphaR = (float)t.w / 255.0f;
  const float alphaL = 1.0f - alphaR;
  const float softmaxL = T.flat_softmaxL[t.w];
  const float softmaxR = 1.0f - softmaxL;
  const float wL = lerpf(alphaL, softmaxL, deghostCoef), wR = lerpf(alphaR, softmaxR, deghostCoef);
  uchar4 o;
  o.x = (unsigned char)trunc_u8((float)b.x * wL + (float)t.x * wR);
  o.y = (unsigned char)trunc_u8((float)b.y * wL + (float)t.y * wR);
  o.z = (unsigned char)trunc_u8((float)b.z * wL + (float)t.z * wR);
  o.w = t.w > b.w ? t.w : b.w;
  return o;
}
__global__ __launch_bounds__(256) void k_flatten(const uchar4* __restrict__ base, const uchar4* __restrict__ top,
                                                 uchar4* __restrict__ out, int w, int h, int flip_top, DevTables T) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  const uchar4 b = base[(size_t)y * w + x];
  const uchar4 t = flip_top ? top[(size_t)(h - 1 - y) * w + (w - 1 - x)] : top[(size_t)y * w + x];
  out[(size_t)y * w + x] = flatten_px(b, t, T);
}
// w % 4 == 0: 4 pixels per thread, 16-byte loads and stores (the flipped top layer is read mirrored)
__global__ __launch_bounds__(256) void k_flatten_v4(const uint4* __restrict__ base, const uint4* __restrict__ top,
                                                    uint4* __restrict__ out, int w4, int h, int flip_top, DevTables T) {
  const int x4 = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x4 >= w4) return;
  const uint4 b = base[(size_t)y * w4 + x4];
  uint4 t;
  if (flip_top) {
    const uint4 r = top[(size_t)(h - 1 - y) * w4 + (w4 - 1 - x4)];
    t = make_uint4(r.w, r.z, r.y, r.x);
  } else {
    t = top[(size_t)y * w4 + x4];
  }
  uint4 o;
  o.x = __builtin_bit_cast(unsigned, flatten_px(__builtin_bit_cast(uchar4, b.x), __builtin_bit_cast(uchar4, t.x), T));
  o.y = __builtin_bit_cast(unsigned, flatten_px(__builtin_bit_cast(uchar4, b.y), __builtin_bit_cast(uchar4, t.y), T));
  o.z = __builtin_bit_cast(unsigned, flatten_px(__builtin_bit_cast(uchar4, b.z), __builtin_bit_cast(uchar4, t.z), T));
  o.w = __builtin_bit_cast(unsigned, flatten_px(__builtin_bit_cast(uchar4, b.w), __builtin_bit_cast(uchar4, t.w), T));
  out[(size_t)y * w4 + x4] = o;
}

// ---- cubemap output (ImageWarper.cpp:95-141 + CvUtil.cpp:117-138) ------------------------------------------------
// One thread per pixel of the stacked stereo cubemap: picks (eye, face, i, j) from the output position (faces are
// flipped horizontally in the "video" layout), reads the cached face warp map and does remap INTER_CUBIC /
// BORDER_WRAP from the eye panorama (taps wrap in x and in y). Output is packed BGR.
__device__ __forceinline__ int border_wrap(int p, int len) {
  if (p < 0) p -= ((p - len + 1) / len) * len;
  if (p >= len) p %= len;
  return p;
}
__global__ __launch_bounds__(256) void k_cubemap(const uchar4* __restrict__ eyeL, const uchar4* __restrict__ eyeR, int sw,
                                                 int sh, const float2* __restrict__ maps /*[6][fh][fw]*/, int fw, int fh,
                                                 int video, uint8_t* __restrict__ out, int ow, int oh,
                                                 const short* __restrict__ tab) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= ow) return;
  const int eyeH = oh / 2;
  const int eye = y >= eyeH, ye = y - eye * eyeH;
  int face, i, j;
  if (video) {  // rows of three faces: {LEFT, RIGHT, TOP}, {BOTTOM, BACK, FRONT} = list indices {1,0,2},{3,4,5}; flipped
    const int col = x / fw, row = ye / fh;
    const int order[6] = {1, 0, 2, 3, 4, 5};
    face = order[row * 3 + col];
    i = fw - 1 - (x - col * fw);
    j = ye - row * fh;
  } else {  // "photo": the six faces stacked vertically in list order
    face = ye / fh;
    i = x;
    j = ye - face * fh;
  }
  const float2 m = maps[((size_t)face * fh + j) * fw + i];
  int sx, sy, fxy;
  remap_coord(m.x, m.y, &sx, &sy, &fxy);
  const short* w = tab + fxy * 16;
  const uchar4* src = eye ? eyeR : eyeL;
  int xs[4], ys[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { xs[q] = border_wrap(sx + q, sw); ys[q] = border_wrap(sy + q, sh); }
  int s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uchar4* S = src + (size_t)ys[r] * sw;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uchar4 p = S[xs[q]];
      const int ww = w[r * 4 + q];
      s0 += p.x * ww; s1 += p.y * ww; s2 += p.z * ww;
    }
  }
  uint8_t* o = out + ((size_t)y * ow + x) * 3;
  o[0] = (uint8_t)sat_u8((s0 + (1 << 14)) >> 15);
  o[1] = (uint8_t)sat_u8((s1 + (1 << 14)) >> 15);
  o[2] = (uint8_t)sat_u8((s2 + (1 << 14)) >> 15);
}

// four BGRA pixels (16 bytes) in, twelve B,G,R bytes (three dwords) out per thread
__global__ __launch_bounds__(256) void k_pack_bgr_v4(const uint4* __restrict__ src, size_t n4, unsigned* __restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const uint4 p = src[i];
  dst[i * 3] = (p.x & 0xffffffu) | (p.y << 24);
  dst[i * 3 + 1] = ((p.y >> 8) & 0xffffu) | (p.z << 16);
  dst[i * 3 + 2] = ((p.z >> 16) & 0xffu) | (p.w << 8);
}
__global__ __launch_bounds__(256) void k_pack_bgr(const uchar4* __restrict__ src, size_t n, uint8_t* __restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uchar4 p = src[i];
  dst[i * 3] = p.x;
  dst[i * 3 + 1] = p.y;
  dst[i * 3 + 2] = p.z;
}

// ---- sharpen (Filter.h:40-127): 2-tap IIR low pass, wrap horizontally / reflect vertically ----
__device__ __forceinline__ int wrap_i(int x, int r) { return x < 0 ? r + x : x >= r ? x - r : x; }
__device__ __forceinline__ int refl_i(int x, int r) { return x < 0 ? -x : x >= r ? 2 * r - x - 2 : x; }
__device__ __forceinline__ float clamp255(float v) { return v < 0.0f ? 0.0f : v > 255.0f ? 255.0f : v; }
// iirLowPass is a first-order recurrence v = ip*(1-alpha) + v*alpha along every row (then every column), forwards
// and backwards: bit-exactness forbids re-associating it, so the parallelism is one chain per (row, channel) — 16 384
// chains for the row pass of a 4096-row eye — and each chain is 2 x 8400 dependent steps. The kernels below keep a
// chain in one lane (a wave = 16 chains x 4 channels; the alpha lane computes a value nobody reads), move the data in
// tiles of 64 positions through LDS so that global memory is only touched with coalesced row segments (row pass) or
// whole 64/256-byte runs per lane (column pass), and prefetch the next tile while the current one is computed. The
// causal pass writes its float results (float4 per pixel) and each chain's final state; the anticausal pass reads them
// back in reverse and writes the clamped 8-bit low pass — or, for the last pass, the sharpened pixel itself
// (sharpenWithIirLowPass, Filter.h:93-127, fused: the low pass is only ever used there).
constexpr int IIR_CH = 16;   // chains per wave
constexpr int IIR_T = 64;    // positions per tile
constexpr int IIR_LD = IIR_T + 1;  // padded LDS row: 16 chains hit 16 different banks
struct IirGeom {
  int n;        // chain length (ROWS: w, columns: h)
  int nchains;  // ROWS: h, columns: w
  int w;        // image row pitch in pixels
};
constexpr int IIR_MAX_IMGS = 32;  // images per launch (blockIdx.y): the two eyes of every frame slot of a batch
struct IirImgs {
  const uchar4* x[IIR_MAX_IMGS];  // input of the pass
  uchar4* out[IIR_MAX_IMGS];      // 8-bit output of the anticausal half
  float4* buf[IIR_MAX_IMGS];      // float results of the causal half (+ carried state behind them)
};
template <bool ROWS>
__device__ __forceinline__ int iir_bnd(int x, int n) { return ROWS ? wrap_i(x, n) : refl_i(x, n); }
template <bool ROWS>
__device__ __forceinline__ size_t iir_px(const IirGeom& g, int chain, int pos) {
  return ROWS ? (size_t)chain * g.w + pos : (size_t)pos * g.w + chain;
}

// The float results of a causal half live in global memory until the anticausal half has read them: three floats per pixel
// (B, G, R) — the alpha chain's value is never read, the output alpha is 255 (Filter.h:40-127 runs on 3-channel images) —
// i.e. 12 + 12 instead of 16 + 16 bytes per pixel and pass pair.
struct IirPx { float b, g, r; };
__device__ __forceinline__ void iir_put(float* __restrict__ B, size_t px, float4 v) {
  IirPx o{v.x, v.y, v.z};
  *reinterpret_cast<IirPx*>(B + px * 3) = o;
}
typedef float iir_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ iir_f4 iir_get(const float* __restrict__ B, size_t px) {
  const IirPx o = *reinterpret_cast<const IirPx*>(B + px * 3);
  return iir_f4{o.b, o.g, o.r, 0.0f};
}
// Causal half: B[e] = v after consuming X[bnd(e+1)], e = 0..n-1, v0 = X[0]; carry[chain] = final v.
template <bool ROWS>
__global__ __launch_bounds__(64) void k_iir_causal(IirImgs im, IirGeom g, size_t npix, float alpha) {
  const uchar4* __restrict__ X = im.x[blockIdx.y];
  float* __restrict__ Bf = reinterpret_cast<float*>(im.buf[blockIdx.y]);
  float* __restrict__ carry = Bf + npix * 3;
  __shared__ unsigned s_in[IIR_CH * IIR_LD];
  __shared__ float s_out[IIR_CH * IIR_LD * 4];
  const int lane = threadIdx.x, k = lane >> 2, c = lane & 3;
  const int chain0 = blockIdx.x * IIR_CH;
  const int ntiles = (g.n + IIR_T - 1) / IIR_T;
  unsigned pre[IIR_CH];
  // X at positions bnd(e + 1). ROWS: load kk is chain kk at e = 64 t + lane (256 contiguous bytes). Columns: a chain is a
  // column, so a load covers 4 image rows x the 16 chains (four 64-byte runs) — lane = (row lane >> 4 of the group, chain
  // lane & 15), load i is rows 4 i .. 4 i + 3 of the tile — instead of one pixel from each of 64 rows.
  const int cl = lane & 15, rl = lane >> 4;
  auto load_tile = [&](int t) {
    if (ROWS) {
      const int e = min(t * IIR_T + lane, g.n - 1);
      const int pos = iir_bnd<ROWS>(e + 1, g.n);
#pragma unroll
      for (int kk = 0; kk < IIR_CH; ++kk)
        pre[kk] = reinterpret_cast<const unsigned*>(X)[iir_px<ROWS>(g, min(chain0 + kk, g.nchains - 1), pos)];
    } else {
      const int ch = min(chain0 + cl, g.nchains - 1);
#pragma unroll
      for (int i = 0; i < IIR_CH; ++i) {
        const int e = min(t * IIR_T + 4 * i + rl, g.n - 1);
        pre[i] = reinterpret_cast<const unsigned*>(X)[(size_t)iir_bnd<ROWS>(e + 1, g.n) * g.w + ch];
      }
    }
  };
  const float am = 1.0f - alpha;
  float v;
  {
    const uchar4 p0 = X[iir_px<ROWS>(g, min(chain0 + k, g.nchains - 1), 0)];
    v = c == 0 ? (float)p0.x : c == 1 ? (float)p0.y : c == 2 ? (float)p0.z : (float)p0.w;
  }
  load_tile(0);
  for (int t = 0; t < ntiles; ++t) {
#pragma unroll
    for (int kk = 0; kk < IIR_CH; ++kk) {
      if (ROWS) s_in[kk * IIR_LD + lane] = pre[kk];
      else s_in[cl * IIR_LD + 4 * kk + rl] = pre[kk];
    }
    S360_WAVE_SYNC();  // (one wave per workgroup: the tile is written position-major and read chain-major)
    if (t + 1 < ntiles) load_tile(t + 1);
    const int cnt = min(IIR_T, g.n - t * IIR_T);
    const unsigned char* sb = reinterpret_cast<const unsigned char*>(s_in) + (size_t)k * IIR_LD * 4 + c;
    float* so = s_out + (size_t)k * IIR_LD * 4 + c;
    int j = 0;
    for (; j + 8 <= cnt; j += 8) {  // 8 inputs fetched together: the dependent chain is then 2 VALU ops per step
      float ip[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) ip[q] = (float)sb[(j + q) * 4] * am;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        v = ip[q] + v * alpha;  // lerp(ip, v, alpha) = ip*(1-alpha) + v*alpha, MathUtil.h
        so[(j + q) * 4] = v;
      }
    }
    for (; j < cnt; ++j) {
      v = (float)sb[j * 4] * am + v * alpha;
      so[j * 4] = v;
    }
    S360_WAVE_SYNC();
    // store the tile's float results: ROWS: 1 KB per chain row; columns: four 256-byte runs (4 rows x 16 chains) per store
    if (ROWS) {
      const int e = t * IIR_T + lane;
      if (e < g.n) {
#pragma unroll
        for (int kk = 0; kk < IIR_CH; ++kk) {
          if (chain0 + kk < g.nchains) {
            const float4 o = *reinterpret_cast<const float4*>(s_out + ((size_t)kk * IIR_LD + lane) * 4);
            iir_put(Bf, iir_px<ROWS>(g, chain0 + kk, e), o);
          }
        }
      }
    } else if (chain0 + cl < g.nchains) {
#pragma unroll
      for (int i = 0; i < IIR_CH; ++i) {
        const int p = 4 * i + rl, e = t * IIR_T + p;
        if (e < g.n) iir_put(Bf, (size_t)e * g.w + chain0 + cl, *reinterpret_cast<const float4*>(s_out + ((size_t)cl * IIR_LD + p) * 4));
      }
    }
  }
  if (chain0 + k < g.nchains) carry[(size_t)(chain0 + k) * 4 + c] = v;
}

// Anticausal half: for e = n-1 .. 0: v = lerp(B[bnd(e-1)], v); OUT[e] = clamp(v). FUSE: OUT is the unsharp mask of
// `img` against that low-pass value, written in place.
template <bool ROWS, bool FUSE>
__global__ __launch_bounds__(64) void k_iir_anticausal(IirImgs im, IirGeom g, size_t npix, float alpha, float amount) {
  const float* __restrict__ Bf = reinterpret_cast<const float*>(im.buf[blockIdx.y]);
  const float* __restrict__ carry = Bf + npix * 3;
  uchar4* __restrict__ out = im.out[blockIdx.y];
  __shared__ float s_in[IIR_CH * IIR_LD * 4];
  __shared__ unsigned s_out[IIR_CH * IIR_LD];
  const int lane = threadIdx.x, k = lane >> 2, c = lane & 3;
  const int chain0 = blockIdx.x * IIR_CH;
  const int ntiles = (g.n + IIR_T - 1) / IIR_T;
  // (native vector type: an array of HIP's float4 structs captured by the lambda is not promoted to registers — it was
  // 272 bytes of scratch memory per lane, with every prefetched tile waited for at once in order to be stored there)
  typedef float f4r __attribute__((ext_vector_type(4)));
  f4r pre[IIR_CH];
  const int cl = lane & 15, rl = lane >> 4;  // columns: lane = (row of a group of 4, chain), as in k_iir_causal
  auto load_tile = [&](int t) {  // B at positions bnd(e - 1)
    if (ROWS) {
      const int e = min(t * IIR_T + lane, g.n - 1);
      const int pos = iir_bnd<ROWS>(e - 1, g.n);
#pragma unroll
      for (int kk = 0; kk < IIR_CH; ++kk)
        pre[kk] = iir_get(Bf, iir_px<ROWS>(g, min(chain0 + kk, g.nchains - 1), pos));
    } else {
      const int ch = min(chain0 + cl, g.nchains - 1);
#pragma unroll
      for (int i = 0; i < IIR_CH; ++i) {
        const int e = min(t * IIR_T + 4 * i + rl, g.n - 1);
        pre[i] = iir_get(Bf, (size_t)iir_bnd<ROWS>(e - 1, g.n) * g.w + ch);
      }
    }
  };
  const float am = 1.0f - alpha;
  float v = carry[(size_t)min(chain0 + k, g.nchains - 1) * 4 + c];
  load_tile(ntiles - 1);
  for (int t = ntiles - 1; t >= 0; --t) {
#pragma unroll
    for (int kk = 0; kk < IIR_CH; ++kk) {
      if (ROWS) *reinterpret_cast<f4r*>(s_in + ((size_t)kk * IIR_LD + lane) * 4) = pre[kk];
      else *reinterpret_cast<f4r*>(s_in + ((size_t)cl * IIR_LD + 4 * kk + rl) * 4) = pre[kk];
    }
    S360_WAVE_SYNC();
    if (t > 0) load_tile(t - 1);
    // FUSE: the image pixels the unsharp mask needs at the end of this tile are requested now (as load-use pairs in the
    // output loop they were 16 serialised round trips per tile)
    unsigned img[IIR_CH];
    if (FUSE) {
#pragma unroll
      for (int kk = 0; kk < IIR_CH; ++kk) {
        const int p = ROWS ? lane : 4 * kk + rl, chn = min(chain0 + (ROWS ? kk : cl), g.nchains - 1), e = min(t * IIR_T + p, g.n - 1);
        img[kk] = reinterpret_cast<const unsigned*>(out)[iir_px<ROWS>(g, chn, e)];
      }
    }
    const int cnt = min(IIR_T, g.n - t * IIR_T);
    const float* si = s_in + (size_t)k * IIR_LD * 4 + c;
    unsigned char* so = reinterpret_cast<unsigned char*>(s_out) + (size_t)k * IIR_LD * 4 + c;
    int j = cnt - 1;
    for (; j >= 7; j -= 8) {
      float ip[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) ip[q] = si[(j - q) * 4] * am;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        v = ip[q] + v * alpha;
        so[(j - q) * 4] = c == 3 ? (unsigned char)255 : (unsigned char)clamp255(v);
      }
    }
    for (; j >= 0; --j) {
      v = si[j * 4] * am + v * alpha;
      so[j * 4] = c == 3 ? (unsigned char)255 : (unsigned char)clamp255(v);
    }
    S360_WAVE_SYNC();
#pragma unroll
    for (int kk = 0; kk < IIR_CH; ++kk) {
      // ROWS: store kk is chain kk at position 64 t + lane; columns: rows 4 kk .. 4 kk + 3 of the tile x the 16 chains
      const int p = ROWS ? lane : 4 * kk + rl, chn = chain0 + (ROWS ? kk : cl), e = t * IIR_T + p;
      if (e < g.n) {
        if (chn < g.nchains) {
          const unsigned lw = s_out[(ROWS ? kk : cl) * IIR_LD + p];
          const size_t o = iir_px<ROWS>(g, chn, e);
          if (FUSE) {
            uchar4 px = __builtin_bit_cast(uchar4, img[kk]);
            auto f = [&](unsigned char pc, unsigned lc) -> unsigned char {
              const float lf = (float)lc;
              const float hp = (float)pc - lf;
              // noise coring: 1 - expf(-(hp^2 * 100)). hp is an integer: the factor is exactly 0 for hp == 0 and
              // exactly 1.0f otherwise (expf(-100) < 2^-24), so no transcendental is needed on the device.
              const float ng = hp == 0.0f ? 0.0f : 1.0f;
              return (unsigned char)clamp255(lf + hp * ng * amount);
            };
            px.x = f(px.x, lw & 255u); px.y = f(px.y, (lw >> 8) & 255u); px.z = f(px.z, (lw >> 16) & 255u);
            out[o] = px;
          } else {
            reinterpret_cast<unsigned*>(out)[o] = lw;
          }
        }
      }
    }
  }
}

// 16-bit samples -> their high byte: what imread's 8-bit decode makes of the ISP's 16-bit PNGs (png_set_strip_16)
__global__ __launch_bounds__(256) void k_u16_high_byte(const unsigned short* __restrict__ src, uint8_t* __restrict__ dst,
                                                       size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (uint8_t)(src[i] >> 8);
}

// ==========================================================================================
static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

void launch_u16_high_byte(hipStream_t st, const unsigned short* src, uint8_t* dst, size_t n) {
  hipLaunchKernelGGL(k_u16_high_byte, dim3(cdiv(n, 256)), dim3(256), 0, st, src, dst, n);
}

void launch_bgr_to_bgra(hipStream_t st, const uint8_t* src, int channels, uchar4* dst, size_t n) {
  hipLaunchKernelGGL(k_bgr_to_bgra, dim3(cdiv(n, 256)), dim3(256), 0, st, src, channels, dst, n);
}
void launch_prepare_side_src(hipStream_t st, const uint8_t* src, int channels, uchar4* dst, int w, int h, int feather) {
  hipLaunchKernelGGL(k_prepare_side_src, dim3(cdiv(w, 256), h), dim3(256), 0, st, src, channels, dst, w, h, feather);
}
void launch_spherical_map(hipStream_t st, float2* map, int dw, int dh, const DevCamera& cam, const float* cosX,
                          const float* sinX, const float* cosY, const float* sinY) {
  dim3 blk(64, 4);
  hipLaunchKernelGGL(k_spherical_map, dim3(cdiv(dw, 64), cdiv(dh, 4)), blk, 0, st, map, dw, dh, cam, cosX, sinX, cosY,
                     sinY);
}
void launch_remap_cubic_u8c4(hipStream_t st, const uchar4* src, int sw, int sh, const float2* map, uchar4* dst, int dw,
                             int dh, const DevTables& T, int alpha_mode, int yFeatherStart, int featherSize, int batch) {
  MapFromBuffer mf{map, dw};
  hipLaunchKernelGGL((k_remap_cubic_u8c4_tiled<MapFromBuffer>), dim3(cdiv(dw, RT_W), cdiv(dh, RT_H), batch),
                     dim3(RT_W, RT_H), 0, st, src, sw, sh, mf, dst, dw, dh, T.bicubic_i, alpha_mode, yFeatherStart,
                     featherSize, (size_t)sw * sh, (size_t)dw * dh);
}
size_t remap_packed_tiles(int dw, int dh) { return (size_t)cdiv(dw, PT_W) * cdiv(dh, PT_H); }
void launch_remap_pack_map(hipStream_t st, const float2* map, int sw, int sh, int dw, int dh, unsigned* packed, void* tiles,
                           int batch) {
  MapFromBuffer mf{map, dw};
  hipLaunchKernelGGL((k_remap_pack<MapFromBuffer>), dim3(cdiv(dw, PT_W), cdiv(dh, PT_H), batch), dim3(PT_W, PT_TY), 0, st, mf, sw,
                     sh, dw, dh, packed, reinterpret_cast<int4*>(tiles), (size_t)dw * dh, (int)remap_packed_tiles(dw, dh));
}
static bool remap_rebuild_weights() {
  static const bool on = [] { const char* e = std::getenv("S360_REMAP_REBUILD_WEIGHTS"); return !(e && e[0] == '0'); }();
  return on;
}
void launch_remap_cubic_u8c4_packed(hipStream_t st, const uchar4* src, int sw, int sh, const float2* map, const unsigned* packed,
                                    const void* tiles, uchar4* dst, int dw, int dh, const DevTables& T, int alpha_mode,
                                    int yFeatherStart, int featherSize, int batch) {
  MapFromBuffer mf{map, dw};
  const dim3 grid(cdiv(dw, PT_W), cdiv(dh, PT_H), batch), block(PT_W, PT_TY);
  const int4* t4 = reinterpret_cast<const int4*>(tiles);
  const size_t sbs = (size_t)sw * sh, dbs = (size_t)dw * dh;
  const int nt = (int)remap_packed_tiles(dw, dh);
  // The weights are rebuilt in the kernel (WT) unless the host's rebuild of the table failed or S360_REMAP_REBUILD_WEIGHTS=0 asks
  // for the table (the A/B switch of round 6's measurement: the kernel's header)
  const bool wt = remap_rebuild_weights() && T.bicubic_res;
#define S360_RP(A, W)                                                                                                              \
  hipLaunchKernelGGL((k_remap_cubic_u8c4_packed<MapFromBuffer, A, W>), grid, block, 0, st, src, sw, sh, packed, t4, mf, dst, dw, dh, \
                     T.bicubic_i, yFeatherStart, featherSize, sbs, dbs, nt, T.bicubic_w1, T.bicubic_res)
  if (alpha_mode == 1) { if (wt) S360_RP(1, true); else S360_RP(1, false); }
  else if (alpha_mode == 2) { if (wt) S360_RP(2, true); else S360_RP(2, false); }
  else { if (wt) S360_RP(0, true); else S360_RP(0, false); }
#undef S360_RP
}
void launch_remap_by_flow(hipStream_t st, const uchar4* src, int w, int h, const float2* flow, uchar4* dst,
                          const DevTables& T) {
  MapFromFlowAdd mf{flow, w};
  hipLaunchKernelGGL((k_remap_cubic_u8c4_tiled<MapFromFlowAdd>), dim3(cdiv(w, RT_W), cdiv(h, RT_H), 1), dim3(RT_W, RT_H), 0, st,
                     src, w, h, mf, dst, w, h, T.bicubic_i, 0, 0, 1, (size_t)0, (size_t)0);
}
void launch_red_mask(hipStream_t st, const uint8_t* bgr, uint8_t* red, size_t n) {
  hipLaunchKernelGGL(k_red_mask, dim3(cdiv(n, 256)), dim3(256), 0, st, bgr, red, n);
}
void launch_circle_alpha(hipStream_t st, const uchar4* src, const uint8_t* red, uchar4* dst, int w, int h, float radius) {
  hipLaunchKernelGGL(k_circle_alpha, dim3(cdiv(w, 256), h), dim3(256), 0, st, src, red, dst, w, h, radius);
}
void launch_pole_removal_combine(hipStream_t st, uchar4* bottom, const uchar4* warped2, size_t n) {
  hipLaunchKernelGGL(k_pole_removal_combine, dim3(cdiv(n, 256)), dim3(256), 0, st, bottom, warped2, n);
}
void launch_crop_overlaps(hipStream_t st, const uchar4* proj, int camW, int camH, int P, int overlapW, uchar4* out,
                          int p0, int p1) {
  const int n = p1 - p0;
  if (n <= 0) return;
  if ((camW & 1) == 0 && (overlapW & 1) == 0 && (reinterpret_cast<uintptr_t>(proj) & 7) == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0)
    hipLaunchKernelGGL(k_crop_overlaps_v2, dim3(cdiv(overlapW / 2, 256), camH, 2 * n), dim3(256), 0, st,
                       reinterpret_cast<const uint2*>(proj), camW / 2, camH, P, overlapW / 2, reinterpret_cast<uint2*>(out), p0, n);
  else
    hipLaunchKernelGGL(k_crop_overlaps, dim3(cdiv(overlapW, 256), camH, 2 * n), dim3(256), 0, st, proj, camW, camH, P,
                       overlapW, out, p0, n);
}
void launch_novel_view(hipStream_t st, const uchar4* overlaps, const float2* flows, uchar4* strips,
                       const NovelViewParams& nv, int p0, int p1, const DevTables& T) {
  if (p1 <= p0) return;
  hipLaunchKernelGGL(k_novel_view, dim3(cdiv(nv.stripW, NV_TW), cdiv(nv.camH, NV_TH), 2 * (p1 - p0)), dim3(NV_TW * NV_TH), 0, st,
                     overlaps, flows, strips, nv, p0, T);
}
void launch_assemble_pano(hipStream_t st, const uchar4* strips_eye, int P, int camH, int stripW, float offset,
                          uchar4* pano, int eqrW, int eqrH) {
  const int padAbove = (eqrH - camH) / 2;
  hipLaunchKernelGGL(k_assemble_pano, dim3(cdiv(cdiv(eqrW, 4), 256), eqrH), dim3(256), 0, st, strips_eye, P, camH, stripW, offset,
                     pano, eqrW, eqrH, padAbove);
}
void launch_flip_both(hipStream_t st, const uchar4* src, uchar4* dst, int w, int h, int rows) {
  hipLaunchKernelGGL(k_flip_both, dim3(cdiv(w, 256), rows), dim3(256), 0, st, src, dst, w, h);
}
// featherAlphaChannel's erode (alpha of a BGRA image -> eroded 8-bit plane) and Gaussian
void launch_erode_alpha(hipStream_t st, const uchar4* img, uint8_t* out, int w, int h, int e) {
  if (e > ER_MAXE) throw std::runtime_error("featherAlphaChannel: erode size above 32 is not supported");
  if (e == 31 && (w & 1) == 0)  // the reference's default std_alpha_feather_size; (even width: 2-byte stores)
    hipLaunchKernelGGL((k_erode_cross_fixed<31>), dim3(cdiv(w, ERF_TW), cdiv(h, ERF_TH)), dim3(256), 0, st, img, out, w, h);
  else
    hipLaunchKernelGGL(k_erode_cross_tiled, dim3(cdiv(w, ER_TW), cdiv(h, ER_TH)), dim3(256), 0, st, img, out, w, h, e);
}
void launch_gauss_u8(hipStream_t st, const uint8_t* a, uint8_t* out, int w, int h, const int* ik, int r) {
  if (r > GU_MAXR) throw std::runtime_error("featherAlphaChannel: Gaussian radius above 16 is not supported");
  if (r == 15)  // ksize 31: the reference's default std_alpha_feather_size
    hipLaunchKernelGGL((k_gauss_u8_fixed<15>), dim3(cdiv(w, GU_TW), cdiv(h, GU_TH)), dim3(256), 0, st, a, out, w, h, ik);
  else
    hipLaunchKernelGGL(k_gauss_u8_tiled, dim3(cdiv(w, GU_TW), cdiv(h, GU_TH)), dim3(256), 0, st, a, out, w, h, ik, r);
}
void launch_extend_wrap(hipStream_t st, const uchar4* img, const uint8_t* alpha, int cols, int rows, uchar4* ext,
                        int extW) {
  if ((cols & 3) == 0 && (extW & 3) == 0 && (reinterpret_cast<uintptr_t>(img) & 15) == 0 && (reinterpret_cast<uintptr_t>(ext) & 15) == 0 &&
      (reinterpret_cast<uintptr_t>(alpha) & 3) == 0)
    hipLaunchKernelGGL(k_extend_wrap_v4, dim3(cdiv(extW / 4, 256), rows), dim3(256), 0, st, reinterpret_cast<const uint4*>(img),
                       reinterpret_cast<const unsigned*>(alpha), cols / 4, rows, reinterpret_cast<uint4*>(ext), extW / 4);
  else
    hipLaunchKernelGGL(k_extend_wrap, dim3(cdiv(extW, 256), rows), dim3(256), 0, st, img, alpha, cols, rows, ext, extW);
}
void launch_pole_warp_packed(hipStream_t st, const uchar4* extFisheye, const float2* flow, uchar4* warpedExt,
                             const PoleWarpParams& pw, const DevTables& T, unsigned* packed, void* tiles) {
  MapFromPoleFlow mf{flow, pw};
  const int nt = (int)remap_packed_tiles(pw.extW, pw.rows);
  hipLaunchKernelGGL((k_remap_pack<MapFromPoleFlow>), dim3(cdiv(pw.extW, PT_W), cdiv(pw.rows, PT_H), 1), dim3(PT_W, PT_TY), 0, st,
                     mf, pw.extW, pw.rows, pw.extW, pw.rows, packed, reinterpret_cast<int4*>(tiles), (size_t)0, nt);
  // (the table here: this frame's warp is at 0.78 of the integer issue roof already, and rebuilding the weights — 35 more VALU
  // operations per pixel — measured 0.865 against 0.787 ms per 8K frame; the statically mapped projections, which wait more than
  // they compute, gain: profiles/r06_v5_remap_rebuilt_weights_ab.txt)
  hipLaunchKernelGGL((k_remap_cubic_u8c4_packed<MapFromPoleFlow, 0, false>), dim3(cdiv(pw.extW, PT_W), cdiv(pw.rows, PT_H), 1),
                     dim3(PT_W, PT_TY), 0, st, extFisheye, pw.extW, pw.rows, packed, reinterpret_cast<const int4*>(tiles), mf,
                     warpedExt, pw.extW, pw.rows, T.bicubic_i, 0, 1, (size_t)0, (size_t)0, nt, T.bicubic_w1, T.bicubic_res);
}
void launch_pole_finish(hipStream_t st, const uchar4* warpedExt, uchar4* out, int eqrH, const PoleWarpParams& pw) {
  if ((pw.cols & 3) == 0 && (pw.extW & 3) == 0 && pw.cols + ((pw.maxBlendX + 3) & ~3) <= pw.extW &&
      (reinterpret_cast<uintptr_t>(warpedExt) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0)
    hipLaunchKernelGGL(k_pole_finish_v4, dim3(cdiv(pw.cols / 4, 256), eqrH), dim3(256), 0, st,
                       reinterpret_cast<const uint4*>(warpedExt), reinterpret_cast<uint4*>(out), eqrH, pw);
  else
    hipLaunchKernelGGL(k_pole_finish, dim3(cdiv(pw.cols, 256), eqrH), dim3(256), 0, st, warpedExt, out, eqrH, pw);
}
void launch_flatten(hipStream_t st, const uchar4* base, const uchar4* top, uchar4* out, int w, int h, int flip_top,
                    const DevTables& T) {
  if ((w & 3) == 0)
    hipLaunchKernelGGL(k_flatten_v4, dim3(cdiv(w / 4, 256), h), dim3(256), 0, st, reinterpret_cast<const uint4*>(base),
                       reinterpret_cast<const uint4*>(top), reinterpret_cast<uint4*>(out), w / 4, h, flip_top, T);
  else
    hipLaunchKernelGGL(k_flatten, dim3(cdiv(w, 256), h), dim3(256), 0, st, base, top, out, w, h, flip_top, T);
}
void launch_cubemap(hipStream_t st, const uchar4* eyeL, const uchar4* eyeR, int sw, int sh, const float2* maps, int fw,
                    int fh, int video, uint8_t* out, const DevTables& T) {
  const int ow = video ? 3 * fw : fw, oh = video ? 4 * fh : 12 * fh;
  hipLaunchKernelGGL(k_cubemap, dim3(cdiv(ow, 256), oh), dim3(256), 0, st, eyeL, eyeR, sw, sh, maps, fw, fh, video, out, ow,
                     oh, T.bicubic_i);
}
void launch_pack_bgr(hipStream_t st, const uchar4* src, int w, int h, uint8_t* dst) {
  const size_t n = (size_t)w * h;
  if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0)
    hipLaunchKernelGGL(k_pack_bgr_v4, dim3(cdiv(n / 4, 256)), dim3(256), 0, st, reinterpret_cast<const uint4*>(src), n / 4,
                       reinterpret_cast<unsigned*>(dst));
  else
    hipLaunchKernelGGL(k_pack_bgr, dim3(cdiv(n, 256)), dim3(256), 0, st, src, n, dst);
}
// iirLowPass (wrap horizontally, reflect vertically) + sharpenWithIirLowPass on one eye, in place (TRSP:688-696).
// scratch: w*h float4 + max(w,h) float4 (the chains' carried state).
int sharpen_max_images() { return IIR_MAX_IMGS; }
size_t sharpen_scratch_bytes(int w, int h) { return ((size_t)w * h + (size_t)std::max(w, h)) * sizeof(float4); }
// n images of the same size in one set of launches (the chains of one image cannot fill the chip: 16 384 of them for
// the row pass of a 4096-row eye, each 2 x 8400 dependent steps).
void launch_sharpen_many(hipStream_t st, uchar4* const* imgs, uchar4* const* lps, float* const* scratch, int n, int w,
                         int h, float amount) {
  if (n < 1 || n > IIR_MAX_IMGS) throw std::runtime_error("launch_sharpen_many: 1..32 images per launch");
  const float alpha = powf(0.25f, 1.0f / 4.0f);  // host libm, Filter.h:49
  const size_t npix = (size_t)w * h;
  const IirGeom gr{w, h, w}, gc{h, w, w};
  IirImgs rows, cols;
  for (int i = 0; i < IIR_MAX_IMGS; ++i) {
    const int k = i < n ? i : 0;
    rows.x[i] = imgs[k]; rows.out[i] = lps[k]; rows.buf[i] = reinterpret_cast<float4*>(scratch[k]);
    cols.x[i] = lps[k]; cols.out[i] = imgs[k]; cols.buf[i] = reinterpret_cast<float4*>(scratch[k]);
  }
  hipLaunchKernelGGL((k_iir_causal<true>), dim3(cdiv(h, IIR_CH), n), dim3(64), 0, st, rows, gr, npix, alpha);
  hipLaunchKernelGGL((k_iir_anticausal<true, false>), dim3(cdiv(h, IIR_CH), n), dim3(64), 0, st, rows, gr, npix, alpha, amount);
  hipLaunchKernelGGL((k_iir_causal<false>), dim3(cdiv(w, IIR_CH), n), dim3(64), 0, st, cols, gc, npix, alpha);
  hipLaunchKernelGGL((k_iir_anticausal<false, true>), dim3(cdiv(w, IIR_CH), n), dim3(64), 0, st, cols, gc, npix, alpha, amount);
}
void launch_sharpen(hipStream_t st, uchar4* img, uchar4* lp, float* scratch, int w, int h, float amount) {
  launch_sharpen_many(st, &img, &lp, &scratch, 1, w, h, amount);
}

}  // namespace s360
