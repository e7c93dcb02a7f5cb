// median.hip — medianBlur(flow, 5) of PixFlow (PixFlow.h:398,411) for gfx950.
//
// Its own translation unit because it is built with -ffinite-math-only: flows never hold NaNs, and without that
// promise every fminf / fmaxf input loaded from memory is first canonicalised (v_max_f32 x, x, x) — 120 extra
// instructions per 16 medians (9 % of the kernel). Nothing else in this file does floating-point arithmetic; the
// result of a min / max of non-NaN values is the same instruction's result either way.
#include "flow_kernels.hpp"

#include "devmath.hpp"

namespace s360 {

typedef float f4a8 __attribute__((ext_vector_type(4), aligned(8)));

static inline dim3 grid2d(int w, int h, int B, dim3 blk) { return dim3((w + blk.x - 1) / blk.x, (h + blk.y - 1) / blk.y, B); }

// ------------------------------------------------------------------------------------------
// medianBlur(5) on CV_32FC2, replicate border (PixFlow.h:398,411): exact per-channel median of 25.
__device__ __forceinline__ void mnmx(float& a, float& b) {
  const float lo = fminf(a, b), hi = fmaxf(a, b);
  a = lo;
  b = hi;
}
// Sort of three: v_min3 / v_med3 / v_max3 — three instructions where three compare-exchanges take six.
__device__ __forceinline__ void sort3(float& a, float& b, float& c) {
  const float lo = fminf(fminf(a, b), c), hi = fmaxf(fmaxf(a, b), c), md = __builtin_amdgcn_fmed3f(a, b, c);
  a = lo;
  b = md;
  c = hi;
}
// Exact median of 25 with the 99-comparator selection network of Devillard's "Fast median search" (after Paeth,
// Graphics Gems): verified for all 2^25 0/1 inputs (0-1 principle), tools/verify_median_network.py. 66 of its
// comparators form 22 runs of three that sort three wires; those are S360_S3 (same function, half the instructions).
// Comparators whose outputs are never read again are removed by the compiler.
__device__ __forceinline__ float median25(const float* in) {
  float p[25];
#pragma unroll
  for (int i = 0; i < 25; ++i) p[i] = in[i];
#define S360_CE(a, b) mnmx(p[a], p[b])
#define S360_S3(a, b, c) sort3(p[a], p[b], p[c])
  S360_CE(0, 1); S360_S3(2, 3, 4); S360_S3(5, 6, 7); S360_S3(8, 9, 10); S360_S3(11, 12, 13); S360_S3(14, 15, 16);
  S360_S3(17, 18, 19); S360_S3(20, 21, 22); S360_CE(23, 24); S360_CE(2, 5); S360_S3(0, 3, 6); S360_S3(1, 4, 7);
  S360_S3(8, 11, 14); S360_S3(9, 12, 15); S360_S3(10, 13, 16); S360_S3(17, 20, 23); S360_S3(18, 21, 24);
  S360_CE(19, 22); S360_CE(8, 17); S360_S3(0, 9, 18); S360_S3(1, 10, 19); S360_S3(2, 11, 20); S360_S3(3, 12, 21);
  S360_S3(4, 13, 22); S360_S3(5, 14, 23); S360_S3(6, 15, 24); S360_CE(7, 16); S360_CE(7, 19); S360_CE(13, 21);
  S360_CE(15, 23); S360_CE(7, 13); S360_CE(7, 15); S360_CE(1, 9); S360_CE(3, 11); S360_CE(5, 17); S360_CE(11, 17);
  S360_CE(9, 17); S360_CE(4, 10); S360_CE(6, 12); S360_CE(7, 14); S360_CE(4, 6); S360_CE(4, 7); S360_CE(12, 14);
  S360_CE(10, 14); S360_CE(6, 7); S360_CE(10, 12); S360_CE(6, 10); S360_CE(6, 17); S360_CE(12, 17); S360_CE(7, 17);
  S360_CE(7, 10); S360_CE(12, 18); S360_CE(7, 12); S360_CE(10, 18); S360_S3(10, 12, 20);
#undef S360_CE
#undef S360_S3
  return p[12];
}
__global__ __launch_bounds__(256) void k_median5_c2(const float2* __restrict__ src, float2* __restrict__ dst, int w,
                                                    int h, size_t bs) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= w || y >= h) return;
  src += bs * blockIdx.z;
  dst += bs * blockIdx.z;
  float vx[25], vy[25];
#pragma unroll
  for (int dy = -2; dy <= 2; ++dy) {
    const float2* r = src + (size_t)clip_idx(y + dy, h) * w;
#pragma unroll
    for (int dx = -2; dx <= 2; ++dx) {
      const float2 p = r[clip_idx(x + dx, w)];
      vx[(dy + 2) * 5 + dx + 2] = p.x;
      vy[(dy + 2) * 5 + dx + 2] = p.y;
    }
  }
  float2 o;
  o.x = median25(vx);
  o.y = median25(vy);
  dst[(size_t)y * w + x] = o;
}

// The same medians for 8 horizontally adjacent pixels per thread from 12 shared columns: every column of 5 is sorted
// once and used by 5 windows, aligned column pairs are merged once and used by 4, and the 6 median candidates of two
// adjacent pairs are selected once and used by 2; the stages are programs over min / max / min3 / max3 / med3 — 66
// instructions per median instead of 112 — and 60 eight-byte loads per 8 pixels instead of 200. The networks are generated
// and verified by tools/gen_median_network.py.
#include "median_tile.inc"
constexpr int MED_T = 8;  // outputs per thread
// A block of 256 threads produces a tile of (BX * 8) x (256 / BX) pixels, BX = 32 / 16 / 8 / 4 threads across: 256 x 8,
// 128 x 16, 64 x 32 or 32 x 64. The pyramid's levels have every width (x 0.9 per level); with the 256-pixel tile alone a
// level 303 wide ran two tiles per row, the second with 6 of its 32 thread columns in use — every wave walks the whole
// network whatever its active lanes — and the side flows' levels used 72 % of the lanes they launched (pole flows: 91 %).
// The launcher takes the shape that wastes least (92 % / 96 %); the narrower tiles also load less halo per pixel.
template <int BX>
struct MedGeom {
  static constexpr int BY = 256 / BX;
  static constexpr int LW = BX * MED_T + 4, LH = BY + 4;  // the tile with its 2-pixel replicate border
  static constexpr int NG = LW / 4, GH = (NG + 1) / 2;    // 4-float groups of a tile row / of its even half
  // floats per LDS row: a wave reads 16 bytes per lane from 64 / BX rows at once, and the rows must start 16 * BX bytes apart
  // modulo the 256 bytes of the banks for those reads to miss each other
  static constexpr int RS = BX == 32 ? LW : BX == 16 ? 136 : BX == 8 ? 96 : 48;
  static_assert((LW & 3) == 0 && RS >= 2 * GH * 4 - 4, "whole groups, and the permuted row fits");
};
// The tile (with its 2-pixel replicate border) goes through LDS once, split into its two channels: a thread then reads
// the 60 values of ONE channel at a time (three 16-byte LDS reads per window row), which keeps the generated network
// at ~100 registers instead of the ~190 it needs with both channels' inputs live.
// column -> position in the LDS row (even groups, then odd groups): a thread's three 16-byte reads then fall on consecutive
// addresses across the lanes instead of every other group (a two-way bank conflict on every read).
template <int BX>
__device__ __forceinline__ int med_pos(int lx) {
  const int g = lx >> 2;
  return (((g >> 1) + (g & 1) * MedGeom<BX>::GH) << 2) | (lx & 3);
}
template <int BX>
__global__ __launch_bounds__(256) void k_median5_c2_row8(const float2* __restrict__ src, float2* __restrict__ dst, int w,
                                                         int h, size_t bs) {
  typedef MedGeom<BX> G;
  __shared__ __attribute__((aligned(16))) float s_p[2][G::LH][G::RS];
  const int tid = threadIdx.x;
  const TileId tile = xcd_tile();  // neighbouring tiles (shared 2-pixel halo) on the same XCD's L2
  const int X0 = tile.x * (BX * MED_T), Y0 = tile.y * G::BY;
  src += bs * tile.z;
  dst += bs * tile.z;
  if (BX == 32) {
    // A thread keeps its column and walks down the 12 rows (no division, one clamped column index); columns 256..259
    // are taken by the first four threads.
    for (int lx = tid; lx < G::LW; lx += 256) {
      const int gx = clip_idx(X0 - 2 + lx, w), px = med_pos<BX>(lx);
#pragma unroll
      for (int ly = 0; ly < G::LH; ++ly) {
        const float2 p = src[(size_t)clip_idx(Y0 - 2 + ly, h) * w + gx];
        s_p[0][ly][px] = p.x;
        s_p[1][ly][px] = p.y;
      }
    }
  } else {
    // The tile's inner BX * 8 columns: a thread keeps its column (one clamped column index, one LDS position) and takes
    // every (256 / width)-th row; the 4 border columns are 4 * LH elements of their own. All of a thread's elements are
    // requested before the first goes to LDS (one exposed memory round trip per tile). As one row-major list over the
    // threads the index arithmetic per element (row and column of a varying linear index, two clamps, a 64-bit address)
    // was 210 instructions per thread beside the network's 1040.
    constexpr int TW = BX * MED_T, RP = 256 / TW, kMain = (G::LH + RP - 1) / RP, kHalo = (4 * G::LH + 255) / 256;
    // (a wave lies inside one tile row when the tile is at least 64 wide: its row arithmetic is then scalar)
    const int c = tid & (TW - 1), r0 = TW >= 64 ? __builtin_amdgcn_readfirstlane(tid / TW) : tid / TW;
    const float2* col = src + clip_idx(X0 + c, w);
    float2 ld[kMain], lh[kHalo];
#pragma unroll
    for (int it = 0; it < kMain; ++it)  // (rows behind the tile's end are read from its last row and never stored)
      ld[it] = col[(size_t)clip_idx(Y0 - 2 + min(r0 + it * RP, G::LH - 1), h) * w];
#pragma unroll
    for (int it = 0; it < kHalo; ++it) {
      const int e = min(tid + it * 256, 4 * G::LH - 1), j = e & 3;
      lh[it] = src[(size_t)clip_idx(Y0 - 2 + (e >> 2), h) * w + clip_idx(X0 - 2 + (j < 2 ? j : TW + j), w)];
    }
    const int px = med_pos<BX>(c + 2);
#pragma unroll
    for (int it = 0; it < kMain; ++it) {
      const int ly = r0 + it * RP;
      if (ly < G::LH) {
        s_p[0][ly][px] = ld[it].x;
        s_p[1][ly][px] = ld[it].y;
      }
    }
#pragma unroll
    for (int it = 0; it < kHalo; ++it) {
      const int e = tid + it * 256, j = e & 3;
      if (e < 4 * G::LH) {
        const int ph = med_pos<BX>(j < 2 ? j : TW + j);
        s_p[0][e >> 2][ph] = lh[it].x;
        s_p[1][e >> 2][ph] = lh[it].y;
      }
    }
  }
  __syncthreads();
  const int tx = tid & (BX - 1), ty = tid / BX;
  const int x0 = X0 + tx * MED_T, y = Y0 + ty;
  if (x0 >= w || y >= h) return;
  float o[2][MED_T];
#pragma unroll
  for (int ch = 0; ch < 2; ++ch) {
    float in[(MED_T + 4) * 5];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      const float4* row = reinterpret_cast<const float4*>(&s_p[ch][ty + r][0]);
#pragma unroll
      for (int q = 0; q < (MED_T + 4) / 4; ++q) {
        const float4 v = row[med_pos<BX>(tx * MED_T + 4 * q) >> 2];
        in[(4 * q) * 5 + r] = v.x;
        in[(4 * q + 1) * 5 + r] = v.y;
        in[(4 * q + 2) * 5 + r] = v.z;
        in[(4 * q + 3) * 5 + r] = v.w;
      }
    }
    median5x5_row8(in, o[ch]);
  }
  float2* out = dst + (size_t)y * w + x0;
  if (x0 + MED_T <= w) {
#pragma unroll
    for (int k = 0; k < MED_T; k += 2) {
      f4a8 q = {o[0][k], o[1][k], o[0][k + 1], o[1][k + 1]};
      *reinterpret_cast<f4a8*>(out + k) = q;
    }
  } else {
    for (int k = 0; k < MED_T && x0 + k < w; ++k) out[k] = make_float2(o[0][k], o[1][k]);
  }
}

template <int BX>
static void launch_row8(hipStream_t st, const float2* src, float2* dst, int w, int h, size_t bs, int B) {
  typedef MedGeom<BX> G;
  dim3 grd((w + BX * MED_T - 1) / (BX * MED_T), (h + G::BY - 1) / G::BY, B);
  hipLaunchKernelGGL(k_median5_c2_row8<BX>, grd, dim3(256), 0, st, src, dst, w, h, bs);
}
// tiles of a launch, the narrow shapes slightly dearer per tile. The weights are a hand-set tie-breaker that picks the fastest
// shape on the five shapes of tools/median_microbench (profiles/r06_v11_median_microbench.txt), not a fit.
static double median_tile_cost(int w, int h, int bx) {
  const int tw = bx * MED_T, th = 256 / bx;
  const double tiles = double((w + tw - 1) / tw) * double((h + th - 1) / th);
  return tiles * (bx >= 16 ? 1.0 : bx == 8 ? 1.05 : 1.1);
}
// S360_MEDIAN_BX=32 / 16 / 8 / 4 forces the tile shape (tuning, tests; the results do not depend on it)
static int median_forced_bx() {
  static const int v = [] {
    const char* e = std::getenv("S360_MEDIAN_BX");
    const int t = e ? std::atoi(e) : 0;
    return t == 32 || t == 16 || t == 8 || t == 4 ? t : 0;
  }();
  return v;
}

void launch_median5_c2(hipStream_t st, const float2* src, float2* dst, int w, int h, size_t bs, int B) {
  dim3 blk(64, 4);
  if (w >= 64) {  // 8 pixels per thread; narrow levels keep one pixel per thread (more threads than the chip otherwise idles)
    int bx = median_forced_bx();
    if (!bx) {
      bx = 32;
      for (int c : {16, 8, 4})
        if (median_tile_cost(w, h, c) < median_tile_cost(w, h, bx)) bx = c;
    }
    switch (bx) {
      case 32: launch_row8<32>(st, src, dst, w, h, bs, B); break;
      case 16: launch_row8<16>(st, src, dst, w, h, bs, B); break;
      case 8: launch_row8<8>(st, src, dst, w, h, bs, B); break;
      default: launch_row8<4>(st, src, dst, w, h, bs, B); break;
    }
    return;
  }
  hipLaunchKernelGGL(k_median5_c2, grid2d(w, h, B, blk), blk, 0, st, src, dst, w, h, bs);
}

}  // namespace s360
