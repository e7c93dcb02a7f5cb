// devmath.hpp — scalar helpers shared by the HIP kernels and by the host-side table
// builders of libs360 (gfx950 only; compiled with -ffp-contract=off so that every float
// expression rounds exactly like the reference's SSE2 build, CMakeLists.txt:33-35).
//
// The integer/rounding rules below are the OpenCV 3.1 semantics the reference inherits
// (SURVEY.md Appendix A): cvRound = round-half-even with INT_MIN on overflow,
// saturate_cast, 1/32-pixel remap coordinates, A=-0.75 bicubic.
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>
#include <stdint.h>

#define S360_HD __host__ __device__ __forceinline__

// Wave-synchronous LDS hand-overs — one lane writes, another lane of the same wave reads, no barrier in between — are
// ordered by the hardware (a wave's LDS accesses execute in program order for all its lanes). The CPU emulation the
// tests run the kernels under (tools/hip_wave_shim) runs the lanes of a wave one after the other between two cross-lane
// operations and needs those points marked; in the product build the mark is nothing at all.
#ifdef S360_WAVE_EMULATION
#define S360_WAVE_SYNC() emu::wave_sync()
#define S360_VM_DRAIN()
#else
#define S360_WAVE_SYNC()
// s_waitcnt vmcnt(0) as a real machine instruction (gfx9 encoding: vmcnt 0, expcnt 7, lgkmcnt 15): waits for every
// outstanding global load / store of the wave AND tells the compiler's wait-count pass that nothing is pending behind
// this point. Placed on COLD paths in front of a hot loop: the pass merges the pending-load state of all static
// predecessors of a block, and a load that can only be pending on a cold path otherwise costs a wait in every iteration
// of the hot one (wherever the hot loop reuses that load's register).
#define S360_VM_DRAIN() __builtin_amdgcn_s_waitcnt(0x0F70)
#endif

namespace s360 {

// XCD-aware tile order (cdna_hip_programming.md T1): the dispatcher places workgroup b on XCD b % 8, each XCD has its own
// L2, so with the plain order the neighbours of a tile — whose halo / source box overlaps its own — run on seven other
// L2s and every overlap is fetched again through the fabric. Here the grid's linear workgroup index is re-dealt so that
// each XCD works through ONE contiguous run of tiles (x fastest, then y, then z): neighbours meet in the same L2.
// Bijective for any grid size; a pure re-ordering of which workgroup renders which tile (results cannot depend on it).
struct TileId { unsigned x, y, z; };
__device__ __forceinline__ TileId xcd_tile() {
  const unsigned gx = gridDim.x, gy = gridDim.y, T = gx * gy * gridDim.z;
  const unsigned L = (blockIdx.z * gy + blockIdx.y) * gx + blockIdx.x;
  unsigned P = L;
  if (T >= 64) {
    const unsigned k = L & 7u, n = T >> 3, rem = T & 7u;  // XCD k serves n (+1 for k < rem) consecutive tiles
    P = k * n + (k < rem ? k : rem) + (L >> 3);
  }
  TileId t;
  t.x = P % gx;
  const unsigned q = P / gx;
  t.y = q % gy;
  t.z = q / gy;
  return t;
}

S360_HD int cv_round(float v) {
  if (!(v >= -2147483648.0f && v < 2147483648.0f)) return INT_MIN;
  return (int)__builtin_rintf(v);
}
S360_HD int cv_floor(float v) {
  int i = (int)v;
  return i - (v < (float)i);
}
S360_HD int sat_u8(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }
S360_HD int sat_s16(int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; }
// implicit float -> uchar of the reference's Vec4b(float, ...) (NovelView.cpp:144-148): truncate.
S360_HD int trunc_u8(float v) {
  int i = (int)v;
  return i < 0 ? 0 : i > 255 ? 255 : i;
}
S360_HD int clip_idx(int x, int n) { return x >= 0 ? (x < n ? x : n - 1) : 0; }
S360_HD int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
  return p;
}

// OpenCV interpolateCubic, A = -0.75 (imgproc resize/remap).
S360_HD void cubic_coeffs(float x, float* c) {
  const float A = -0.75f;
  c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
  c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
  c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
  c[3] = 1.f - c[0] - c[1] - c[2];
}
// resize source coordinate: f = (float)((d+0.5)*scale-0.5); s = floor(f); f -= s.
S360_HD void resize_coord(int d, double scale, int* s, float* f) {
  const float fx = (float)((d + 0.5) * scale - 0.5);
  const int sx = cv_floor(fx);
  *s = sx;
  *f = fx - sx;
}
// remap: float map entry -> first-tap integer coordinate and 10-bit fraction index.
S360_HD void remap_coord(float mx, float my, int* sx, int* sy, int* fxy) {
  const int ix = cv_round(mx * 32.f), iy = cv_round(my * 32.f);
  *fxy = (iy & 31) * 32 + (ix & 31);
  *sx = sat_s16(ix >> 5) - 1;
  *sy = sat_s16(iy >> 5) - 1;
}
// MathUtil.h:29-31, :52-59
S360_HD float rampf(float x, float a, float b) {
  const float t = (x - a) / (b - a);
  const float m = (t < 1.0f) ? t : 1.0f;  // std::min(1.0f, t)
  return (0.0f < m) ? m : 0.0f;           // std::max(0.0f, m)
}
S360_HD float lerpf(float x0, float x1, float a) { return x0 * (1.0f - a) + x1 * a; }

}  // namespace s360
