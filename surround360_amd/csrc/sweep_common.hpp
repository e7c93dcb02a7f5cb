// sweep_common.hpp — device helpers shared by the PixFlow sweep kernels (flow_kernels.hip, sweep_lock.hip):
// errorFunction (PixFlow.h:493-534, no directional term) and getPixBilinear32FExtend (:457-475) in the
// reference's float operation order (no FMA contraction).
#pragma once
#include <hip/hip_runtime.h>

#include "flow_kernels.hpp"

namespace s360 {

struct SweepConst {
  float smoothnessCoef, vertCoef, horizCoef, gradStep;
  float fcols, frows;  // float(I0.cols), float(I0.rows)
  float wm2, hm2;      // w - 2.0f, h - 2.0f
};

typedef float f4a8 __attribute__((ext_vector_type(4), aligned(8)));
struct Texels { float4 r0, r1; };  // (x0,y0),(x0+1,y0) and (x0,y0+1),(x0+1,y0+1) as (Ix,Iy,Ix,Iy)
struct Foot { int off; float xR, yR; };
// getPixBilinear32FExtend's clamp + split (PixFlow.h:457-464)
__device__ __forceinline__ Foot footprint(int w, float x, float y, const SweepConst& c) {
  x = (0.0f < x) ? x : 0.0f;
  x = (x < c.wm2) ? x : c.wm2;
  y = (0.0f < y) ? y : 0.0f;
  y = (y < c.hm2) ? y : c.hm2;
  const int x0 = (int)x, y0 = (int)y;
  Foot f;
  f.off = y0 * w + x0;
  f.xR = x - (float)x0;
  f.yR = y - (float)y0;
  return f;
}
// errorFunction (PixFlow.h:493-534) on already-gathered texels
__device__ __forceinline__ float error_from(const Texels& t, const Foot& ft, float g0x, float g0y, float bfx,
                                            float bfy, float fdx, float fdy, const SweepConst& c) {
  float i1x, i1y;
  {
    const float a1 = t.r0.x, a2 = t.r0.z - t.r0.x, a3 = t.r1.x - t.r0.x, a4 = t.r0.x + t.r1.z - t.r0.z - t.r1.x;
    i1x = a1 + a2 * ft.xR + a3 * ft.yR + a4 * ft.xR * ft.yR;
  }
  {
    const float a1 = t.r0.y, a2 = t.r0.w - t.r0.y, a3 = t.r1.y - t.r0.y, a4 = t.r0.y + t.r1.w - t.r0.w - t.r1.y;
    i1y = a1 + a2 * ft.xR + a3 * ft.yR + a4 * ft.xR * ft.yR;
  }
  const float dfx = bfx - fdx, dfy = bfy - fdy;
  const float smoothness = sqrtf(dfx * dfx + dfy * dfy);
  const float ex = g0x - i1x, ey = g0y - i1y;
  return sqrtf(ex * ex + ey * ey) + smoothness * c.smoothnessCoef + c.vertCoef * fabsf(fdy) / c.fcols +
         c.horizCoef * fabsf(fdx) / c.frows;
}
// Correctly rounded x / c for a divisor known in advance (Markstein's sequence): q = RN(x*rc), r = x - q*c (exact
// in one FMA), q' = RN(q + r*rc), with rc = RN(1/c). For the (c, rc) pairs this kernel is launched with, the
// result has been checked on the device against the IEEE division for every float significand
// (k_verify_div below); that covers every x whose intermediates stay normal, i.e. x == 0 or |x| >= 2^-96 —
// smaller non-zero numerators take the IEEE path (see `tiny`).
__device__ __forceinline__ float fdiv_m(float x, float c, float rc) {
  const float q = x * rc;
  const float r = __builtin_fmaf(-q, c, x);
  return __builtin_fmaf(r, rc, q);
}
// Correctly rounded sqrtf for x == 0 or x >= 2^-96: v_sqrt_f32 (1 ulp) + the two-sided residual fix-up that the
// compiler's own IEEE expansion uses, without its denormal pre-scaling.
__device__ __forceinline__ float sqrt_cr(float x) {
  float s = __builtin_amdgcn_sqrtf(x);
  const float sm = __int_as_float(__float_as_int(s) - 1), sp = __int_as_float(__float_as_int(s) + 1);
  const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
  s = (rm <= 0.0f) ? sm : s;
  s = (rp > 0.0f) ? sp : s;
  return s;
}
// key for the "tiny non-zero" test of a non-negative float: bits - 1 (0 wraps to 0xFFFFFFFF, NaN/inf are large)
__device__ __forceinline__ unsigned tiny_key(float v) { return __float_as_uint(v) - 1u; }
constexpr unsigned kTinyBits = 0x0F800000u;  // 2^-96

struct SweepFast {  // reciprocals of the three divisors of the sweep, verified on the device
  float rcCols, rcRows, rcEps;
  int dbg;  // always 0 in the product library (see S360_DBG)
};
// Timing experiments that invalidate the results (skip polls / publishes / gathers ...) exist only in the developer
// tools: tools/Makefile builds the sweep sources with -DS360_TIMING_EXPERIMENTS, the product library never does, and
// there S360_DBG() is the constant 0 — no environment variable can switch a result-changing path on.
#ifdef S360_TIMING_EXPERIMENTS
#define S360_DBG(fc, bits) ((fc).dbg & (bits))
#define S360_DBG_FROM_ENV() (std::getenv("S360_SWEEP_DBG") ? std::atoi(std::getenv("S360_SWEEP_DBG")) : 0)
#else
#define S360_DBG(fc, bits) 0
#define S360_DBG_FROM_ENV() 0
#endif

// errorFunction (PixFlow.h:493-534) with the verified fast divisions / square roots, in two parts (round 5): what does not
// depend on I1's texels — the smoothness term sqrt(|blurredFlow - flow|^2) * coef and the two regularisation terms: a square root
// with its fix-up and two divisions, ~30 instructions — and what does. The latency sweep kernel evaluates the first part between
// the issue of its bilinear gathers and the data's arrival (it used to sit behind the wait), the same operations on the same
// operands in the same order: same bits. Sets tinyFlag when an operand falls outside the proven range of the fast division /
// square root (the caller then re-evaluates with the IEEE expansion).
struct ErrPre { float smTerm, vTerm, hTerm; unsigned key; };
__device__ __forceinline__ ErrPre error_fast_pre(float bfx, float bfy, float fdx, float fdy, const SweepConst& c, const SweepFast& fc) {
  const float dfx = bfx - fdx, dfy = bfy - fdy;
  const float sm2 = dfx * dfx + dfy * dfy;
  const float smoothness = sqrt_cr(sm2);
  const float vn = c.vertCoef * fabsf(fdy), hn = c.horizCoef * fabsf(fdx);
  ErrPre p;
  p.smTerm = smoothness * c.smoothnessCoef;
  p.vTerm = fdiv_m(vn, c.fcols, fc.rcCols);
  p.hTerm = fdiv_m(hn, c.frows, fc.rcRows);
  p.key = min(tiny_key(sm2), min(tiny_key(vn), tiny_key(hn)));
  return p;
}
__device__ __forceinline__ float error_fast_post(const Texels& t, float xR, float yR, float g0x, float g0y, const ErrPre& p, bool& tinyFlag) {
  float i1x, i1y;
  {
    const float a1 = t.r0.x, a2 = t.r0.z - t.r0.x, a3 = t.r1.x - t.r0.x, a4 = t.r0.x + t.r1.z - t.r0.z - t.r1.x;
    i1x = a1 + a2 * xR + a3 * yR + a4 * xR * yR;
  }
  {
    const float a1 = t.r0.y, a2 = t.r0.w - t.r0.y, a3 = t.r1.y - t.r0.y, a4 = t.r0.y + t.r1.w - t.r0.w - t.r1.y;
    i1y = a1 + a2 * xR + a3 * yR + a4 * xR * yR;
  }
  const float ex = g0x - i1x, ey = g0y - i1y;
  const float d2 = ex * ex + ey * ey;
  tinyFlag = min(p.key, tiny_key(d2)) < kTinyBits - 1u;
  return sqrt_cr(d2) + p.smTerm + p.vTerm + p.hTerm;
}
__device__ __forceinline__ float error_fast(const Texels& t, float xR, float yR, float g0x, float g0y, float bfx,
                                            float bfy, float fdx, float fdy, const SweepConst& c, const SweepFast& fc,
                                            bool& tinyFlag) {
  return error_fast_post(t, xR, yR, g0x, g0y, error_fast_pre(bfx, bfy, fdx, fdy, c, fc), tinyFlag);
}

inline SweepConst make_sweep_const(const PixFlowConsts& pc, int w, int h) {
  SweepConst c;
  c.smoothnessCoef = pc.smoothnessCoef;
  c.vertCoef = pc.verticalRegularizationCoef;
  c.horizCoef = pc.horizontalRegularizationCoef;
  c.gradStep = pc.gradientStepSize;
  c.fcols = (float)w;
  c.frows = (float)h;
  c.wm2 = (float)w - 2.0f;
  c.hm2 = (float)h - 2.0f;
  return c;
}

}  // namespace s360
