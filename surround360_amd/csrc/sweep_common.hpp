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
