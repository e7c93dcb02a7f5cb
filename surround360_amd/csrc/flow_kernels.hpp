// flow_kernels.hpp — launchers of the PixFlow HIP kernels (flow_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include <vector>

namespace s360 {

struct BlurTaps {  // centre tap k[0] and the r symmetric taps k[1..r]
  int r;
  float k[8];
};
// Which image planes a flow of the batch uses: flow b matches image i0[b] (I0) against i1[b] (I1).
// The 14 side pairs need only 28 image pyramids for 28 flows (LtoR and RtoL share them). Device arrays of B ints
// (a batch can hold the flows of many frames).
constexpr int kMaxFlows = 2048;  // (a limit of the tables only: the batch is the z dimension of the launches)
struct FlowIdx {
  const int* i0;
  const int* i1;
};
struct PixFlowConsts {  // OpticalFlowFactory.h:26-41 / :45-60
  float pyrScaleFactor, smoothnessCoef, verticalRegularizationCoef, horizontalRegularizationCoef;
  float gradientStepSize, downscaleFactor;
  int maxPercentage;
};

// src_tab (optional): device array of B source pointers used instead of src + sbs * b (images of a batch that do not
// live in one allocation); likewise dst_tab / src_tab of the other launchers that take one.
void launch_resize_cubic_u8c4(hipStream_t st, const uchar4* src, int sw, int sh, size_t sbs, uchar4* dst, int dw,
                              int dh, size_t dbs, int B, const uchar4* const* src_tab = nullptr);
void launch_gray_alpha(hipStream_t st, const uchar4* src, size_t n, size_t sbs, float* gray, float* alpha, size_t pbs,
                       int B);
void launch_motion(hipStream_t st, const uchar4* cur, const uchar4* prev, size_t n, size_t sbs, float* motion,
                   size_t pbs, int B);
void launch_sepblur(hipStream_t st, const float* src, float* dst, int w, int h, int cn, size_t bs, int B,
                    const BlurTaps& t, float* const* dst_tab = nullptr);
void launch_diffusion(hipStream_t st, const float2* flow, float2* dst, int w, int h, size_t bs, int B,
                      const BlurTaps& t, const float* A, const FlowIdx& idx);
void launch_diffusion_adjust(hipStream_t st, const float2* flow, float2* dst, int w, int h, size_t bs, int B, const BlurTaps& t,
                             const float* A, const FlowIdx& idx, const float2* prev, const float* motion, float prev_scale);
void launch_upscale_blur(hipStream_t st, const float2* src, int sw, int sh, size_t sbs, float2* dst, int dw, int dh,
                         size_t dbs, int B, float post_scale, const BlurTaps& t, float* const* dst_tab = nullptr);
void launch_gradients(hipStream_t st, const float* I, float2* G, int w, int h, size_t bs, int B, const BlurTaps& t);
// G == nullptr: half-records (float2, throughput sweep kernel); G given: full records (float4, latency sweep kernel)
void launch_blur_to_records(hipStream_t st, const float2* flow, void* rec, int w, int h, size_t bs, int B,
                            const BlurTaps& t, const float2* G, const float* A, const FlowIdx& idx,
                            unsigned* rowflags = nullptr);
void launch_resize_linear_f32(hipStream_t st, const float* src, int sw, int sh, size_t sbs, float* dst, int dw, int dh,
                              size_t dbs, int cn, int B, float post_scale, int do_scale);
void launch_resize_cubic_f32c2(hipStream_t st, const float2* src, int sw, int sh, size_t sbs, float2* dst, int dw,
                               int dh, size_t dbs, int B, float post_scale, const float2* const* src_tab = nullptr);
void launch_median5_c2(hipStream_t st, const float2* src, float2* dst, int w, int h, size_t bs, int B);
// lockstep banded sweep (sweep_lock.hip): nw compute waves (4 rows each) + 2 service waves per workgroup
int sweep_lock_waves();  // compute waves per workgroup: 2 (tools/sweep_microbench builds: S360_LOCK_NW = 1 / 2 / 4)
int sweep_lock_num_wgs(int h, int nw);
size_t sweep_lock_handoff_bytes(int w, int h, int B, int nw);
// rec: per flow and pixel {I0x | NaN = not updated, I0y, blurredFlow} (launch_blur_to_records with G); G: the gradient planes
// per IMAGE — a flow's taps sample plane idx.i1[b]
void launch_sweep_lock(hipStream_t st, const float4* rec, const float2* G, float2* flow, void* handoff,
                       unsigned* errflag, int w, int h, size_t bs, int B, const FlowIdx& idx, int dir,
                       const PixFlowConsts& pc, bool fast);
// true when the kernel's fast exact division may be used for all of these divisors (checked on the device, cached)
bool sweep_verify_divisors(hipStream_t st, const std::vector<float>& divisors);
// throughput-oriented sweep (sweep_quad.hip): one wave per workgroup, 16 rows x 4 or 20 rows x 3 lanes per pixel, two rounds
size_t sweep_quad_handoff_bytes(int w, int h, int B);
// rec: per flow and pixel the HALF-record {blurredFlow.x | NaN = not updated, blurredFlow.y} (launch_blur_to_records without G);
// a pixel's record is completed with I0's gradient from plane idx.i0[b] of G
void launch_sweep_quad(hipStream_t st, const float2* rec, const float2* G, float2* flow, void* handoff,
                       unsigned* errflag, int w, int h, size_t bs, int B, const FlowIdx& idx, int dir,
                       const PixFlowConsts& pc, bool fast, const unsigned* rowflags = nullptr);
void launch_search_init(hipStream_t st, const float* I, const float* A, int w, int h, size_t pbs, int B,
                        const FlowIdx& idx, float2* flow, int hint, int dist, float* I1eq);

}  // namespace s360
