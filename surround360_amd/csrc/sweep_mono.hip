// sweep_mono.hip — PixFlow propagation sweep for gfx950 with ONE lane per pixel ("mono"): the throughput kernel.
//
// Same recurrence and same results as sweep_lock.hip / sweep_quad.hip (PixFlow.h:388-410). With the chip full of
// sweeps the limit is instruction issue, so what counts is instructions per pixel update:
//   lockstep kernel: 16 lanes per pixel, 9 speculative evaluations in one round          (lowest latency of ONE flow)
//   quad kernel:      4 lanes per pixel, 5 evaluations in two rounds, ~250 instr / 16 px  (~16 per pixel)
//   this kernel:      1 lane per pixel, the reference's 5 evaluations one after the other, ~340 instr / 64 px (~5 per pixel)
// A wave owns a band of 64 consecutive rows; lane r handles column s - r at step s (skewed), so inside the wave the
// raster-order dependencies are register hand-offs: left neighbour = the lane's own previous result, up neighbour =
// lane r-1's previous result (DPP row_shr:1, row_bcast:15 across the 16-lane DPP rows). Bands of one flow are
// separate one-wave workgroups; band k+1 takes the results of band k's last row from 8-byte {fx,fy} granules in global
// memory (all-ones = not written: the data is the flag; bands are ticketed in band-major order, so a band's
// predecessor has always started; every spin is bounded and sets an error flag). What the kernel gives up is latency:
// a step is five dependent gather rounds deep, and a band starts 64 steps after the band above. It therefore needs
// many flows in flight (several frames, or the 28 side flows of one frame) to fill the chip — FlowEngine uses it in
// throughput mode only.
// Everything that is not the pixel update is amortised with wave-uniform control, as in the quad kernel: the band
// above is polled every kMNeed steps (a poll returns up to 64 granules), results go through an LDS ring and are written
// back once per kMChunk steps as 64-byte row segments (loads and stores retire in order through one counter on gfx950:
// a global store per step would sit in front of every gather), the last row's granules are published every kMPub
// steps, and operands outside the proven range of the fast division / square root re-run the update with the IEEE
// expansions (one branch per step).
#include <cstdlib>
#include <type_traits>

#include "devmath.hpp"
#include "sweep_common.hpp"

namespace s360 {

namespace {

constexpr unsigned long long kEmptyGranuleM = 0xFFFFFFFFFFFFFFFFull;
constexpr int kMRows = 64;    // rows per wave
constexpr int kMUpRing = 64;  // columns of the band above kept in LDS
constexpr int kMChunk = 8;    // steps per write-back
constexpr int kMResRing = 16; // result columns per row kept in LDS
constexpr int kMResStride = 18;  // float2 per row: lane r writes column (s - r) & 15 -> bank 2r + const: conflict-free per 16 lanes
constexpr int kMNeed = 4;     // row 0 checks the band above every kMNeed steps
constexpr int kMPub = 4;      // the last row publishes its granules every kMPub steps

// previous result of the row above = lane - 1: row_shr:1 inside a 16-lane DPP row, lane 15 of the previous DPP row for
// lanes 16 / 32 / 48 (row_bcast:15), lane 0 keeps `old` (the granule-fed value of the band above).
__device__ __forceinline__ float from_lane_above(float old, float v) {
  int r = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), 0x142, 0xE, 0x1, false);
  r = __builtin_amdgcn_update_dpp(r, __builtin_bit_cast(int, v), 0x111, 0xF, 0xF, false);
  return __builtin_bit_cast(float, r);
}

}  // namespace

template <bool FAST>
__global__ __launch_bounds__(64) void k_sweep_mono(const float4* __restrict__ rec, const float2* __restrict__ G,
                                                   float2* __restrict__ flow, unsigned long long* __restrict__ H,
                                                   unsigned* __restrict__ hdr, int w, int h, size_t bs, FlowIdx idx,
                                                   int dir, SweepConst c, SweepFast fc, int nb, int B,
                                                   unsigned* __restrict__ errflag) {
  __shared__ float2 s_up[kMUpRing];
  __shared__ float2 s_res[kMRows * kMResStride];
  __shared__ unsigned s_ticket;
  const int lane = threadIdx.x;
  if (lane == 0) s_ticket = atomicAdd(hdr, 1u) + 1u;  // the counter starts at 0xFFFFFFFF (all-ones arena)
  __syncthreads();
  const unsigned tk = s_ticket;
  const int band = (int)(tk / (unsigned)B), b = (int)(tk - (unsigned)band * (unsigned)B);
  if (band >= nb) return;
  const float2* __restrict__ G1 = G + bs * idx.i1[b];
  const char* __restrict__ G1b0 = reinterpret_cast<const char*>(G1);
  const char* __restrict__ G1b1 = reinterpret_cast<const char*>(G1 + w);
  rec += bs * b;
  flow += bs * b;
  H += (size_t)b * nb * w;
  const unsigned long long* Hin = H + (size_t)band * w;
  unsigned long long* Hout = H + (size_t)(band + 1) * w;
  const int r = lane;
  const int yi = band * kMRows + r;
  const bool rowValid = yi < h;
  const int yic = rowValid ? yi : h - 1;
  const int y = dir > 0 ? yic : h - 1 - yic;
  const bool hasUp = yi > 0;
  const bool hasUpBand = band > 0;
  const bool publishes = band + 1 < nb;
  const int lastRow = min(kMRows, h - band * kMRows) - 1;  // last valid row of this band (what the band below reads)
  const float4* __restrict__ recRow = rec + (size_t)y * w;
  float2* __restrict__ flowRow = flow + (size_t)y * w;
  const float fy = (float)y;
  const float kEps = 0.001f, kInf = __int_as_float(0x7f800000);
  const int nsteps = w + kMRows - 1;
  auto col = [&](int xi) { const int xc = min(max(xi, 0), w - 1); return dir > 0 ? xc : w - 1 - xc; };

  // errorFunction at (x + ax, y + ay) for this lane's pixel (PixFlow.h:493-534)
  auto evaluate = [&](auto ieee, int x, float4 rc, float ax, float ay, bool& tiny) -> float {
    const float mx = __builtin_amdgcn_fmed3f((float)x + ax, 0.0f, c.wm2);
    const float my = __builtin_amdgcn_fmed3f(fy + ay, 0.0f, c.hm2);
    const int x0 = (int)mx, y0 = (int)my;
    const float xR = __builtin_amdgcn_fractf(mx), yR = __builtin_amdgcn_fractf(my);
    unsigned boff = (unsigned)(__umul24(y0, w) + x0) << 3;
    if (S360_DBG(fc, 1)) boff = (unsigned)lane << 4;  // (developer tools only) gathers that always hit
    const f4a8 ta = *reinterpret_cast<const f4a8*>(G1b0 + boff);
    const f4a8 tb = *reinterpret_cast<const f4a8*>(G1b1 + boff);
    Texels tt;
    tt.r0 = make_float4(ta.x, ta.y, ta.z, ta.w);
    tt.r1 = make_float4(tb.x, tb.y, tb.z, tb.w);
    if (decltype(ieee)::value) {
      Foot ft;
      ft.off = 0; ft.xR = xR; ft.yR = yR;
      return error_from(tt, ft, rc.x, rc.y, rc.z, rc.w, ax, ay, c);
    }
    bool t1;
    const float e = error_fast(tt, xR, yR, rc.x, rc.y, rc.z, rc.w, ax, ay, c, fc, t1);
    tiny = tiny || t1;
    return e;
  };
  // One pixel update exactly as the reference writes it (PixFlow.h:390-397 / 403-410): current error, the left and the
  // up proposal (proposeFlowUpdate :415-435), then the finite-difference gradient step on the winner (errorGradient
  // :195-217). All five evaluations always run (no divergence); unavailable proposals lose by +inf.
  auto update = [&](auto ieee, int x, int xi, float4 rc, float2 fo, float2 fl, float2 up, bool& tiny) -> float2 {
    float2 f = fo;
    float cur = evaluate(ieee, x, rc, fo.x + 0.0f, fo.y + 0.0f, tiny);
    float e1 = evaluate(ieee, x, rc, fl.x + 0.0f, fl.y + 0.0f, tiny);
    float e2 = evaluate(ieee, x, rc, up.x + 0.0f, up.y + 0.0f, tiny);
    if (!(xi > 0)) e1 = kInf;  // no left proposal in the first column
    if (!hasUp) e2 = kInf;     // no up proposal in the first row
    if (e1 < cur) { f = fl; cur = e1; }
    if (e2 < cur) { f = up; cur = e2; }
    const float ex = evaluate(ieee, x, rc, f.x + kEps, f.y + 0.0f, tiny);
    const float ey = evaluate(ieee, x, rc, f.x + 0.0f, f.y + kEps, tiny);
    const float nx = ex - cur, ny = ey - cur;
    float ggx, ggy;
    if (decltype(ieee)::value) {
      ggx = nx / kEps;
      ggy = ny / kEps;
    } else {
      ggx = fdiv_m(nx, kEps, fc.rcEps);
      ggy = fdiv_m(ny, kEps, fc.rcEps);
      tiny = tiny || min(tiny_key(fabsf(nx)), tiny_key(fabsf(ny))) < kTinyBits - 1u;
    }
    float2 res;
    res.x = f.x - c.gradStep * ggx;
    res.y = f.y - c.gradStep * ggy;
    return res;
  };

  // ---- granules of the band above -> s_up ring. Wave-uniform state; columns [.., upFilled) have been taken ----
  int upFilled = hasUpBand ? 0 : 0x3fffffff;
  bool pending = false, dead = S360_DBG(fc, 2) != 0;
  unsigned long long pv = kEmptyGranuleM;
  auto issue = [&]() {
    const int xi = upFilled + lane;
    pv = kEmptyGranuleM;
    if (xi < w) pv = __hip_atomic_load(Hin + xi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    pending = true;
  };
  auto process = [&](int limit) {  // takes the leading run of written granules, never beyond column `limit`
    const int xi = upFilled + lane;
    const unsigned long long bad = __ballot(xi >= w || (pv == kEmptyGranuleM && !dead));
    int n = bad ? (int)__ffsll((long long)bad) - 1 : 64;
    n = min(n, limit - upFilled);
    if (lane < n)
      s_up[xi & (kMUpRing - 1)] = make_float2(__uint_as_float((unsigned)pv), __uint_as_float((unsigned)(pv >> 32)));
    upFilled = __builtin_amdgcn_readfirstlane(upFilled + max(n, 0));
    pending = false;
  };

  float2 fl = make_float2(0.f, 0.f);  // result of the previous pixel of this row
  float4 nrc;
  float2 nfo;
  {
    const int x0c = col(0 - r);
    nrc = recRow[x0c];
    nfo = flowRow[x0c];
  }
  for (int s0 = 0; s0 < nsteps; s0 += kMChunk) {
    const int send = min(s0 + kMChunk, nsteps);
    for (int s = s0; s < send; ++s) {
      if (hasUpBand && (s & (kMNeed - 1)) == 0 && s < w) {  // row 0 needs columns [s, s + kMNeed) of the band above
        const int need = min(s + kMNeed, w), limit = s + kMUpRing;
        if (pending) process(limit);
        unsigned spins = 0;
        while (upFilled < need) {
          if (spins) __builtin_amdgcn_s_sleep(2);
          issue();
          process(limit);
          if (++spins > (1u << 20) ||
              ((spins & 255u) == 0 && __hip_atomic_load(errflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            dead = true;  // the band above is gone: stop waiting, flag the result invalid, keep draining
            if (lane == 0) atomicExch(errflag, 1u);
          }
        }
        if (upFilled < w && upFilled - s < 2 * kMNeed + 8) issue();  // running low: taken at the next check
      }
      const float4 rc = nrc;
      const float2 fo = nfo;
      {  // inputs of the next step, one step ahead
        const int xn = col(s + 1 - r);
        nrc = recRow[xn];
        nfo = flowRow[xn];
      }
      const float2 upl = s_up[s & (kMUpRing - 1)];
      const int xi = s - r;
      const bool active = rowValid && xi >= 0 && xi < w;
      const int x = dir > 0 ? xi : w - 1 - xi;  // unclamped: out-of-range columns are inactive, their gathers are clamped
      const bool upd = rc.x == rc.x;
      float2 up;
      up.x = from_lane_above(upl.x, fl.x);
      up.y = from_lane_above(upl.y, fl.y);
      float2 res;
      if (FAST) {
        bool tiny = false;
        res = update(std::false_type{}, x, xi, rc, fo, fl, up, tiny);
        if (__builtin_expect(__ballot(tiny) != 0ull, 0)) res = update(std::true_type{}, x, xi, rc, fo, fl, up, tiny);
      } else {
        bool tiny = false;
        res = update(std::true_type{}, x, xi, rc, fo, fl, up, tiny);
      }
      const bool take = active && upd;
      const float2 alt = active ? fo : fl;
      res.x = take ? res.x : alt.x;
      res.y = take ? res.y : alt.y;
      fl = res;
      s_res[r * kMResStride + (xi & (kMResRing - 1))] = res;
      if (publishes && ((s & (kMPub - 1)) == kMPub - 1 || s == nsteps - 1)) {  // the last row's granules for the band below
        const int xi0 = (s & ~(kMPub - 1)) - lastRow + lane;
        if (lane < kMPub && xi0 >= 0 && xi0 < w && xi0 <= s - lastRow) {
          const float2 v = s_res[lastRow * kMResStride + (xi0 & (kMResRing - 1))];
          __hip_atomic_store(Hout + xi0, ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    // ---- write-back of the chunk: row rr produced columns [s0 - rr, send - rr); 8 rows x 8 columns per instruction ----
    if (!S360_DBG(fc, 4)) {
#pragma unroll
      for (int k = 0; k < kMRows / 8; ++k) {
        const int rr = 8 * k + (lane >> 3);
        const int xi = s0 - rr + (lane & 7);
        const int yy = band * kMRows + rr;
        if (yy < h && xi >= 0 && xi < w && xi < send - rr) {
          const int gy = dir > 0 ? yy : h - 1 - yy;
          flow[(size_t)gy * w + (dir > 0 ? xi : w - 1 - xi)] = s_res[rr * kMResStride + (xi & (kMResRing - 1))];
        }
      }
    }
  }
}

// ==========================================================================================
int sweep_mono_num_bands(int h) { return (h + kMRows - 1) / kMRows; }
size_t sweep_mono_handoff_bytes(int w, int h, int B) {
  return 256 + (size_t)B * sweep_mono_num_bands(h) * w * sizeof(unsigned long long);
}
void launch_sweep_mono(hipStream_t st, const float4* rec, const float2* G, float2* flow, void* handoff,
                       unsigned* errflag, int w, int h, size_t bs, int B, const FlowIdx& idx, int dir,
                       const PixFlowConsts& pc, bool fast) {
  const SweepConst c = make_sweep_const(pc, w, h);
  SweepFast fc;
  fc.rcCols = 1.0f / c.fcols;
  fc.rcRows = 1.0f / c.frows;
  fc.rcEps = 1.0f / 0.001f;
  fc.dbg = S360_DBG_FROM_ENV();  // developer tools only
  const int nb = sweep_mono_num_bands(h);
  // `handoff` must be all-ones (ticket counter in the first 256 bytes, then the granules): FlowEngine resets the
  // hand-off arena of all its sweep launches with one memset.
  unsigned* hdr = reinterpret_cast<unsigned*>(handoff);
  unsigned long long* H = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(handoff) + 256);
  if (fast)
    hipLaunchKernelGGL((k_sweep_mono<true>), dim3(nb * B), dim3(64), 0, st, rec, G, flow, H, hdr, w, h, bs, idx, dir, c,
                       fc, nb, B, errflag);
  else
    hipLaunchKernelGGL((k_sweep_mono<false>), dim3(nb * B), dim3(64), 0, st, rec, G, flow, H, hdr, w, h, bs, idx, dir, c,
                       fc, nb, B, errflag);
}

}  // namespace s360
