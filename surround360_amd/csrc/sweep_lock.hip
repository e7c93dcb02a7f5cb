// sweep_lock.hip — the production PixFlow propagation sweep for gfx950: "lockstep" banded wavefront.
//
// Reference: PixFlow.h:388-410 (the forward / backward raster sweeps of patchMatchPropagationAndSearch), with
// errorFunction (:493-534), proposeFlowUpdate (:415-435) and errorGradient (:195-217). The raster order is a
// Gauss-Seidel recurrence — pixel (x,y) reads the already-updated flow at (x-1,y) and (x,y-1) — so the only
// parallel order that reproduces it bit-for-bit is an anti-diagonal wavefront. The kernel is bound by the
// LATENCY of one pixel update (its dependent instruction chain), not by HBM or ALU throughput; everything here
// is arranged to keep that chain short.
//
// Geometry. A workgroup owns R = 4*NW consecutive rows of one flow: NW compute waves x 4 rows, 16 lanes per
// pixel. Row r of wave j handles column s - r at the wave's local step s; wave j runs kLag steps behind wave
// j-1. The whole workgroup advances one step per s_barrier ("lockstep"), so every intra-workgroup dependency
// is either a register/DPP hand-off (left neighbour = the row's own previous result; up neighbour = the row
// above's previous result, DPP row_bcast:15) or an LDS slot written at least two barriers earlier — no
// counters, no polling between compute waves. Workgroups of one flow are chained through 8-byte {fx,fy}
// granules in global memory (one per column of the band's last row; all-ones = "not written yet": the data is
// the flag, MI355X_MICROARCH.md hand-off price list). Bands take (band, flow) from a ticket counter in
// band-major order, so a band's predecessor has always started: no co-residency assumption, every spin bounded.
//
// Speculation. A pixel update needs 5 errorFunction evaluations in two dependent rounds (current / left / up
// proposals, then the two finite-difference probes of the winner). The 16 lanes of a pixel evaluate all 9
// possible ones at once (bank 0: current flow, bank 1: left, bank 2: up; lane 0/1/2 of a bank: f, f+(eps,0),
// f+(0,eps)), exchange the 9 scalars with DPP row broadcasts and replay the reference's sequential selection.
// One step is therefore ONE gather round + ONE evaluation deep.
//
// Service waves. On gfx950 a wave's loads and stores retire in order through one counter (vmcnt), so a compute
// wave that also streamed its inputs/outputs would wait for HBM round trips every step. Two extra waves per
// workgroup own that traffic: wave NW ("bulk") streams the per-pixel records {I0x|NaN mask, I0y, blurredFlow}
// and the current flow into an LDS ring 16-32 steps ahead, writes results back, and touches the I1-gradient
// lines the coming pixels will sample so that the compute waves' bilinear gathers hit L1; wave NW+1 ("hand-off")
// polls the granules of the band above and publishes the band's last row. The compute waves' memory queue
// only ever holds their own gathers.
//
// Measured and NOT adopted (round 4; profiles/r04_v1_lock_window_*): the LDS window of I1-gradient texels that the
// throughput kernel uses (sweep_quad.hip), here filled per chunk by the bulk service wave, the taps as two ds_read2_b64 kept
// in flight across the barrier (s_waitcnt lgkmcnt(2)), global gathers as the per-step fallback. Bit-exact, and on the same
// box slower than this kernel: 8K frame 136.2 against 133.3 ms (sweeps 120.2 / 117.2), stream 111.5 against 108.9 ms per
// frame, micro-benchmark within 1 % on the large levels and 4-8 % slower on the small ones. The gathers it replaces already hit
// L1 — the bulk wave touches the lines the coming pixels will sample — and their latency sits behind the barrier; what the
// window adds (placement reductions and six loads per chunk in the service wave that shares a SIMD with a compute wave,
// the window test in front of every tap) costs more than the ~100 cycles of L1-against-LDS latency it saves. The "290
// cycles waiting for gathers" of round 3's instrumented build were an artefact of its timestamps (s_memtime returns
// through the same counter as the LDS reads).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <type_traits>
#include <vector>

#include "devmath.hpp"
#include "sweep_common.hpp"

namespace s360 {

namespace {

#ifdef S360_SWEEP_TIMING  // tools/sweep_microbench only: per-phase cycle counts of compute wave 0 of ticket 0 -> hdr[8..]
#define TS_DECL unsigned long long ts_last = __builtin_amdgcn_s_memtime(), ts_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define TS(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ts_acc[i] += t_ - ts_last; ts_last = t_; } while (0)
#define TS_WAITV() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define TS_DUMP() do { if (tk == 0 && j == 0 && lane == 0) for (int q = 0; q < 8; ++q) ((unsigned long long*)(hdr + 8))[q] = ts_acc[q]; } while (0)
#else
#define TS_DECL
#define TS(i)
#define TS_WAITV()
#define TS_DUMP()
#endif

constexpr unsigned long long kEmptyGranule = 0xFFFFFFFFFFFFFFFFull;
constexpr int kLag = 5;     // steps between consecutive compute waves of a workgroup (>= 5: see preload below)
constexpr int kRingK = 32;  // LDS ring depth in steps (two chunks)
constexpr int kChunk = 16;  // steps per bulk-service transfer (16 steps x 4 rows = 64 lanes)

struct __attribute__((aligned(16))) LkIn {
  float4 rec;   // {I0x (NaN: pixel not updated), I0y, blurredFlow.x, blurredFlow.y}
  float2 flow;  // the flow before this sweep touches the pixel
  float2 pad;
};

typedef float f4n __attribute__((ext_vector_type(4)));
typedef float f2n __attribute__((ext_vector_type(2)));

// One barrier per step. Hand-written so that no vmcnt wait is attached (the service waves keep loads in flight
// across steps; __syncthreads()/the s_barrier builtin would drain them on gfx9-class targets).
#ifdef S360_WAVE_EMULATION
__device__ __forceinline__ void wg_barrier() { __syncthreads(); }
#define WG_BARRIER_AFTER(a, b, c, d, e, f, g, h) __syncthreads()
#else
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// ... with values that must have been COMPUTED before the barrier is reached: the texel-independent part of the step is meant to
// run while the gathers are in flight, and without this the compiler sinks it into the block that uses it — behind the wait
#define WG_BARRIER_AFTER(a, b, c, d, e, f, g, h) \
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : : "memory")
#endif

template <int K>
__device__ __forceinline__ float row_bcast(float v) {  // value of lane K of this lane's 16-lane row
  const int i = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, 0x150 + K, 0xF, 0xF, true));
}
// lane 15 of the previous 16-lane row, written only to rows 1..3 and only to the banks in BANKS; others keep old
template <int BANKS>
__device__ __forceinline__ float from_row_above(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v),
                                                              0x142, 0xE, BANKS, false));
}

}  // namespace

__global__ __launch_bounds__(256) void k_verify_div(const float* __restrict__ cs, unsigned* __restrict__ bad) {
  const float cc = cs[blockIdx.y];
  const float rc = 1.0f / cc;
  const float x = __uint_as_float(0x3f800000u | (blockIdx.x * 256u + threadIdx.x));
  if (x / cc != fdiv_m(x, cc, rc) || (-x) / cc != fdiv_m(-x, cc, rc)) atomicOr(bad + blockIdx.y, 1u);
}

template <int NW, bool FAST>
__global__ __launch_bounds__((NW + 2) * 64) void k_sweep_lock(const float4* __restrict__ rec,
                                                              const float2* __restrict__ G, float2* __restrict__ flow,
                                                              unsigned long long* __restrict__ H,
                                                              unsigned* __restrict__ hdr, int w, int h, size_t bs,
                                                              FlowIdx idx, int dir, SweepConst c, SweepFast fc,
                                                              int nwg, int B, unsigned* __restrict__ errflag) {
  constexpr int R = NW * 4;
  __shared__ LkIn s_in[NW][kRingK][4];
  __shared__ float2 s_out[NW][kRingK][4];
  __shared__ float2 s_up0[kRingK];
  __shared__ unsigned s_ticket;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  if (threadIdx.x == 0) s_ticket = atomicAdd(hdr, 1u) + 1u;  // the counter starts at 0xFFFFFFFF (memset 0xFF)
  __syncthreads();
  const unsigned tk = s_ticket;
  const int wgband = (int)(tk / (unsigned)B), b = (int)(tk - (unsigned)wgband * (unsigned)B);
  if (wgband >= nwg) return;
  const float2* __restrict__ G1 = G + bs * idx.i1[b];
  rec += bs * b;
  flow += bs * b;
  H += (size_t)b * nwg * w;
  const int rows0 = wgband * R;
  const int nsteps = w + 3;                // local steps of one compute wave (row skew 0..3)
  const int T = nsteps + kLag * (NW - 1);  // global steps of the workgroup
  auto col = [&](int xi) { const int xc = min(max(xi, 0), w - 1); return dir > 0 ? xc : w - 1 - xc; };

  if (wave < NW) {
    // ------------------------------------ compute waves ------------------------------------
    const int j = wave;
    if (!S360_DBG(fc, 128)) __builtin_amdgcn_s_setprio(3);  // the update chain outranks the service waves on its SIMD
    const int rr = lane >> 4, k = lane & 15, bank = k >> 2, role = k & 3;
    const int yi = rows0 + j * 4 + rr;
    const bool rowValid = yi < h;
    const int yic = rowValid ? yi : h - 1;
    const int y = dir > 0 ? yic : h - 1 - yic;
    const bool hasUp = yi > 0;
    const float kEps = 0.001f;
    const float ox = role == 1 ? kEps : 0.0f, oy = role == 2 ? kEps : 0.0f;
    const float fy = (float)y;
    const bool fromLds = bank == 0 || (bank == 2 && rr == 0);  // candidate is the old flow / the granule-fed up value
    const float kInf = __int_as_float(0x7f800000);
    const char* __restrict__ G1b0 = reinterpret_cast<const char*>(G1);
    const char* __restrict__ G1b1 = reinterpret_cast<const char*>(G1 + w);
    float2 fl = make_float2(0.f, 0.f);  // final flow of the previous pixel of this row (same in all 16 lanes)
    LkIn nin;
    nin.rec = make_float4(0.f, 0.f, 0.f, 0.f);
    nin.flow = make_float2(0.f, 0.f);
    float2 nup = make_float2(0.f, 0.f);
    TS_DECL;
    // One iteration = one step. The barrier sits in the shadow of the bilinear gathers: everything a step needs
    // from other waves was read from LDS one iteration earlier (after that iteration's barrier), so the wave
    // starts its dependent chain without waiting for anybody.
    // STEADY: the steps at which all four rows of the wave are inside the image with a left neighbour
    // (local steps 4 .. w-1). There the range tests, the first-column case and the per-wave choice of the LDS source
    // of the up value are compile-time facts; the same code, ~40 instructions per step shorter.
    const float2* upBase = j == 0 ? &s_up0[0] : &s_out[j > 0 ? j - 1 : 0][0][3];
    const int upStride = j == 0 ? 1 : 4, upShift = j == 0 ? 0 : 3;
    const int xLane = dir > 0 ? -rr : w - 1 + rr, xSign = dir > 0 ? 1 : -1;
    auto step = [&](auto steady, int t) {
      constexpr bool ST = decltype(steady)::value;
      TS(0);
      const int s = t - kLag * j;
      const bool run = ST || (s >= 0 && s < nsteps && !S360_DBG(fc, 16));
      const LkIn in = nin;
      const float2 upl = nup;
      const int xi = s - rr;
      const bool active = ST ? rowValid : (rowValid && xi >= 0 && xi < w);
      const int x = ST ? xLane + s * xSign : (dir > 0 ? xi : w - 1 - xi);  // unclamped: out-of-range columns are inactive, their gathers are clamped
      const float4 rc = in.rec;
      const float2 fo = in.flow;
      const bool upd = rc.x == rc.x;
      // Pixels below the alpha threshold keep their flow (PixFlow.h:390 / :403). When none of the wave's 4 pixels is
      // updated at this step (the upper ~60 % of the pole flows, which the side cameras do not cover) the gathers
      // and the evaluation are skipped; workgroups made of such rows run at the speed of the empty iteration.
      const bool take = active && upd;
      const bool any = run && __ballot(take) != 0ull;
      // (deliberately not initialised: set and read only on the `any` path; zero-filling them cost 14 moves per step)
      float upx, upy, altx, alty;
      float ax, ay, xR, yR;
      f4a8 ta, tb;
      float preS, preV, preH;
      unsigned preK;
      if (any) {
        // this lane's candidate: bank 0 current flow, bank 1 left result, bank 2 up result
        float2 cand;
        cand.x = fromLds ? (bank == 0 ? fo.x : upl.x) : fl.x;
        cand.y = fromLds ? (bank == 0 ? fo.y : upl.y) : fl.y;
        cand.x = from_row_above<0x4>(cand.x, fl.x);
        cand.y = from_row_above<0x4>(cand.y, fl.y);
        ax = cand.x + ox;
        ay = cand.y + oy;
        // Every lane evaluates (idle lanes and masked pixels produce values nobody reads; addresses are clamped).
        // getPixBilinear32FExtend's clamp (PixFlow.h:457-464): max(0, .) then min(., size-2) == med3 here; the
        // operands are non-negative, so (int) truncation == floor and x - float(int(x)) == fract(x) exactly.
        const float mx = __builtin_amdgcn_fmed3f((float)x + ax, 0.0f, c.wm2);
        const float my = __builtin_amdgcn_fmed3f(fy + ay, 0.0f, c.hm2);
        const int x0 = (int)mx, y0 = (int)my;
        xR = __builtin_amdgcn_fractf(mx);
        yR = __builtin_amdgcn_fractf(my);
        const unsigned boff = (unsigned)(__umul24(y0, w) + x0) << 3;
        if (!S360_DBG(fc, 64)) {  // (timing experiment: 64 = no gathers)
          ta = *reinterpret_cast<const f4a8*>(G1b0 + boff);
          tb = *reinterpret_cast<const f4a8*>(G1b1 + boff);
        } else {
          ta.x = xR; ta.y = yR; tb.z = mx; tb.w = my;
        }
        // ---- in the gathers' shadow: everything of this step that does not need the texels ----
        // up neighbour in every lane (needed by the selection below)
        upx = from_row_above<0xF>(upl.x, fl.x);
        upy = from_row_above<0xF>(upl.y, fl.y);
        // not-updated pixels keep their flow; inactive lanes keep the previous result
        altx = active ? fo.x : fl.x;
        alty = active ? fo.y : fl.y;
        if (FAST) {
          const ErrPre p = error_fast_pre(rc.z, rc.w, ax, ay, c, fc);
          preS = p.smTerm; preV = p.vTerm; preH = p.hTerm; preK = p.key;
        }
      }
      TS(1);
      if (FAST && any) { WG_BARRIER_AFTER(preS, preV, preH, preK, upx, upy, altx, alty); }
      else wg_barrier();
      TS(2);
      // Preload the LDS operands of step s+1. The up value of row 0 is the row-3 result of wave j-1 at column
      // s+1, i.e. its local step s+4 = global step t-1 (kLag = 5): written before the barrier just passed.
      const int s1 = s + 1;
      if (ST) {
        nin = s_in[j][s1 & (kRingK - 1)][rr];
        nup = upBase[((s1 + upShift) & (kRingK - 1)) * upStride];
      } else if (s1 >= 0 && s1 < nsteps) {
        nin = s_in[j][s1 & (kRingK - 1)][rr];
        nup = (j == 0) ? s_up0[s1 & (kRingK - 1)] : s_out[j > 0 ? j - 1 : 0][(s1 + 3) & (kRingK - 1)][3];
      }
#ifdef S360_SWEEP_TIMING
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      TS(6);
#endif
      if (!run) return;
      if (!any) {
        const float2 keep = active ? fo : fl;
        fl = keep;
        if (active && k == 0) s_out[j][s & (kRingK - 1)][rr] = keep;
        return;
      }
      TS_WAITV();
      TS(3);
      Texels tt;
      tt.r0 = make_float4(ta.x, ta.y, ta.z, ta.w);
      tt.r1 = make_float4(tb.x, tb.y, tb.z, tb.w);
      float e;
      if (FAST) {
        bool tiny;
        ErrPre pre;
        pre.smTerm = preS; pre.vTerm = preV; pre.hTerm = preH; pre.key = preK;
        e = error_fast_post(tt, xR, yR, rc.x, rc.y, pre, tiny);
        if (__builtin_expect(__ballot(tiny) != 0ull, 0)) {
          Foot ft;
          ft.off = 0; ft.xR = xR; ft.yR = yR;
          e = error_from(tt, ft, rc.x, rc.y, rc.z, rc.w, ax, ay, c);
        }
      } else {
        Foot ft;
        ft.off = 0; ft.xR = xR; ft.yR = yR;
        e = error_from(tt, ft, rc.x, rc.y, rc.z, rc.w, ax, ay, c);
      }
      TS(4);
      const float e0 = row_bcast<0>(e), e0x = row_bcast<1>(e), e0y = row_bcast<2>(e);
      float e1 = row_bcast<4>(e);
      const float e1x = row_bcast<5>(e), e1y = row_bcast<6>(e);
      float e2 = row_bcast<8>(e);
      const float e2x = row_bcast<9>(e), e2y = row_bcast<10>(e);
      if (!ST && !(xi > 0)) e1 = kInf;  // no left proposal in the first column (PixFlow.h:392 / :405)
      if (!hasUp) e2 = kInf;            // no up proposal in the first row (:393 / :406)
      // proposeFlowUpdate x2 in the reference's order, then the gradient step on the winner
      float2 f = fo;
      float cur = e0, ex = e0x, ey = e0y;
      if (e1 < cur) { f = fl; cur = e1; ex = e1x; ey = e1y; }
      if (e2 < cur) { f = make_float2(upx, upy); cur = e2; ex = e2x; ey = e2y; }
      const float nx = ex - cur, ny = ey - cur;
      float ggx, ggy;
      if (FAST) {
        ggx = fdiv_m(nx, kEps, fc.rcEps);
        ggy = fdiv_m(ny, kEps, fc.rcEps);
        const bool tiny = min(tiny_key(fabsf(nx)), tiny_key(fabsf(ny))) < kTinyBits - 1u;
        if (__builtin_expect(__ballot(tiny) != 0ull, 0)) {
          ggx = nx / kEps;
          ggy = ny / kEps;
        }
      } else {
        ggx = nx / kEps;
        ggy = ny / kEps;
      }
      float2 res;
      res.x = f.x - c.gradStep * ggx;
      res.y = f.y - c.gradStep * ggy;
      res.x = take ? res.x : altx;
      res.y = take ? res.y : alty;
      fl = res;
      if (active && k == 0) s_out[j][s & (kRingK - 1)][rr] = res;
      TS(5);
    };
    const int tA = min(max(kLag * j + 4, -1), T), tB = min(max(kLag * j + w, tA), T);  // steady: local steps [4, w)
    for (int t = -1; t < tA; ++t) step(std::false_type{}, t);
    int t2 = tA;
    for (; t2 + 1 < tB; t2 += 2) {
      step(std::true_type{}, t2);
      step(std::true_type{}, t2 + 1);
    }
    if (t2 < tB) step(std::true_type{}, t2);
    for (int t = tB; t < T; ++t) step(std::false_type{}, t);
    TS_DUMP();
    wg_barrier();
    return;
  }

  if (wave >= NW && S360_DBG(fc, 32)) return;  // timing experiment: compute waves alone
  if (wave == NW) {
    // ------------------------------------ bulk service wave ------------------------------------
    // One event per compute wave every 16 steps (staggered by kLag). An event first consumes what was issued at
    // earlier events (>= kLag steps old: no stall), then issues new loads/stores and returns without waiting.
    const int st = lane & 15, rr = lane >> 4;
    const f4n* __restrict__ recN = reinterpret_cast<const f4n*>(rec);
    const f2n* __restrict__ flowN = reinterpret_cast<const f2n*>(flow);
    const unsigned* __restrict__ G1w = reinterpret_cast<const unsigned*>(G1);
    int rowOff[NW];
    bool rowOk[NW];
    float rowY[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int yi = rows0 + j * 4 + rr;
      rowOk[j] = yi < h;
      const int yic = rowOk[j] ? yi : h - 1;
      const int y = dir > 0 ? yic : h - 1 - yic;
      rowOff[j] = y * w;
      rowY[j] = (float)y;
    }
    f4n rRec[NW];
    f2n rFlow[NW];
    unsigned sink = 0, pf0 = 0, pf1 = 0, pf2 = 0, pf3 = 0;
    const int nchunks = (nsteps + kChunk - 1) / kChunk;
    auto load_chunk = [&](int j, int cidx) {
      const int x = col(cidx * kChunk + st - rr);
      rRec[j] = __builtin_nontemporal_load(recN + rowOff[j] + x);
      rFlow[j] = __builtin_nontemporal_load(flowN + rowOff[j] + x);
    };
    auto write_chunk = [&](int j, int cidx) {
      LkIn v;
      v.rec = make_float4(rRec[j].x, rRec[j].y, rRec[j].z, rRec[j].w);
      v.flow = make_float2(rFlow[j].x, rFlow[j].y);
      v.pad = make_float2(0.f, 0.f);
      s_in[j][(cidx * kChunk + st) & (kRingK - 1)][rr] = v;
    };
    auto flush_chunk = [&](int j, int cidx) {
      const int sidx = cidx * kChunk + st;
      const int xi = sidx - rr;
      if (rowOk[j] && xi >= 0 && xi < w && sidx < nsteps) flow[rowOff[j] + col(xi)] = s_out[j][sidx & (kRingK - 1)][rr];
    };
    // The service work of compute wave j entering chunk cc (chunk cc-1 is complete and its ring slots are free) is
    // spread over four consecutive steps, so that no step's share outlasts the compute waves' own step: this wave
    // shares a SIMD with a compute wave at lower priority, and the whole workgroup waits for it at every barrier
    // (one 150-instruction event per chunk cost ~0.9 us on the steps it fell on, ~15 % of the sweep).
    //   phase 0: chunk cc+1 (in registers since the previous event) -> LDS ring, and where its pixels will sample
    //            I1's gradients (predicted by the blurred flow and by the current flow)
    //   phase 1: chunk cc+2 -> registers            phase 2: results of chunk cc-1 -> global memory
    //   phase 3: touch the predicted gradient lines, so that the compute waves' gathers find them in L1
    int po[NW][4];
#pragma unroll
    for (int j = 0; j < NW; ++j) po[j][0] = po[j][1] = po[j][2] = po[j][3] = 0;
    auto phase = [&](int j, int cc, int ph) {
      if (S360_DBG(fc, 8)) return;
      const bool wr = cc + 1 < nchunks;
      if (ph == 0) {
        if (wr) {
          const float xf = (float)col((cc + 1) * kChunk + st - rr);
          const Foot fa = footprint(w, xf + rRec[j].z, rowY[j] + rRec[j].w, c);
          const Foot fb = footprint(w, xf + rFlow[j].x, rowY[j] + rFlow[j].y, c);
          po[j][0] = 2 * fa.off; po[j][1] = 2 * (fa.off + w); po[j][2] = 2 * fb.off; po[j][3] = 2 * (fb.off + w);
          write_chunk(j, cc + 1);
        }
      } else if (ph == 1) {
        if (cc + 2 < nchunks) load_chunk(j, cc + 2);
      } else if (ph == 2) {
        if (cc >= 1) flush_chunk(j, cc - 1);
      } else {
        sink ^= pf0 ^ pf1 ^ pf2 ^ pf3;  // previous round (long since landed)
        if (wr && !S360_DBG(fc, 4)) {
          pf0 = G1w[po[j][0]]; pf1 = G1w[po[j][1]]; pf2 = G1w[po[j][2]]; pf3 = G1w[po[j][3]];
        }
      }
    };
    // prologue: chunk 0 into LDS, chunk 1 into registers
#pragma unroll
    for (int j = 0; j < NW; ++j) load_chunk(j, 0);
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      phase(j, -1, 0);
      phase(j, -1, 1);
      phase(j, -1, 3);
    }
    for (int t = -1; t < T; ++t) {
      wg_barrier();
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        const int s = t - kLag * j;
        const int ph = (s - 1) & (kChunk - 1);
        if (s >= 1 && ph < 4) phase(j, (s - 1) >> 4, ph);
      }
    }
    wg_barrier();
    // epilogue: the chunks the loop did not reach (chunk cc-1 is flushed at local step 16 cc + 3)
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int sl = T - 1 - kLag * j;
      const int done = sl >= 3 ? ((sl - 3) >> 4) : 0;  // chunks [0, done) were flushed in the loop
      for (int cidx = done; cidx < nchunks; ++cidx) flush_chunk(j, cidx);
    }
    sink ^= pf0 ^ pf1 ^ pf2 ^ pf3;
    if (sink == 0x9e3779b9u && lane == 0) hdr[1] = sink;  // keeps the prefetch loads alive; never true in practice
    return;
  }

  // ------------------------------------ hand-off service wave ------------------------------------
  // Every step: (1) take in the granule poll issued two steps ago, (2) block only if the band above has fallen
  // behind what the next step needs, (3) publish the finished columns of this band's last row, (4) issue the next
  // poll. Everything waited on is two steps old.
  {
    const bool hasUpWg = wgband > 0 && !S360_DBG(fc, 1);
    const bool publishes = wgband + 1 < nwg && !S360_DBG(fc, 2);
    const unsigned long long* Hin = H + (size_t)wgband * w;
    unsigned long long* Hout = H + (size_t)(wgband + 1) * w;
    constexpr int jl = NW - 1;
    int upFilled = hasUpWg ? 0 : 0x3fffffff, pub = 0;
    bool pending = false, dead = false;
    unsigned long long pv = kEmptyGranule;
    auto issue = [&]() {
      const int xi = upFilled + lane;
      pv = kEmptyGranule;
      if (lane < 16 && xi < w) pv = __hip_atomic_load(Hin + xi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pending = true;
    };
    auto process = [&](int limit) {  // leading run of written granules -> s_up0; upFilled never exceeds limit
      const int xi = upFilled + lane;
      const unsigned long long bad = __ballot(lane < 16 && (xi >= w || (pv == kEmptyGranule && !dead)));
      int n = bad ? (int)__ffsll((long long)bad) - 1 : 16;
      n = min(n, limit - upFilled);
      if (n > 0) {
        if (lane < n)
          s_up0[xi & (kRingK - 1)] = make_float2(__uint_as_float((unsigned)pv), __uint_as_float((unsigned)(pv >> 32)));
        upFilled = __builtin_amdgcn_readfirstlane(upFilled + n);
      }
      pending = false;
    };
    auto ensure = [&](int need, int limit) {  // block (bounded) until columns [0, need) are in LDS
      unsigned spins = 0;
      while (upFilled < need) {
        if (!pending) issue();
        process(limit);
        if (upFilled < need) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > (1u << 21) ||
              ((spins & 1023u) == 0 && __hip_atomic_load(errflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            // the band above is gone: stop waiting, flag the result invalid, keep the pipeline draining
            dead = true;
            if (lane == 0) atomicExch(errflag, 1u);
          }
        }
      }
    };
    auto publish = [&](int xdone) {
      const int avail = min(xdone, w - 1) - pub + 1;
      if (avail > 0) {
        const int n = min(16, avail);
        if (lane < n) {
          const int xi = pub + lane;
          const float2 v = s_out[jl][(xi + 3) & (kRingK - 1)][3];
          __hip_atomic_store(Hout + xi, ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        pub += n;
      }
    };
    // The band above is needed only where the workgroup's first row is updated (pixels below the alpha threshold
    // keep their flow): column c is waited for iff its record is not NaN-masked, and columns the first row passed
    // without needing them are dropped. Workgroups whose first row is never updated — most of a pole flow — wait
    // for nobody.
    int pendT = -100;
    bool wantPoll = false;
    if (hasUpWg) {
      const int y0 = dir > 0 ? rows0 : h - 1 - rows0;
      const float4* __restrict__ rec0 = rec + (size_t)y0 * w;
      const float m0 = rec0[col(0)].x, m1 = rec0[col(1)].x;  // (the LDS ring is not filled yet)
      if (m0 == m0 || m1 == m1) {
        ensure(min(w, 2), 26);
        if (upFilled < w) issue();
      }
    }
    for (int t = -1; t < T; ++t) {
      wg_barrier();
      // By the end of this iteration the compute waves need columns <= t+2; the ring allows columns <= t+32.
      // Order per iteration: take in the poll issued two iterations ago, publish, issue the next poll — so that
      // whatever the next wait covers is two steps old.
      if (hasUpWg && upFilled < w) {
        const int c = t + 2;  // the column that becomes necessary now; its record has been in the ring for >= 13 steps
        wantPoll = false;
        if (c < w) {
          const float m = s_in[0][c & (kRingK - 1)][0].rec.x;
          wantPoll = m == m;
        }
        if (wantPoll) {
          if (upFilled < c) {
            upFilled = c;
            pending = false;
          }
          const int need = c + 1;
          if (pending && (t - pendT >= 2 || upFilled < need)) process(t + 31);
          if (upFilled < need) ensure(need, t + 31);
        }
      }
      // columns of the band's last row that are complete and visible: wave jl finished local step t-1-kLag*jl
      if (publishes) publish(t - 1 - kLag * jl - 3);
      if (hasUpWg && upFilled < w && !pending && wantPoll) {
        issue();
        pendT = t;
      }
    }
    wg_barrier();
    if (publishes)
      while (pub < w) publish(w - 1);
  }
}

// ==========================================================================================
int sweep_lock_num_wgs(int h, int nw) { return (h + 4 * nw - 1) / (4 * nw); }
size_t sweep_lock_handoff_bytes(int w, int h, int B, int nw) {
  return 256 + (size_t)B * sweep_lock_num_wgs(h, nw) * w * sizeof(unsigned long long);
}

// Device check of fdiv_m for a set of divisors: every significand, both signs. Results are cached per process.
bool sweep_verify_divisors(hipStream_t st, const std::vector<float>& cs) {
  static std::mutex mu;
  static std::map<unsigned, bool> known;
  std::lock_guard<std::mutex> lk(mu);
  std::vector<float> todo;
  for (float v : cs) {
    unsigned bits;
    std::memcpy(&bits, &v, 4);
    if (!known.count(bits) && std::find(todo.begin(), todo.end(), v) == todo.end()) todo.push_back(v);
  }
#ifdef S360_WAVE_EMULATION  // (tools/hip_wave_shim: the same check as k_verify_div as a host loop, not as 8 M emulated lanes)
  for (float cc : todo) {
    const float rc = 1.0f / cc;
    bool ok = true;
    for (unsigned m = 0; m < (1u << 23) && ok; ++m) {
      const float x = __uint_as_float(0x3f800000u | m);
      ok = x / cc == fdiv_m(x, cc, rc) && (-x) / cc == fdiv_m(-x, cc, rc);
    }
    unsigned bits;
    std::memcpy(&bits, &cc, 4);
    known[bits] = ok;
  }
  todo.clear();
#endif
  if (!todo.empty()) {
    float* dc = nullptr;
    unsigned* dbad = nullptr;
    std::vector<unsigned> bad(todo.size(), 1u);
    if (hipMalloc(&dc, todo.size() * sizeof(float)) == hipSuccess &&
        hipMalloc(&dbad, todo.size() * sizeof(unsigned)) == hipSuccess) {
      (void)hipMemcpyAsync(dc, todo.data(), todo.size() * sizeof(float), hipMemcpyHostToDevice, st);
      (void)hipMemsetAsync(dbad, 0, todo.size() * sizeof(unsigned), st);
      hipLaunchKernelGGL(k_verify_div, dim3((1u << 23) / 256, (unsigned)todo.size()), dim3(256), 0, st, dc, dbad);
      if (hipMemcpyAsync(bad.data(), dbad, todo.size() * sizeof(unsigned), hipMemcpyDeviceToHost, st) != hipSuccess ||
          hipStreamSynchronize(st) != hipSuccess)
        std::fill(bad.begin(), bad.end(), 1u);
    }
    if (dc) (void)hipFree(dc);
    if (dbad) (void)hipFree(dbad);
    for (size_t i = 0; i < todo.size(); ++i) {
      unsigned bits;
      std::memcpy(&bits, &todo[i], 4);
      known[bits] = bad[i] == 0;
    }
  }
  for (float v : cs) {
    unsigned bits;
    std::memcpy(&bits, &v, 4);
    if (!known[bits]) return false;
  }
  return true;
}

template <int NW, bool FAST>
static void launch_lock_t(hipStream_t st, const float4* rec, const float2* G, float2* flow, unsigned long long* H,
                          unsigned* hdr, unsigned* errflag, int w, int h, size_t bs, int B, const FlowIdx& idx, int dir,
                          const SweepConst& c, const SweepFast& fc, int nwg) {
  hipLaunchKernelGGL((k_sweep_lock<NW, FAST>), dim3(nwg * B), dim3((NW + 2) * 64), 0, st, rec, G, flow, H, hdr, w,
                     h, bs, idx, dir, c, fc, nwg, B, errflag);
}

// Compute waves per workgroup (4 rows each): 2 — one compute wave per SIMD next to the two service waves, bands of 8
// rows. (Measured on an 8K frame, round 2's bench record: 4 waves 119.2 ms of sweeps per frame, 2 waves with the
// specialised steady-state steps 117.0 ms, 8 waves 186 ms; the other builds are gone.)
#ifdef S360_TIMING_EXPERIMENTS  // tools/sweep_microbench only: S360_LOCK_NW = 1 / 2 / 4 compute waves per workgroup (bands of 4 / 8 / 16 rows)
static int lock_waves_experiment() {
  static const int v = [] { const char* e = std::getenv("S360_LOCK_NW"); const int n = e ? std::atoi(e) : 2; return (n == 1 || n == 4) ? n : 2; }();
  return v;
}
int sweep_lock_waves() { return lock_waves_experiment(); }
#else
int sweep_lock_waves() { return 2; }
#endif

void launch_sweep_lock(hipStream_t st, const float4* rec, const float2* G, float2* flow, void* handoff,
                       unsigned* errflag, int w, int h, size_t bs, int B, const FlowIdx& idx, int dir,
                       const PixFlowConsts& pc, bool fast) {
  const int nw = sweep_lock_waves();
  const SweepConst c = make_sweep_const(pc, w, h);
  SweepFast fc;
  fc.rcCols = 1.0f / c.fcols;
  fc.rcRows = 1.0f / c.frows;
  fc.rcEps = 1.0f / 0.001f;
  fc.dbg = S360_DBG_FROM_ENV();  // developer tools only
  const int nwg = sweep_lock_num_wgs(h, nw);
  // `handoff` must be all-ones: ticket counter (first 256 bytes) and every granule start as "not written"
  // (FlowEngine resets the hand-off arena of all its sweep launches with one memset)
  unsigned* hdr = reinterpret_cast<unsigned*>(handoff);
  unsigned long long* H = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(handoff) + 256);
#ifdef S360_TIMING_EXPERIMENTS
  if (nw == 1) { launch_lock_t<1, true>(st, rec, G, flow, H, hdr, errflag, w, h, bs, B, idx, dir, c, fc, nwg); return; }
  if (nw == 4) { launch_lock_t<4, true>(st, rec, G, flow, H, hdr, errflag, w, h, bs, B, idx, dir, c, fc, nwg); return; }
#endif
  if (fast) launch_lock_t<2, true>(st, rec, G, flow, H, hdr, errflag, w, h, bs, B, idx, dir, c, fc, nwg);
  else launch_lock_t<2, false>(st, rec, G, flow, H, hdr, errflag, w, h, bs, B, idx, dir, c, fc, nwg);
}

}  // namespace s360
