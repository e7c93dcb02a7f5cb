// sweep_quad.hip — throughput-oriented PixFlow propagation sweep for gfx950 ("quad").
//
// Same recurrence and same results as sweep_lock.hip (PixFlow.h:388-410), arranged for many flows / frames in
// flight instead of for the latency of one flow. A wave's step is a latency chain (~280 mostly dependent instructions
// at ~6 cycles each); with the chip full of sweeps a SIMD interleaves two of them (the staging registers and LDS of a
// wave allow two waves per SIMD) and reaches ~0.33 of the ~0.4 instructions per cycle it sustains (tools/issue_rate),
// so this variant minimises instructions per STEP (DESIGN.md section 5 has the measurements, including the mappings
// with fewer instructions per pixel that were not faster):
//   * 4 lanes per pixel, 16 rows per wave (row r handles column s - r at step s);
//   * the reference's two dependent rounds are kept: round 1 evaluates the current / left / up proposals in lanes
//     0..2 of the quad, round 2 the two finite-difference probes of the winner in lanes 0..1 — 5 evaluations instead
//     of the lockstep kernel's 9;
//   * the step itself is ~280 instructions; everything else is amortised: the band above is checked every 4 steps
//     with wave-uniform control (a poll returns up to 64 granules), results go to an LDS ring and are written back once
//     per 16 steps, the last row's granules are published every 4 steps, and the rare operands outside the proven range
//     of the fast division / square root re-run the whole update with the IEEE expansions (one branch per step instead
//     of three);
//   * the steady step contains no wait on the memory counter (loads and stores retire in order through one counter on
//     gfx950): the next chunk's records / flows / window are requested so that the checks of the band above — the only
//     consumers of a load inside a chunk — sit three steps behind them, and nothing that is pending on a cold path can
//     look pending to the steady loops (ONE call site of win_issue, S360_VM_DRAIN behind the edge chunks: the
//     compiler's wait-count pass otherwise drains the counter in every step, which is how the build profiled as
//     r03_v8 ran);
//   * no service waves, no barriers: a workgroup is ONE wave; other resident waves cover its memory latency. Left
//     neighbour = registers, up neighbour = DPP (row_shr:4 / row_bcast:15), a band's first row takes the last row of
//     the band above from 8-byte {fx,fy} granules in global memory (all-ones = not written; bands are ticketed in
//     band-major order, spins are bounded);
//   * waves are persistent (capped grid, a finished band takes the next ticket);
//   * the records and incoming flows of a 16-step chunk are fetched once (four pixels per lane) while the previous chunk
//     computes and staged in LDS; all-interior chunks run a step with the range tests folded away;
//   * the I1-gradient texels the chunk's bilinear taps can touch are staged in LDS as well — a window of
//     kWinRows x kWinCols texels placed around the cells the chunk's own incoming flows point at, filled with coalesced
//     row segments once per chunk — so that the two dependent gather rounds of a step are LDS reads (~100 cycles)
//     instead of L1/L2 gathers (500+). A wave any of whose relevant taps leaves the window gathers from global memory
//     for that round: same texels, same bits. (Measured against the previous build — global gathers in round 1, the
//     probes of round 2 taking the winner's texels by ds_bpermute — on 12-slot 8K batches: 16.3 vs 17.4 ms of sweeps per
//     frame; profiles/r03_v1_*);
//   * pixels below the alpha threshold are not updated (PixFlow.h:390): a step none of whose 16 pixels is updated
//     skips both rounds, a band waits for the band above only where its first row is updated, and a band without any
//     updated pixel (per-row flags written by the record kernel) hands its last row on and leaves — 63 % of a pole
//     flow's pixels are like that.
// Measured (tools/sweep_microbench tp1): 32.8 Gpx/s on saturated side levels (336 flows of 607x884, 3 lanes per pixel),
// 37.0 Gpx/s on the 48 pole flows of a 12-frame batch (4 lanes per pixel); one frame's flows alone run as fast as with
// sweep_lock.hip on the large levels and 5-30 % slower on the small ones (profiles/r03_v9_*). FlowEngine picks this kernel
// in throughput mode (s360_set_sweep_mode).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "devmath.hpp"
#include "sweep_common.hpp"

namespace s360 {

namespace {

constexpr unsigned long long kEmptyGranuleQ = 0xFFFFFFFFFFFFFFFFull;
// Lanes per pixel (LPP). 4: a pixel is a DPP quad, 16 rows per wave; round 1 uses 3 of the 4 lanes, round 2 two: 5
// evaluations in 8 lane slots. 3: 5 evaluations in 6 slots, 20 rows per wave — a 16-lane DPP row holds 5 pixels, lanes
// {0,1,2} {3,4,5} {6,7,8} {9,10,11} {12,13,14}; lane 15 is a passive copy of the last pixel's result, so that row_bcast:15
// can hand it to the first pixel of the next DPP row; the values of a round are exchanged with row_shr / row_shl by 1 and
// 2 and a select per role (a quad broadcast does not exist for groups of three): ~16 more instructions per step for 25 %
// more pixels per step, and bands of 20 rows (20 % fewer bands and hand-offs per flow).
constexpr int quad_rows(int lpp) { return lpp == 3 ? 20 : 16; }
constexpr int kUpRing = 64;  // columns of the band above kept in LDS

template <int K>
__device__ __forceinline__ float quad_bcast(float v) {  // value of lane K of this lane's quad
  const int i = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, K | (K << 2) | (K << 4) | (K << 6), 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ float dpp_row(float v) {  // row_shr:n (0x110 + n) / row_shl:n (0x100 + n); lanes without a source keep v
  const int i = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, CTRL, 0xF, 0xF, false));
}
// (LPP 3) the values of the pixel's lanes 0, 1, 2 in every lane of the pixel: role q has its own value, takes the others
// from 1 or 2 lanes to the left / right (the passive lane 15 computes garbage that nobody reads)
__device__ __forceinline__ void tri_exchange(float v, int q, float& v0, float& v1, float& v2) {
  const float l1 = dpp_row<0x111>(v), l2 = dpp_row<0x112>(v), r1 = dpp_row<0x101>(v), r2 = dpp_row<0x102>(v);
  v0 = q == 0 ? v : (q == 1 ? l1 : l2);
  v1 = q == 0 ? r1 : (q == 1 ? v : l1);
  v2 = q == 0 ? r2 : (q == 1 ? r1 : v);
}
// (LPP 3) previous result of the row above = the pixel three lanes to the left. The first pixel of DPP rows 1..3 takes
// lane 15 of the previous DPP row (the passive copy of its last pixel: row_bcast:15 into lanes 0..3, then row_shr:3
// overwrites lane 3 with lane 0); the first pixel of the wave (row 0 of the band) keeps `old` = the granule-fed value.
__device__ __forceinline__ float from_row_above_t(float old, float v) {
  int r = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), 0x142, 0xE, 0x1, false);
  r = __builtin_amdgcn_update_dpp(r, __builtin_bit_cast(int, v), 0x113, 0xF, 0xF, false);
  return __builtin_bit_cast(float, r);
}
// previous result of the row above: lane - 4. Inside a 16-lane DPP row that is row_shr:4 (banks 1..3); the first
// quad of DPP rows 1..3 takes lane 15 of the previous DPP row (row_bcast:15, bank 0); the first quad of the wave
// (row 0 of the band) keeps `old` = the granule-fed value.
__device__ __forceinline__ float from_row_above_q(float old, float v) {
  int r = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), 0x114, 0xF, 0xE, false);
  r = __builtin_amdgcn_update_dpp(r, __builtin_bit_cast(int, v), 0x142, 0xE, 0x1, false);
  return __builtin_bit_cast(float, r);
}

}  // namespace

// Everything that is not the pixel update itself is amortised over several steps, with wave-uniform control: the
// results of kQChunk steps are written back together (through an LDS ring, so that no global store sits in front of
// the next step's loads — loads and stores retire in order through one counter on gfx950), the band above is
// checked every kQNeed steps and the last row's granules are published every kQPub steps.
constexpr int kQChunk = 16;
#ifndef S360_QNEED
#define S360_QNEED 4
#endif
#ifndef S360_QPUB
#define S360_QPUB 4
#endif
constexpr int kQNeed = S360_QNEED;  // row 0 checks the band above every kQNeed steps (2..8 measured: no difference) ...
constexpr int kQPhase = kQNeed - 1;  // ... at the steps with s % kQNeed == kQPhase, for the columns of the kQNeed steps after it
constexpr int kQPub = S360_QPUB;   // the last row publishes its granules every kQPub steps

// The LDS window of I1-gradient texels. In image coordinates the pixels of a chunk form a parallelogram:
// row y of the band lags one column per row, so x + y is the same for the 16 pixels of a step and spans 16 values over a
// chunk. The window is stored in those coordinates: LDS row jy holds image row wy0 + jy, column ju holds image column
// (wu0 + ju) - (wy0 + jy). A bilinear cell (x0, y0) is inside iff 0 <= y0 - wy0 <= kWinRows - 2 and
// 0 <= x0 + y0 - wu0 <= kWinCols - 3; its texels are [jy][ju], [jy][ju + 1], [jy + 1][ju + 1], [jy + 1][ju + 2].
// rows + 2 rows and 16 + 2 columns are the chunk itself; the rest is room for the flows' spread inside the chunk and for
// the left / up candidates and the probes around them. Rows are padded to 34 texels = 68 dwords: the 16 rows of a step
// read 16-byte pieces at (almost) the same ju, which then fall on 16 disjoint groups of 4 banks.
constexpr int win_rows(int lpp) { return quad_rows(lpp) + 8; }  // 24 / 28: the band + the cell's second row + slack
constexpr int kWinCols = 32;
constexpr int kWinStride = 34;
constexpr int kNoWin = 0x3fffffff;
constexpr int kWinAhead = 4;  // the next chunk's window is requested this many steps before the chunk ends

#ifdef S360_WAVE_EMULATION
// developer statistics (CPU emulation only): wave-steps that ran an update round / that left the window for global memory
unsigned long long g_quad_rounds = 0, g_quad_fallbacks = 0, g_quad_chunks = 0, g_quad_fills = 0;
#define S360_QSTAT(v) __atomic_fetch_add(&(v), 1ull, __ATOMIC_RELAXED)
#else
#define S360_QSTAT(v)
#endif

// Persistent waves: the grid is capped (launch_sweep_quad) and a wave that finishes a band takes the next ticket.
template <bool FAST, int LPP>
__global__ __launch_bounds__(64) void k_sweep_quad(const float2* __restrict__ recAll, const float2* __restrict__ G,
                                                   float2* __restrict__ flowAll, unsigned long long* __restrict__ HAll,
                                                   unsigned* __restrict__ hdr, int w, int h, size_t bs, FlowIdx idx,
                                                   int dir, SweepConst c, SweepFast fc, int nb, int B,
                                                   unsigned* __restrict__ errflag,
                                                   const unsigned* __restrict__ rowflags) {
  // The records and flows of a 16-step chunk are fetched once (four pixels per lane, eight loads per chunk instead of
  // two per step) and staged in LDS; a slot of s_res holds a pixel's flow before its step and its result after it,
  // indexed by step. Row strides of 17 elements keep the 16 rows of a read on distinct banks.
  constexpr int kQRows = quad_rows(LPP), kWinRows = win_rows(LPP);
  constexpr int kItems = kQRows * kQChunk / 64;  // pixel-steps of a chunk per lane (4 / 5)
  constexpr int kRW = kQChunk + 1;
  typedef float f4r __attribute__((ext_vector_type(4)));
  typedef float f2r __attribute__((ext_vector_type(2)));
  __shared__ float2 s_up[kUpRing];
  __shared__ float2 s_res[kQRows][kRW];
  __shared__ float4 s_rec[kQRows][kRW];
  __shared__ f2r s_win[kWinRows * kWinStride];
  const int lane = threadIdx.x;
  for (;;) {
  unsigned tk = 0;
  if (lane == 0) tk = atomicAdd(hdr, 1u) + 1u;  // the counter starts at 0xFFFFFFFF (memset 0xFF)
  tk = __builtin_amdgcn_readfirstlane(tk);
  const int band = (int)(tk / (unsigned)B), b = (int)(tk - (unsigned)band * (unsigned)B);
  if (band >= nb) return;
  const float2* __restrict__ G1 = G + bs * (size_t)__builtin_amdgcn_readfirstlane(idx.i1[b]);  // (wave-uniform: keeps the base in SGPRs)
  const char* __restrict__ G1b0 = reinterpret_cast<const char*>(G1);
  const char* __restrict__ G1b1 = reinterpret_cast<const char*>(G1 + w);
  // a pixel's record {I0x | NaN = not updated, I0y, blurredFlow} comes from two planes: I0's gradient (the gradient kernel's
  // plane of image i0[b]) and the half-record the 15x15 blur wrote {blurredFlow.x | NaN, blurredFlow.y}
  const float2* __restrict__ rec = recAll + bs * b;
  const float2* __restrict__ G0 = G + bs * (size_t)__builtin_amdgcn_readfirstlane(idx.i0[b]);
  float2* __restrict__ flow = flowAll + bs * b;
  unsigned long long* __restrict__ H = HAll + (size_t)b * nb * w;
  const unsigned long long* Hin = H + (size_t)band * w;
  unsigned long long* Hout = H + (size_t)(band + 1) * w;
  if (rowflags) {
    // A band none of whose rows has an updated pixel (the record kernel leaves the row's word all-ones) changes
    // nothing: it hands its last row's flow to the band below as it is and takes the next ticket. Most bands of a
    // pole flow are like that.
    bool real = false;
    if (lane < kQRows) {
      const int yiL = band * kQRows + lane;
      if (yiL < h) real = rowflags[(size_t)b * h + (dir > 0 ? yiL : h - 1 - yiL)] == 0u;
    }
    if (__ballot(real) == 0ull) {
      if (band + 1 < nb) {  // (then all 16 rows exist)
        const int yl = band * kQRows + kQRows - 1;
        const float2* __restrict__ last = flow + (size_t)(dir > 0 ? yl : h - 1 - yl) * w;
        for (int xi = lane; xi < w; xi += 64) {
          const float2 v = last[dir > 0 ? xi : w - 1 - xi];
          __hip_atomic_store(Hout + xi, ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      continue;
    }
  }
  // r: row of the band; q: role in the pixel (0 current / x probe, 1 left / y probe, 2 up; 3: spare lane)
  const int l16 = lane & 15, g5 = min(l16 / 3, 4);
  const int r = LPP == 3 ? (lane >> 4) * 5 + g5 : lane >> 2, q = LPP == 3 ? l16 - 3 * g5 : lane & 3;
  const int yi = band * kQRows + r;
  const bool rowValid = yi < h;
  const int yic = rowValid ? yi : h - 1;
  const int y = dir > 0 ? yic : h - 1 - yic;
  const bool hasUp = yi > 0;
  const bool hasUpBand = band > 0;
  const bool publishes = band + 1 < nb;
  const float fy = (float)y;
  const float kEps = 0.001f, kInf = __int_as_float(0x7f800000);
  const int nsteps = w + kQRows - 1;
  auto col = [&](int xi) { const int xc = min(max(xi, 0), w - 1); return dir > 0 ? xc : w - 1 - xc; };
  int wy0 = kNoWin, wu0 = 0;  // placement of the LDS window, wave-uniform
  // the lanes whose evaluation counts in round 1 (current, left, up where a row above exists) and in round 2 (the probes)
  const unsigned long long lanesRound1 = __ballot(q == 0 || q == 1 || (q == 2 && hasUp)), lanesRound2 = __ballot(q < 2);

  // getPixBilinear32FExtend's clamp + split (PixFlow.h:457-464) of the tap (x + ax, y + ay) of this lane's pixel
  struct Cell { float mx, my; int x0, y0; };
  auto cell_of_row = [&](int x, float fyRow, float ax, float ay) -> Cell {
    Cell k;
    k.mx = __builtin_amdgcn_fmed3f((float)x + ax, 0.0f, c.wm2);
    k.my = __builtin_amdgcn_fmed3f(fyRow + ay, 0.0f, c.hm2);
    k.x0 = (int)k.mx;
    k.y0 = (int)k.my;
    return k;
  };
  auto cell_of = [&](int x, float ax, float ay) -> Cell { return cell_of_row(x, fy, ax, ay); };
  // errorFunction (PixFlow.h:493-534) on the texels of cell k. `tiny` collects the lanes whose operands leave the
  // proven range of the fast division / square root.
  auto error_of = [&](auto ieee, const Texels& tt, const Cell& k, float4 rc, float ax, float ay, bool& tiny) -> float {
    const float xR = __builtin_amdgcn_fractf(k.mx), yR = __builtin_amdgcn_fractf(k.my);
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
  auto gather = [&](const Cell& k) -> Texels {  // the cell's four texels from global memory
    unsigned boff = (unsigned)(__umul24(k.y0, w) + k.x0) << 3;
    if (S360_DBG(fc, 1)) boff = (unsigned)lane << 4;  // timing experiment (results invalid): gathers that always hit
    const f4a8 ta = *reinterpret_cast<const f4a8*>(G1b0 + boff);
    const f4a8 tb = *reinterpret_cast<const f4a8*>(G1b1 + boff);
    Texels tt;
    tt.r0 = make_float4(ta.x, ta.y, ta.z, ta.w);
    tt.r1 = make_float4(tb.x, tb.y, tb.z, tb.w);
    return tt;
  };
  // errorFunction of cell k with its four texels from the LDS window — or, for the whole wave, from global memory if a
  // lane that matters (`rel`) has its cell outside the window. The two paths are complete evaluations that only meet
  // in the resulting error: if they met in the texel registers, every step would wait there for ALL outstanding global
  // loads (one in-order counter), i.e. for the next chunk's prefetches as well.
  // (`rel` && lane in `relLanes`: the lanes that matter. The part of the test that only depends on the lane's role is a
  // constant lane mask in SGPRs — as a per-lane boolean it cost an exec-mask detour of ~15 instructions per round)
  // The test itself is scalar: two compares write their lane masks, the rest is 64-bit SALU (a ballot of a boolean that
  // was combined from masks goes through a VGPR and back).
  auto evaluate = [&](auto ieee, const Cell& k, unsigned long long relLanes, float4 rc, float ax, float ay,
                      bool& tiny) -> float {
    const int jy = k.y0 - wy0, ju = k.x0 + k.y0 - wu0;
    const bool outY = (unsigned)jy > (unsigned)(kWinRows - 2), outU = (unsigned)ju > (unsigned)(kWinCols - 3);
    const bool in = !outY && !outU;
    S360_QSTAT(g_quad_rounds);
    if (__builtin_expect(((__ballot(outY) | __ballot(outU)) & relLanes) != 0ull, 0)) {
      S360_QSTAT(g_quad_fallbacks);
      float e = error_of(ieee, gather(k), k, rc, ax, ay, tiny);
#ifndef S360_WAVE_EMULATION
      asm volatile("; window miss: evaluated from global memory" : "+v"(e));  // (keeps the two paths from being merged again)
#endif
      return e;
    }
    // (lanes that do not matter read slot 0; the 24-bit multiply-add is a full-rate instruction, the 32-bit one the
    // compiler picks for jy * kWinStride + ju — v_mad_u64_u32 — is not, and it sits in front of the LDS read)
    const int off = in ? (int)__umul24((unsigned)jy, (unsigned)kWinStride) + ju : 0;
    const f4a8 ta = *reinterpret_cast<const f4a8*>(&s_win[off]);
    const f4a8 tb = *reinterpret_cast<const f4a8*>(&s_win[off + kWinStride + 1]);
    Texels tt;
    if constexpr (!decltype(ieee)::value) {
      // Round 5: the part of errorFunction that does not need the texels (a square root with its fix-up, two divisions: ~30
      // instructions, sweep_common.hpp) runs while the two LDS reads are in flight — as the compiler scheduled it, the wave
      // waited for the window right behind the reads and evaluated everything afterwards. Same operations, same order: same bits.
      ErrPre pre = error_fast_pre(rc.z, rc.w, ax, ay, c, fc);
#ifndef S360_WAVE_EMULATION
      asm volatile("; texel-independent terms before the window's data is waited for"
                   : "+v"(pre.smTerm), "+v"(pre.vTerm), "+v"(pre.hTerm), "+v"(pre.key)
                   :
                   : "memory");
#endif
      tt.r0 = make_float4(ta.x, ta.y, ta.z, ta.w);
      tt.r1 = make_float4(tb.x, tb.y, tb.z, tb.w);
      bool t1;
      const float e = error_fast_post(tt, __builtin_amdgcn_fractf(k.mx), __builtin_amdgcn_fractf(k.my), rc.x, rc.y, pre, t1);
      tiny = tiny || t1;
      return e;
    }
    tt.r0 = make_float4(ta.x, ta.y, ta.z, ta.w);
    tt.r1 = make_float4(tb.x, tb.y, tb.z, tb.w);
    return error_of(ieee, tt, k, rc, ax, ay, tiny);
  };
  // One pixel update (PixFlow.h:390-397 / 403-410) for the quad's pixel: round 1 evaluates the current / left / up
  // proposals in lanes 0..2, round 2 the two finite-difference probes of the winner in lanes 0..1.
  auto update = [&](auto ieee, auto steady, int x, int xi, float4 rc, float2 fo, float2 fl, float2 up, bool take,
                    bool& tiny) -> float2 {
    constexpr bool ST = decltype(steady)::value;
    const float2 cand = q == 0 ? fo : (q == 2 ? up : fl);
    const float ax = cand.x, ay = cand.y;
    const Cell k = cell_of(x, ax, ay);
    const unsigned long long takeLanes = __ballot(take);
    const float e = evaluate(ieee, k, ST ? takeLanes & lanesRound1 : __ballot(take && (q == 0 || (q == 1 && xi > 0) || (q == 2 && hasUp))),
                             rc, ax, ay, tiny);
    float e0, e1, e2;
    if constexpr (LPP == 3) tri_exchange(e, q, e0, e1, e2);
    else { e0 = quad_bcast<0>(e); e1 = quad_bcast<1>(e); e2 = quad_bcast<2>(e); }
    if (!ST && !(xi > 0)) e1 = kInf;  // no left proposal in the first column
    if (!hasUp) e2 = kInf;            // no up proposal in the first row
    // proposeFlowUpdate x2 in the reference's order, written as selects (with an index beside them the compiler would
    // pick the winner's flow from a table in scratch memory)
    const bool b1 = e1 < e0;
    const float c1 = b1 ? e1 : e0;
    const bool b2 = e2 < c1;
    float2 f;
    f.x = b2 ? up.x : (b1 ? fl.x : fo.x);
    f.y = b2 ? up.y : (b1 ? fl.y : fo.y);
    const float cur = b2 ? e2 : c1;
    const float pax = f.x + (q == 0 ? kEps : 0.0f), pay = f.y + (q == 1 ? kEps : 0.0f);
    const Cell pk = cell_of(x, pax, pay);
    const float pe = evaluate(ieee, pk, takeLanes & lanesRound2, rc, pax, pay, tiny);
    float ex, ey;
    if constexpr (LPP == 3) { float unused; tri_exchange(pe, q, ex, ey, unused); }
    else { ex = quad_bcast<0>(pe); ey = quad_bcast<1>(pe); }
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
  bool pending = false, dead = S360_DBG(fc, 2) != 0;  // (dbg 2: timing experiment without the band-to-band wait)
  unsigned long long pv = kEmptyGranuleQ;
  auto issue = [&]() {
    const int xi = upFilled + lane;
    pv = kEmptyGranuleQ;
    if (xi < w) pv = __hip_atomic_load(Hin + xi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    pending = true;
  };
  auto process = [&](int limit) {  // takes the leading run of written granules, never beyond column `limit`
    const int xi = upFilled + lane;
    const unsigned long long bad = __ballot(xi >= w || (pv == kEmptyGranuleQ && !dead));
    int n = bad ? (int)__ffsll((long long)bad) - 1 : 64;
    n = min(n, limit - upFilled);
    if (lane < n)
      s_up[xi & (kUpRing - 1)] = make_float2(__uint_as_float((unsigned)pv), __uint_as_float((unsigned)(pv >> 32)));
    upFilled = __builtin_amdgcn_readfirstlane(upFilled + max(n, 0));
    pending = false;
  };

  float2 fl = make_float2(0.f, 0.f);  // result of the previous pixel of this row (same in the 4 lanes of the quad)
  float4 nrc;
  float2 nfo;
  // The next chunk's records / flows: its kQRows x 16 pixel-steps are dealt to the lanes as items — item i of a lane is
  // row 4 i + (lane >> 4), step lane & 15, i.e. a wave-wide load covers 16 consecutive pixels of four rows — and go
  // through native vector types, which stay in VGPRs through the lambdas' captures (HIP's float4 struct went to scratch).
  f2r cg[kItems], cb[kItems];  // I0's gradient, half-record
  f2r cf[kItems];
  const int ioStep = lane & 15;
  int ioOff[kItems];   // y * w of the item's row (clamped rows: never used, see ioOk)
  bool ioOk[kItems];   // the item's row exists
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    const int yiI = band * kQRows + 4 * i + (lane >> 4);
    ioOk[i] = yiI < h;
    const int yc = ioOk[i] ? yiI : h - 1;
    ioOff[i] = (dir > 0 ? yc : h - 1 - yc) * w;
  }
  auto chunk_load = [&](int sbase) {
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
      const int xc = col(sbase + ioStep - (4 * i + (lane >> 4)));
      cg[i] = *reinterpret_cast<const f2r*>(G0 + ioOff[i] + xc);
      cb[i] = *reinterpret_cast<const f2r*>(rec + ioOff[i] + xc);
      cf[i] = *reinterpret_cast<const f2r*>(flow + ioOff[i] + xc);
    }
  };
  auto chunk_store = [&]() {
#pragma unroll
    for (int i = 0; i < kItems; ++i) {
      f4r rv;
      rv.x = cb[i].x == cb[i].x ? cg[i].x : __int_as_float(0x7fc00000);  // (the NaN of a pixel that is not updated moves to .x)
      rv.y = cg[i].y; rv.z = cb[i].x; rv.w = cb[i].y;
      if (!ioOk[i]) rv.x = __int_as_float(0x7fc00000);  // rows below the image: "not updated", like a pixel below the alpha threshold
      *reinterpret_cast<f4r*>(&s_rec[4 * i + (lane >> 4)][ioStep]) = rv;
      *reinterpret_cast<f2r*>(&s_res[4 * i + (lane >> 4)][ioStep]) = cf[i];
    }
  };
  // Places the window around the cells the incoming flows of the chunk starting at step `sbase` point at (cr / cf hold
  // that chunk: four pixels per lane) and loads it into registers: 12 coalesced loads of 8 bytes per lane, two window
  // rows per wave-wide load. Pixels that are not updated do not count; a chunk without updated pixels leaves the window
  // alone. Issued four steps before the chunk ends so that the loads land behind the remaining steps; win_commit then
  // moves them into LDS between the chunks.
  f2r wv[kWinRows / 2];
  int ny0 = kNoWin, nu0 = 0;  // placement of the window being loaded (kNoWin: none)
  auto win_issue = [&](int sbase) {
    int ymin = 0x7fffffff, umin = 0x7fffffff, ymax = -1, umax = -1;
#pragma unroll
    for (int j = 0; j < kItems; ++j) {
      const int xi = sbase + ioStep - (4 * j + (lane >> 4));
      if (ioOk[j] && xi >= 0 && xi < w && cb[j].x == cb[j].x) {
        const int yiJ = band * kQRows + 4 * j + (lane >> 4);
        const Cell k = cell_of_row(dir > 0 ? xi : w - 1 - xi, (float)(dir > 0 ? yiJ : h - 1 - yiJ), cf[j].x + 0.0f, cf[j].y + 0.0f);
        ymin = min(ymin, k.y0); ymax = max(ymax, k.y0);
        umin = min(umin, k.x0 + k.y0); umax = max(umax, k.x0 + k.y0);
      }
    }
    // wave-wide minima / maxima: inside the 16-lane DPP rows by row_shr 1, 2, 4, 8 (lane 15 of a row then holds the row's
    // value), across the four rows by v_readlane of lanes 15 / 31 / 47 / 63 and scalar min / max — no LDS round trips
    auto row_red = [&](int v, bool mx) -> int {
#define S360_ROW_SHR_STEP(SH)                                                                                      \
  {                                                                                                                \
    const int o = __builtin_amdgcn_update_dpp(v, v, 0x110 + SH, 0xF, 0xF, false); /* lanes without a source keep v */ \
    v = mx ? max(v, o) : min(v, o);                                                                                \
  }
      S360_ROW_SHR_STEP(1) S360_ROW_SHR_STEP(2) S360_ROW_SHR_STEP(4) S360_ROW_SHR_STEP(8)
#undef S360_ROW_SHR_STEP
      const int a = __builtin_amdgcn_readlane(v, 15), b = __builtin_amdgcn_readlane(v, 31);
      const int c2 = __builtin_amdgcn_readlane(v, 47), d = __builtin_amdgcn_readlane(v, 63);
      return mx ? max(max(a, b), max(c2, d)) : min(min(a, b), min(c2, d));
    };
    S360_QSTAT(g_quad_chunks);
    ymin = row_red(ymin, false); umin = row_red(umin, false);
    ymax = row_red(ymax, true); umax = row_red(umax, true);
    ny0 = kNoWin;
    if (ymax < ymin) return;  // nothing to update in this chunk: no taps
    // rows ymin .. ymax + 1 and columns umin .. umax + 2 are what the incoming flows need; the slack goes evenly to
    // both sides (the left / up candidates and the probes land next to them)
    ny0 = ymin - max(0, (kWinRows - (ymax - ymin + 2)) >> 1);
    nu0 = umin - max(0, (kWinCols - (umax - umin + 3)) >> 1);
    S360_QSTAT(g_quad_fills);
    const int jc = lane & 31, jr = lane >> 5;
#pragma unroll
    for (int i = 0; i < kWinRows / 2; ++i) {
      const int Y = ny0 + 2 * i + jr, X = nu0 + jc - Y;
      // (32-bit byte offset from the wave-uniform plane base: one address VGPR, the base stays in SGPRs)
      wv[i] = *reinterpret_cast<const f2r*>(G1b0 + ((unsigned)(__umul24(min(max(Y, 0), h - 1), w) + min(max(X, 0), w - 1)) << 3));
    }
  };
  auto win_commit = [&]() {
    if (ny0 == kNoWin) return;
    const int jc = lane & 31, jr = lane >> 5;
    S360_WAVE_SYNC();  // (the previous chunk's taps have been read)
#pragma unroll
    for (int i = 0; i < kWinRows / 2; ++i) s_win[(2 * i + jr) * kWinStride + jc] = wv[i];
    S360_WAVE_SYNC();
    wy0 = ny0;
    wu0 = nu0;
  };
  chunk_load(0);
  win_issue(0);
  win_commit();
  chunk_store();
  S360_WAVE_SYNC();
  nrc = s_rec[r][0];
  nfo = s_res[r][0];
  // A chunk all of whose 16 steps have every row of the wave inside the image with a left neighbour (local steps
  // 16 .. w-1, chunk-aligned) runs the step with the range tests, the first-column case, the sweep direction select and
  // the publish / write-back bounds folded away (`steady`).
  const int xLane = dir > 0 ? -r : w - 1 + r, xSign = dir > 0 ? 1 : -1;
  auto step = [&](auto steady, int s, int send) {
    constexpr bool ST = decltype(steady)::value;
    // Row 0 needs column s of the band above only if its pixel is updated at this step (pixels below the alpha
    // threshold keep their flow): bands whose first row is never updated — most bands of the pole flows — run
    // without waiting for anybody. Checked every kQNeed steps for kQNeed columns, or on demand after a stretch
    // of steps that did not need the band above (the columns passed meanwhile are dropped).
    if (hasUpBand && (ST || s < w) && ((s & (kQNeed - 1)) == kQPhase || upFilled <= s) && (__ballot(rowValid && nrc.x == nrc.x) & 1ull)) {
      // through the column of the next scheduled check (which needs its own column like any other step)
      const int need = min(((s + 1) | (kQNeed - 1)) + 1, w), limit = s + kUpRing;
      if (upFilled < s) {
        upFilled = s;
        pending = false;
      }
      if (pending) process(limit);
      unsigned spins = 0;
      while (upFilled < need) {
        if (spins) {  // back off: a band waiting for its predecessor should leave the issue slots and the L2 to it
          if (spins < 4) __builtin_amdgcn_s_sleep(8);
          else __builtin_amdgcn_s_sleep(48);
        }
        issue();
        process(limit);
        if (++spins > (1u << 20) ||
            ((spins & 255u) == 0 && __hip_atomic_load(errflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
          dead = true;  // the band above is gone: stop waiting, flag the result invalid, keep draining
          if (lane == 0) atomicExch(errflag, 1u);
        }
      }
      if (upFilled < w && upFilled - s < 2 * kQNeed + 8) issue();  // running low: taken at the next check
    }
    const float4 rc = nrc;
    const float2 fo = nfo;
    if (s + 1 < send) {  // inputs of the next step of this chunk (the first step of the next chunk is read after the refill)
      nrc = s_rec[r][(s + 1) & (kQChunk - 1)];
      nfo = s_res[r][(s + 1) & (kQChunk - 1)];
    }
    const float2 upl = s_up[s & (kUpRing - 1)];
    const int xi = s - r;
    const bool active = ST ? rowValid : (rowValid && xi >= 0 && xi < w);
    const int x = ST ? xLane + s * xSign : (dir > 0 ? xi : w - 1 - xi);  // unclamped: out-of-range columns are inactive
    const bool upd = rc.x == rc.x;
    float2 up;
    up.x = LPP == 3 ? from_row_above_t(upl.x, fl.x) : from_row_above_q(upl.x, fl.x);
    up.y = LPP == 3 ? from_row_above_t(upl.y, fl.y) : from_row_above_q(upl.y, fl.y);
    // Pixels below the alpha threshold keep their flow (PixFlow.h:390 / :403): when none of the wave's 16 pixels is
    // updated at this step — whole bands of the pole flows, whose upper ~60 % the side cameras do not cover — the
    // two rounds are skipped.
    const bool take = ST ? upd : active && upd;  // (steady: active = the row exists, and the rows that do not have NaN records)
    const float2 alt = active ? fo : fl;
    float2 res = alt;
    if (__ballot(take) != 0ull) {
      if (FAST) {
        bool tiny = false;
        res = update(std::false_type{}, steady, x, xi, rc, fo, fl, up, take, tiny);
        if (__builtin_expect(__ballot(tiny) != 0ull, 0)) res = update(std::true_type{}, steady, x, xi, rc, fo, fl, up, take, tiny);
      } else {
        bool tiny = false;
        res = update(std::true_type{}, steady, x, xi, rc, fo, fl, up, take, tiny);
      }
      res.x = take ? res.x : alt.x;
      res.y = take ? res.y : alt.y;
    }
    if constexpr (LPP == 3) {  // lane 15 of every DPP row: passive copy of its last pixel's result (for row_bcast:15)
      const float px = dpp_row<0x111>(res.x), py = dpp_row<0x111>(res.y);
      res.x = q == 3 ? px : res.x;
      res.y = q == 3 ? py : res.y;
    }
    fl = res;
    if (q == 0) s_res[r][s & (kQChunk - 1)] = res;
    S360_WAVE_SYNC();  // (read below by the publishing lanes and by the chunk's write-back)
    if (publishes && ((s & (kQPub - 1)) == kQPub - 1 || (!ST && s == nsteps - 1))) {  // the last row's granules for the band below
      const int xi0 = (s & ~(kQPub - 1)) - (kQRows - 1) + lane;
      if (ST ? lane < kQPub : (lane < kQPub && xi0 >= 0 && xi0 < w && xi0 <= s - (kQRows - 1))) {
        const float2 v = s_res[kQRows - 1][(xi0 + kQRows - 1) & (kQChunk - 1)];
        __hip_atomic_store(Hout + xi0, ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  };
  for (int s0 = 0; s0 < nsteps; s0 += kQChunk) {
    const int send = min(s0 + kQChunk, nsteps);
    const bool steadyChunk = s0 >= kQRows && s0 + kQChunk <= w;
    const bool more = send < nsteps;
    if (more) chunk_load(send);  // lands during the chunk, stored at its end
    // The chunk's requests and the checks of the band above share the one in-order memory counter: a check waits for
    // its poll and thereby for every load requested before it. The checks therefore sit at steps 3, 7, 11, 15 of a chunk
    // (kQPhase), three steps behind the requests of the records / flows (chunk start) and of the window (step 12).
    if (steadyChunk) {
      for (int s = s0; s < s0 + kQChunk - kWinAhead; ++s) step(std::true_type{}, s, s0 + kQChunk);
    } else {
      for (int s = s0; s < send; ++s) step(std::false_type{}, s, send);
      // (edge chunks: two or three per band. What they leave pending must not look pending to the steady loops: the
      // compiler's wait-count pass would wait in every steady step wherever one of those registers is reused — devmath.hpp)
      S360_VM_DRAIN();
    }
    // (ONE call site: with a second one behind the edge chunks' loop, block placement left a static path from those window
    // loads to the steady loop, and every steady step drained the memory counter for loads that are never pending there)
    if (more) win_issue(send);
    if (steadyChunk) for (int s = s0 + kQChunk - kWinAhead; s < s0 + kQChunk; ++s) step(std::true_type{}, s, s0 + kQChunk);
    // ---- write-back of the chunk: row rr produced columns [s0 - rr, send - rr); item = (row, step) as in chunk_load ----
    {
#pragma unroll
      for (int i = 0; i < kItems; ++i) {
        const int rr = 4 * i + (lane >> 4), xi = s0 + ioStep - rr;
        if (ioOk[i] && (steadyChunk || (xi >= 0 && xi < w && s0 + ioStep < send)) && !S360_DBG(fc, 4))
          *reinterpret_cast<f2r*>(flow + ioOff[i] + (dir > 0 ? xi : w - 1 - xi)) = *reinterpret_cast<const f2r*>(&s_res[rr][ioStep]);
      }
    }
    if (more) {  // the next chunk's inputs take the slots the write-back has just read
      win_commit();
      S360_WAVE_SYNC();
      chunk_store();
      S360_WAVE_SYNC();
      nrc = s_rec[r][0];
      nfo = s_res[r][0];
    }
  }
  }  // next ticket
}

// ==========================================================================================
// Lanes per pixel of a launch. Three lanes per pixel carry 25 % more pixels per instruction and pay ~16 instructions per
// step for it: a launch that saturates the chip (the side flows of a batch of frame slots: thousands of bands) is
// instruction-issue-bound and gains (30.3 against 27.2 Gpx/s on a saturated side level), a launch that is a dependency
// chain (the pole flows: every band waits for its predecessor, about one band per SIMD) pays the longer step (33.1
// against 34.3 Gpx/s); profiles/r03_v4_*. The choice only depends on the launch's shape, so that the hand-off arena can
// be sized before the launch. S360_QUAD_LPP=3 / 4 overrides it (tests, tuning; the results do not depend on it).
static int quad_lpp(int h, int B) {
  static const int forced = [] {
    const char* e = std::getenv("S360_QUAD_LPP");
    return e && (e[0] == '3' || e[0] == '4') ? e[0] - '0' : 0;
  }();
  if (forced) return forced;
  return (long long)B * ((h + 15) / 16) >= 4096 ? 3 : 4;  // bands of 16 rows in the launch against 1024 SIMDs x 4
}
int sweep_quad_num_bands(int h, int B) { const int rows = quad_rows(quad_lpp(h, B)); return (h + rows - 1) / rows; }
size_t sweep_quad_handoff_bytes(int w, int h, int B) {
  return 256 + (size_t)B * sweep_quad_num_bands(h, B) * w * sizeof(unsigned long long);
}
void launch_sweep_quad(hipStream_t st, const float2* rec, const float2* G, float2* flow, void* handoff,
                       unsigned* errflag, int w, int h, size_t bs, int B, const FlowIdx& idx, int dir,
                       const PixFlowConsts& pc, bool fast, const unsigned* rowflags) {
  const SweepConst c = make_sweep_const(pc, w, h);
  SweepFast fc;
  fc.rcCols = 1.0f / c.fcols;
  fc.rcRows = 1.0f / c.frows;
  fc.rcEps = 1.0f / 0.001f;
  fc.dbg = S360_DBG_FROM_ENV();  // developer tools only: 1 gathers always hit, 2 no waiting on the band above, 4 no write-back
  const int nb = sweep_quad_num_bands(h, B);
  // `handoff` must be all-ones (ticket counter in the first 256 bytes, then the granules): FlowEngine resets the
  // hand-off arena of all its sweep launches with one memset.
  unsigned* hdr = reinterpret_cast<unsigned*>(handoff);
  unsigned long long* H = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(handoff) + 256);
  // S360_QUAD_WAVES_PER_CU: persistent waves per CU of one launch (tuning only; the results do not depend on it)
  // (read at every launch: tools/overlap_probe.py changes it inside one process)
  const int perCu = [] {
    const char* e = std::getenv("S360_QUAD_WAVES_PER_CU");
    const int v = e ? std::atoi(e) : 0;
    return v > 0 ? v : 8;  // (two waves per SIMD)
  }();
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  const int grid = std::min(nb * B, cus * perCu);
#define S360_LAUNCH_QUAD(F, L)                                                                                       \
  hipLaunchKernelGGL((k_sweep_quad<F, L>), dim3(grid), dim3(64), 0, st, rec, G, flow, H, hdr, w, h, bs, idx, dir, c, \
                     fc, nb, B, errflag, rowflags)
  if (quad_lpp(h, B) == 3) { if (fast) S360_LAUNCH_QUAD(true, 3); else S360_LAUNCH_QUAD(false, 3); }
  else { if (fast) S360_LAUNCH_QUAD(true, 4); else S360_LAUNCH_QUAD(false, 4); }
#undef S360_LAUNCH_QUAD
}

}  // namespace s360
