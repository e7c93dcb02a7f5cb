// sweep_tri.hip — the throughput sweep with THREE lanes per pixel ("tri"): an experiment behind S360_SWEEP_TRI=1.
//
// Same recurrence, same results and same structure as sweep_quad.hip (read that header first): persistent single-wave
// workgroups, granule hand-off between bands, LDS-staged inputs per 16-step chunk, masked-step and masked-band short
// cuts. What differs is the lane mapping. The quad kernel spends 4 lanes on a pixel and uses 3 of them in the first
// evaluation round (current / left / up proposals) and 2 in the second (the finite-difference probes): 5 evaluations in
// 8 lane slots. Here a pixel has 3 lanes — 5 evaluations in 6 slots — and a wave carries 20 rows instead of 16:
//   * a 16-lane DPP row holds 5 pixels: lanes {0,1,2} {3,4,5} {6,7,8} {9,10,11} {12,13,14}; lane 15 is a passive copy of
//     the last pixel's result, so that row_bcast:15 can hand it to the first pixel of the next DPP row;
//   * the three evaluations of a round are exchanged with row_shr / row_shl by 1 and 2 and a per-role select (a quad
//     broadcast does not exist for groups of three): ~16 more instructions per step for 25 % more pixels per step;
//   * bands are 20 rows high: w + 19 steps per band, 20 % fewer bands and hand-offs per flow.
// S360_SWEEP_TRI=2 adds the round-2 texel exchange of sweep_quad.hip (the probes take the winner's texels from its lane).
// Bit-identical to the other sweeps on the CPU emulation (tests/test_cpu_sweep_emulation.py); not yet timed on hardware.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "devmath.hpp"
#include "sweep_common.hpp"

namespace s360 {

namespace {

constexpr unsigned long long kEmptyGranuleQ = 0xFFFFFFFFFFFFFFFFull;
constexpr int kQRows = 20;   // rows per wave (4 DPP rows x 5 pixels)
constexpr int kUpRing = 64;  // columns of the band above kept in LDS

template <int CTRL>
__device__ __forceinline__ float dpp_row(float v) {  // row_shr:n (0x110 + n) / row_shl:n (0x100 + n); lanes without a source keep v
  const int i = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, CTRL, 0xF, 0xF, false));
}
// previous result of the row above = the pixel three lanes to the left. The first pixel of DPP rows 1..3 takes lane 15 of
// the previous DPP row (the passive copy of its last pixel: row_bcast:15 into lanes 0..3, then row_shr:3 overwrites lane 3
// with lane 0); the first pixel of the wave (row 0 of the band) keeps `old` = the granule-fed value.
// v of the lane whose byte address (lane * 4) is `src`
__device__ __forceinline__ float lane_value_t(int src, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, v)));
}
__device__ __forceinline__ float from_row_above_t(float old, float v) {
  int r = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), 0x142, 0xE, 0x1, false);
  r = __builtin_amdgcn_update_dpp(r, __builtin_bit_cast(int, v), 0x113, 0xF, 0xF, false);
  return __builtin_bit_cast(float, r);
}

}  // namespace

// Everything that is not the pixel update itself is amortised over several steps, with wave-uniform control: the
// results of kQChunk steps are written back together (through an LDS ring, so that no global store sits in front of
// the next step's gathers — loads and stores retire in order through one counter on gfx950), the band above is
// checked every kQNeed steps and the last row's granules are published every kQPub steps.
constexpr int kQChunk = 16;
constexpr int kQResRing = 2 * kQChunk;  // result columns per row kept in LDS
#ifndef S360_QNEED
#define S360_QNEED 4
#endif
#ifndef S360_QPUB
#define S360_QPUB 4
#endif
constexpr int kQNeed = S360_QNEED;  // row 0 checks the band above every kQNeed steps (2..8 measured: no difference)
constexpr int kQPub = S360_QPUB;   // the last row publishes its granules every kQPub steps

// Persistent waves: the grid is capped (launch_sweep_quad) and a wave that finishes a band takes the next ticket.
// The sweeps of one launch then hold a bounded share of every CU — a wave's working set is ~8 KB (17 gradient rows +
// its record / flow lines) and beyond ~15 waves per CU the 32 KB L1 thrashes — and the kernels of another context
// find free wave slots, registers and LDS next to them.
template <bool FAST, bool R2X>
__global__ __launch_bounds__(64) void k_sweep_tri(const float4* __restrict__ recAll, const float2* __restrict__ G,
                                                   float2* __restrict__ flowAll, unsigned long long* __restrict__ HAll,
                                                   unsigned* __restrict__ hdr, int w, int h, size_t bs, FlowIdx idx,
                                                   int dir, SweepConst c, SweepFast fc, int nb, int B,
                                                   unsigned* __restrict__ errflag,
                                                   const unsigned* __restrict__ rowflags) {
  // LDSIN: the records and flows of a 16-step chunk are fetched once (four pixels per lane, eight loads per chunk
  // instead of two per step) and staged in LDS; a slot of s_res then holds a pixel's flow before its step and its
  // result after it, indexed by step. Row strides of 17 elements keep the 16 rows of a read on distinct banks.
  constexpr bool LDSIN = true;
  constexpr int kRW = kQChunk + 1;
  __shared__ float2 s_up[kUpRing];
  __shared__ float2 s_res[kQRows][kRW];
  __shared__ float4 s_rec[LDSIN ? kQRows : 1][LDSIN ? kQChunk + 1 : 1];
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
  const float4* __restrict__ rec = recAll + bs * b;
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
  const int l16 = lane & 15, g5 = min(l16 / 3, 4);
  const int r = (lane >> 4) * 5 + g5, q = l16 - 3 * g5;  // q: 0 current / x probe, 1 left / y probe, 2 up, 3 = lane 15 (passive)
  const int yi = band * kQRows + r;
  const bool rowValid = yi < h;
  const int yic = rowValid ? yi : h - 1;
  const int y = dir > 0 ? yic : h - 1 - yic;
  const bool hasUp = yi > 0;
  const bool hasUpBand = band > 0;
  const bool publishes = band + 1 < nb;
  const float4* __restrict__ recRow = rec + (size_t)y * w;
  float2* __restrict__ flowRow = flow + (size_t)y * w;
  const float fy = (float)y;
  const float kEps = 0.001f, kInf = __int_as_float(0x7f800000);
  const int nsteps = w + kQRows - 1;
  auto col = [&](int xi) { const int xc = min(max(xi, 0), w - 1); return dir > 0 ? xc : w - 1 - xc; };

  // errorFunction at (x + ax, y + ay) for this lane's pixel (PixFlow.h:493-534). `tiny` collects the lanes whose
  // operands leave the proven range of the fast division / square root.
  auto evaluate = [&](auto ieee, int x, float4 rc, float ax, float ay, bool& tiny) -> float {
    const float mx = __builtin_amdgcn_fmed3f((float)x + ax, 0.0f, c.wm2);
    const float my = __builtin_amdgcn_fmed3f(fy + ay, 0.0f, c.hm2);
    const int x0 = (int)mx, y0 = (int)my;
    const float xR = __builtin_amdgcn_fractf(mx), yR = __builtin_amdgcn_fractf(my);
    unsigned boff = (unsigned)(__umul24(y0, w) + x0) << 3;
    if (S360_DBG(fc, 1)) boff = (unsigned)lane << 4;  // timing experiment (results invalid): gathers that always hit
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
  // One pixel update (PixFlow.h:390-397 / 403-410) for the quad's pixel: round 1 evaluates the current / left / up
  // proposals in lanes 0..2, round 2 the two finite-difference probes of the winner in lanes 0..1.
  // R2X (S360_SWEEP_TRI=2): the round-2 texel exchange of sweep_quad.hip — the probes take the winner's texels from its lane.
  struct Cell { float mx, my; int x0, y0; };
  bool r2xTake = true;
  auto cell_of = [&](int x, float ax, float ay) -> Cell {
    Cell k;
    k.mx = __builtin_amdgcn_fmed3f((float)x + ax, 0.0f, c.wm2);
    k.my = __builtin_amdgcn_fmed3f(fy + ay, 0.0f, c.hm2);
    k.x0 = (int)k.mx;
    k.y0 = (int)k.my;
    return k;
  };
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
  auto update = [&](auto ieee, auto steady, int x, int xi, float4 rc, float2 fo, float2 fl, float2 up, bool& tiny) -> float2 {
    const float2 cand = q == 0 ? fo : (q == 1 ? fl : up);
    float e;
    float g0 = 0, g1 = 0, g2 = 0, g3 = 0, g4 = 0, g5 = 0, g6 = 0, g7 = 0;  // (R2X) the texels this lane gathered in round 1
    if constexpr (R2X) {
      const float ax = cand.x + 0.0f, ay = cand.y + 0.0f;
      const Cell k = cell_of(x, ax, ay);
      const unsigned boff = (unsigned)(__umul24(k.y0, w) + k.x0) << 3;
      const f4a8 ta = *reinterpret_cast<const f4a8*>(G1b0 + boff);
      const f4a8 tb = *reinterpret_cast<const f4a8*>(G1b1 + boff);
      g0 = ta.x; g1 = ta.y; g2 = ta.z; g3 = ta.w; g4 = tb.x; g5 = tb.y; g6 = tb.z; g7 = tb.w;
      Texels t1;
      t1.r0 = make_float4(g0, g1, g2, g3);
      t1.r1 = make_float4(g4, g5, g6, g7);
      e = error_of(ieee, t1, k, rc, ax, ay, tiny);
    } else {
      e = evaluate(ieee, x, rc, cand.x + 0.0f, cand.y + 0.0f, tiny);
    }
    // the three errors of the pixel in each of its lanes: neighbours one and two lanes to either side, picked by role
    const float er1 = dpp_row<0x111>(e), er2 = dpp_row<0x112>(e), el1 = dpp_row<0x101>(e), el2 = dpp_row<0x102>(e);
    const float e0 = q == 0 ? e : (q == 1 ? er1 : er2);
    float e1 = q == 0 ? el1 : (q == 1 ? e : er1), e2 = q == 0 ? el2 : (q == 1 ? el1 : e);
    if (!decltype(steady)::value && !(xi > 0)) e1 = kInf;  // no left proposal in the first column
    if (!hasUp) e2 = kInf;     // no up proposal in the first row
    float2 f = fo;
    float cur = e0;
    float pe;
    if constexpr (R2X) {
      const bool b1 = e1 < e0;
      const float c1 = b1 ? e1 : e0;
      const bool b2 = e2 < c1;
      f.x = b2 ? up.x : (b1 ? fl.x : fo.x);
      f.y = b2 ? up.y : (b1 ? fl.y : fo.y);
      cur = b2 ? e2 : c1;
      const int win = b2 ? 2 : (b1 ? 1 : 0);  // = the role of the lane that evaluated the winner
      const float pax = f.x + (q == 0 ? kEps : 0.0f), pay = f.y + (q == 1 ? kEps : 0.0f);
      const Cell pk = cell_of(x, pax, pay), wk = cell_of(x, f.x + 0.0f, f.y + 0.0f);
      if (__ballot(q < 2 && r2xTake && (pk.x0 != wk.x0 || pk.y0 != wk.y0)) == 0ull) {
        const int src = (lane - min(q, 2) + win) << 2;
        Texels t2;
        t2.r0 = make_float4(lane_value_t(src, g0), lane_value_t(src, g1), lane_value_t(src, g2), lane_value_t(src, g3));
        t2.r1 = make_float4(lane_value_t(src, g4), lane_value_t(src, g5), lane_value_t(src, g6), lane_value_t(src, g7));
        pe = error_of(ieee, t2, pk, rc, pax, pay, tiny);
      } else {
        pe = evaluate(ieee, x, rc, pax, pay, tiny);
      }
    } else {
      if (e1 < cur) { f = fl; cur = e1; }
      if (e2 < cur) { f = up; cur = e2; }
      pe = evaluate(ieee, x, rc, f.x + (q == 0 ? kEps : 0.0f), f.y + (q == 1 ? kEps : 0.0f), tiny);
    }
    const float pr1 = dpp_row<0x111>(pe), pr2 = dpp_row<0x112>(pe), pl1 = dpp_row<0x101>(pe);
    const float ex = q == 0 ? pe : (q == 1 ? pr1 : pr2), ey = q == 0 ? pl1 : (q == 1 ? pe : pr1);
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
  // the next chunk's records / flows of this lane's steps q, q + 3, ... (< 16; lane 15 takes none). Native vector types: an
  // array of HIP's float4 structs captured by the lambdas below is not promoted to registers (the quad kernel's goes
  // through scratch memory, and waits for its loads right after issuing them in order to store them there).
  typedef float f4r __attribute__((ext_vector_type(4)));
  typedef float f2r __attribute__((ext_vector_type(2)));
  f4r cr[6];
  f2r cf[6];
  auto chunk_load = [&](int sbase) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int e = q + 3 * j;
      if (q < 3 && e < kQChunk) {
        const int xc = col(sbase + e - r);
        cr[j] = *reinterpret_cast<const f4r*>(recRow + xc);
        cf[j] = *reinterpret_cast<const f2r*>(flowRow + xc);
      }
    }
  };
  auto chunk_store = [&]() {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int e = q + 3 * j;
      if (q < 3 && e < kQChunk) {
        *reinterpret_cast<f4r*>(&s_rec[r][e]) = cr[j];
        *reinterpret_cast<f2r*>(&s_res[r][e]) = cf[j];
      }
    }
  };
  if (LDSIN) {
    chunk_load(0);
    chunk_store();
    S360_WAVE_SYNC();
    nrc = s_rec[r][0];
    nfo = s_res[r][0];
  } else {
    const int x0c = col(0 - r);
    nrc = recRow[x0c];
    nfo = flowRow[x0c];
  }
  for (int s0 = 0; s0 < nsteps; s0 += kQChunk) {
    const int send = min(s0 + kQChunk, nsteps);
    if (LDSIN && s0 + kQChunk < nsteps) chunk_load(s0 + kQChunk);  // lands during the chunk, stored at its end
    for (int s = s0; s < send; ++s) {
      // Row 0 needs column s of the band above only if its pixel is updated at this step (pixels below the alpha
      // threshold keep their flow): bands whose first row is never updated — most bands of the pole flows — run
      // without waiting for anybody. Checked every kQNeed steps for kQNeed columns, or on demand after a stretch
      // of steps that did not need the band above (the columns passed meanwhile are dropped).
      if (hasUpBand && s < w && ((s & (kQNeed - 1)) == 0 || upFilled <= s) && (__ballot(rowValid && nrc.x == nrc.x) & 1ull)) {
        const int need = min(s + kQNeed, w), limit = s + kUpRing;
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
      if (LDSIN) {  // inputs of the next step of this chunk (the first step of the next chunk is read after the refill)
        if (s + 1 < send) {
          nrc = s_rec[r][(s + 1) & (kQChunk - 1)];
          nfo = s_res[r][(s + 1) & (kQChunk - 1)];
        }
      } else {  // inputs of the next step (one step ahead: their latency hides behind this step's two gather rounds)
        const int xn = S360_DBG(fc, 8) ? col(-r) : col(s + 1 - r);  // (dbg 8: timing experiment, inputs that always hit)
        if (S360_DBG(fc, 16)) {  // (dbg 16: inputs as streaming loads that do not allocate in L1; results unchanged)
          typedef float f4n __attribute__((ext_vector_type(4)));
          typedef float f2n __attribute__((ext_vector_type(2)));
          const f4n a = __builtin_nontemporal_load(reinterpret_cast<const f4n*>(recRow) + xn);
          const f2n bq = __builtin_nontemporal_load(reinterpret_cast<const f2n*>(flowRow) + xn);
          nrc = make_float4(a.x, a.y, a.z, a.w);
          nfo = make_float2(bq.x, bq.y);
        } else {
          nrc = recRow[xn];
          nfo = flowRow[xn];
        }
      }
      const float2 upl = s_up[s & (kUpRing - 1)];
      const int xi = s - r;
      const bool active = rowValid && xi >= 0 && xi < w;
      const int x = dir > 0 ? xi : w - 1 - xi;  // unclamped: out-of-range columns are inactive, their gathers are clamped
      const bool upd = rc.x == rc.x;
      float2 up;
      up.x = from_row_above_t(upl.x, fl.x);
      up.y = from_row_above_t(upl.y, fl.y);
      // Pixels below the alpha threshold keep their flow (PixFlow.h:390 / :403): when none of the wave's 16 pixels is
      // updated at this step — whole bands of the pole flows, whose upper ~60 % the side cameras do not cover — the
      // two gather rounds and the evaluations are skipped.
      const bool take = active && upd;
      const float2 alt = active ? fo : fl;
      float2 res = alt;
      r2xTake = take;
      if (__ballot(take) != 0ull) {
        if (FAST) {
          bool tiny = false;
          res = update(std::false_type{}, std::false_type{}, x, xi, rc, fo, fl, up, tiny);
          if (__builtin_expect(__ballot(tiny) != 0ull, 0)) res = update(std::true_type{}, std::false_type{}, x, xi, rc, fo, fl, up, tiny);
        } else {
          bool tiny = false;
          res = update(std::true_type{}, std::false_type{}, x, xi, rc, fo, fl, up, tiny);
        }
        res.x = take ? res.x : alt.x;
        res.y = take ? res.y : alt.y;
      }
      {  // lane 15: the passive copy of its DPP row's last pixel (read by row_bcast:15 at the next step)
        const float cx = dpp_row<0x111>(res.x), cy = dpp_row<0x111>(res.y);
        res.x = q == 3 ? cx : res.x;
        res.y = q == 3 ? cy : res.y;
      }
      fl = res;
      if (q == 0) s_res[r][LDSIN ? (s & (kQChunk - 1)) : (xi & (kQResRing - 1))] = res;
      S360_WAVE_SYNC();  // (read below by the publishing lanes and by the chunk's write-back)
      if (publishes && ((s & (kQPub - 1)) == kQPub - 1 || s == nsteps - 1)) {  // the last row's granules for the band below
        const int xi0 = (s & ~(kQPub - 1)) - (kQRows - 1) + lane;
        if (lane < kQPub && xi0 >= 0 && xi0 < w && xi0 <= s - (kQRows - 1)) {
          const float2 v = s_res[kQRows - 1][LDSIN ? ((xi0 + kQRows - 1) & (kQChunk - 1)) : (xi0 & (kQResRing - 1))];
          __hip_atomic_store(Hout + xi0, ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    // ---- write-back of the chunk: row r produced columns [s0 - r, send - r) ----
    {
      const int base = s0 - r + q;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int xi = base + 3 * k;
        if (q < 3 && q + 3 * k < kQChunk && rowValid && xi >= 0 && xi < w && xi < send - r && !S360_DBG(fc, 4))
          flowRow[dir > 0 ? xi : w - 1 - xi] = s_res[r][LDSIN ? ((xi + r) & (kQChunk - 1)) : (xi & (kQResRing - 1))];
      }
    }
    if (LDSIN && send < nsteps) {  // the next chunk's inputs take the slots the write-back has just read
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
int sweep_tri_num_bands(int h) { return (h + kQRows - 1) / kQRows; }
size_t sweep_tri_handoff_bytes(int w, int h, int B) {
  return 256 + (size_t)B * sweep_tri_num_bands(h) * w * sizeof(unsigned long long);
}
void launch_sweep_tri(hipStream_t st, const float4* rec, const float2* G, float2* flow, void* handoff, unsigned* errflag,
                      int w, int h, size_t bs, int B, const FlowIdx& idx, int dir, const PixFlowConsts& pc, bool fast,
                      const unsigned* rowflags) {
  const SweepConst c = make_sweep_const(pc, w, h);
  SweepFast fc;
  fc.rcCols = 1.0f / c.fcols;
  fc.rcRows = 1.0f / c.frows;
  fc.rcEps = 1.0f / 0.001f;
  fc.dbg = 0;
  const int nb = sweep_tri_num_bands(h);
  // `handoff` must be all-ones (ticket counter in the first 256 bytes, then the granules), like the quad kernel's
  unsigned* hdr = reinterpret_cast<unsigned*>(handoff);
  unsigned long long* H = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(handoff) + 256);
  static const int perCu = [] {  // persistent waves per CU (the quad kernel's switch)
    const char* e = std::getenv("S360_QUAD_WAVES_PER_CU");
    const int v = e ? std::atoi(e) : 10;
    return v > 0 ? v : 10;
  }();
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  const int grid = std::min(nb * B, cus * perCu);
  static const bool r2x = [] {  // S360_SWEEP_TRI=2: with the round-2 texel exchange
    const char* e = std::getenv("S360_SWEEP_TRI");
    return e && e[0] == '2';
  }();
#define S360_LAUNCH_TRI(F, X)                                                                                                   \
  hipLaunchKernelGGL((k_sweep_tri<F, X>), dim3(grid), dim3(64), 0, st, rec, G, flow, H, hdr, w, h, bs, idx, dir, c, fc, nb, B, \
                     errflag, rowflags)
  if (fast) {
    if (r2x) S360_LAUNCH_TRI(true, true);
    else S360_LAUNCH_TRI(true, false);
  } else {
    if (r2x) S360_LAUNCH_TRI(false, true);
    else S360_LAUNCH_TRI(false, false);
  }
#undef S360_LAUNCH_TRI
}

}  // namespace s360
