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
#define S360_QUAD_KERNEL_NAME k_sweep_quad
#define S360_QUAD_KERNEL_ATTR
#include "sweep_quad_kernel.inc"
#undef S360_QUAD_KERNEL_NAME
#undef S360_QUAD_KERNEL_ATTR
// The same text with the register allocator held to three waves per SIMD (168 VGPRs; the LDS of the 16-row mapping,
// 13 568 bytes per wave, allows twelve waves per CU, the 20-row mapping's 16 288 do not). What does not fit goes to
// scratch memory OUTSIDE the steady steps (per band and per chunk: tools/isa_loops.py, tests/test_cpu_isa.py hold the
// steady loops of both builds free of scratch accesses and of waits on the memory counter). Off unless S360_QUAD_OCC3
// asks for it (launch_sweep_quad): not measured on hardware yet.
#define S360_QUAD_KERNEL_NAME k_sweep_quad_occ3
#ifdef S360_WAVE_EMULATION
#define S360_QUAD_KERNEL_ATTR  // (the CPU emulation runs the same text; a host compiler has no such attribute)
#else
#define S360_QUAD_KERNEL_ATTR __attribute__((amdgpu_waves_per_eu(3, 3)))
#endif
#include "sweep_quad_kernel.inc"
#undef S360_QUAD_KERNEL_NAME
#undef S360_QUAD_KERNEL_ATTR

// ==========================================================================================
// Lanes per pixel of a launch. Three lanes per pixel carry 25 % more pixels per instruction and pay ~16 instructions per
// step for it: a launch that saturates the chip (the side flows of a batch of frame slots: thousands of bands) is
// instruction-issue-bound and gains (30.3 against 27.2 Gpx/s on a saturated side level), a launch that is a dependency
// chain (the pole flows: every band waits for its predecessor, about one band per SIMD) pays the longer step (33.1
// against 34.3 Gpx/s); profiles/r03_v4_*. The choice only depends on the launch's shape, so that the hand-off arena can
// be sized before the launch. S360_QUAD_LPP=3 / 4 overrides it (tests, tuning; the results do not depend on it).
static bool quad_saturates(int h, int B) { return (long long)B * ((h + 15) / 16) >= 4096; }  // bands of 16 rows in the launch against 1024 SIMDs x 4
// S360_QUAD_OCC3 (tuning; the results do not depend on it): 1 = the launches that saturate the chip run the build held to
// three waves per SIMD — which exists for the 16-row mapping only, so those launches then take four lanes per pixel —,
// 2 = every launch does, 0 / unset = none (the measured default).
static int quad_occ3_mode() {
  static const int m = [] {
    const char* e = std::getenv("S360_QUAD_OCC3");
    return e && (e[0] == '1' || e[0] == '2') ? e[0] - '0' : 0;
  }();
  return m;
}
static bool quad_occ3(int h, int B) { return quad_occ3_mode() == 2 || (quad_occ3_mode() == 1 && quad_saturates(h, B)); }
static int quad_lpp(int h, int B) {
  static const int forced = [] {
    const char* e = std::getenv("S360_QUAD_LPP");
    return e && (e[0] == '3' || e[0] == '4') ? e[0] - '0' : 0;
  }();
  if (quad_occ3(h, B)) return 4;
  if (forced) return forced;
  return quad_saturates(h, B) ? 3 : 4;
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
  static const int perCu = [] {
    const char* e = std::getenv("S360_QUAD_WAVES_PER_CU");
    const int v = e ? std::atoi(e) : 0;
    return v;  // (0: as many as are resident — 8 at two waves per SIMD, 12 at three)
  }();
  static const int cus = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n > 0 ? n : 256;
  }();
  const bool occ3 = quad_occ3(h, B);
  const int grid = std::min(nb * B, cus * (perCu > 0 ? perCu : (occ3 ? 12 : 8)));
#define S360_LAUNCH_QUAD(K, F, L)                                                                                    \
  hipLaunchKernelGGL((K<F, L>), dim3(grid), dim3(64), 0, st, rec, G, flow, H, hdr, w, h, bs, idx, dir, c, fc, nb, B, \
                     errflag, rowflags)
  if (occ3) { if (fast) S360_LAUNCH_QUAD(k_sweep_quad_occ3, true, 4); else S360_LAUNCH_QUAD(k_sweep_quad_occ3, false, 4); }
  else if (quad_lpp(h, B) == 3) { if (fast) S360_LAUNCH_QUAD(k_sweep_quad, true, 3); else S360_LAUNCH_QUAD(k_sweep_quad, false, 3); }
  else { if (fast) S360_LAUNCH_QUAD(k_sweep_quad, true, 4); else S360_LAUNCH_QUAD(k_sweep_quad, false, 4); }
#undef S360_LAUNCH_QUAD
}

}  // namespace s360
