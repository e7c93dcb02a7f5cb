// sweep_tile.hip — "tile" sweep: the PixFlow propagation sweeps (PixFlow.h:388-410) with every memory access
// turned into a stream and every gather served from LDS.
//
// What limits the other two sweep kernels once the chip is full is not arithmetic but the vector-memory path: a
// wave that walks 16 rows touches ~48 different cache lines per step (per-row records, per-row flow in/out and the
// data-dependent bilinear gathers of I1's gradients); the 32 KB L1 of a CU thrashes as soon as a few such waves
// share it (rocprofv3: 170 L1 accesses and 52 L2 requests per wave-step, 62 % of wave cycles waiting). Here:
//   * the per-pixel inputs {I0x|NaN, I0y, blurredFlow, old flow} are written by k_make_records_skew in the ORDER
//     the sweep consumes them — [band][step][row], 32 bytes each — so one step of a band is 512 contiguous bytes;
//     the wave brings 16 steps at a time into an LDS ring with 16-byte loads issued 16..32 steps early;
//   * results go out in the same skewed order (one 128-byte line per step) and k_unskew_flow puts them back;
//   * the I1 gradients live in an LDS window (16 + 2*8 + 1 rows x 128 columns, ring in x) that slides with the
//     band: it is filled with coalesced loads issued 16 steps before they are needed, and the four texels of a
//     bilinear tap are four ds_read_b64. A tap outside the window (flow larger than +-46 px in x / +-8 px in y at
//     this pyramid level) makes the whole wave take the global-gather path for that evaluation: same result.
// Nothing the wave waits on was issued less than ~16 steps earlier, so the in-order vmcnt queue never stalls it.
// Lanes and arithmetic are those of sweep_quad.hip: 16 rows per wave, 4 lanes per pixel, two evaluation rounds
// (5 errorFunction evaluations, ~14 VALU instructions per pixel); bands of a flow are chained through 8-byte
// {fx,fy} granules (all-ones = not written), ticketed in band-major order, every spin bounded.
#include "devmath.hpp"
#include "sweep_common.hpp"

namespace s360 {

namespace {

constexpr unsigned long long kEmptyGranuleT = 0xFFFFFFFFFFFFFFFFull;
constexpr int kTRows = 16;      // rows per band (wave)
constexpr int kTUpRing = 64;    // columns of the band above kept in LDS
constexpr int kTRecRing = 32;   // steps of skewed records kept in LDS (two chunks of 16)
constexpr int kTWinCols = 128;  // I1-gradient window: image columns (ring)
constexpr int kTFy = 8;         // rows of margin above / below the band
constexpr int kTWinRows = kTRows + 2 * kTFy + 1;

struct __attribute__((aligned(16))) SkewRec {  // one pixel of the skewed input stream (32 bytes)
  float4 a;  // {I0x (NaN: pixel not updated), I0y, blurredFlow.x, blurredFlow.y}
  float4 b;  // {old flow.x, old flow.y, 0, 0}
};

template <int K>
__device__ __forceinline__ float quad_bcast_t(float v) {
  const int i = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, K | (K << 2) | (K << 4) | (K << 6), 0xF, 0xF, true));
}
__device__ __forceinline__ float from_row_above_t(float old, float v) {  // lane - 4 (see sweep_quad.hip)
  int r = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), 0x114, 0xF, 0xE, false);
  r = __builtin_amdgcn_update_dpp(r, __builtin_bit_cast(int, v), 0x142, 0xE, 0x1, false);
  return __builtin_bit_cast(float, r);
}

}  // namespace

// Skewed input stream: thread = (row r of the band, step s), record for pixel (xi = s - r, yi = band*16 + r).
__global__ __launch_bounds__(256) void k_make_records_skew(const float2* __restrict__ G, const float* __restrict__ A,
                                                           const float2* __restrict__ blurred,
                                                           const float2* __restrict__ flow, SkewRec* __restrict__ out,
                                                           int w, int h, size_t bs, FlowIdx idx, int dir, int nb,
                                                           int nsteps) {
  const int r = threadIdx.x & 15;
  const int s = blockIdx.x * 16 + (threadIdx.x >> 4);
  const int band = blockIdx.y, b = blockIdx.z;
  if (s >= nsteps) return;
  const int xi = s - r, yi = band * kTRows + r;
  SkewRec v;
  v.a = make_float4(__int_as_float(0x7fc00000), 0.f, 0.f, 0.f);
  v.b = make_float4(0.f, 0.f, 0.f, 0.f);
  if (xi >= 0 && xi < w && yi < h) {
    const int x = dir > 0 ? xi : w - 1 - xi, y = dir > 0 ? yi : h - 1 - yi;
    const size_t i = (size_t)y * w + x;
    const float2 g = G[bs * idx.i0[b] + i];
    const float2 bf = blurred[bs * b + i];
    const float2 f = flow[bs * b + i];
    const bool upd = A[bs * idx.i0[b] + i] > 0.9f && A[bs * idx.i1[b] + i] > 0.9f;  // PixFlow.h:390 / :403
    v.a = make_float4(upd ? g.x : __int_as_float(0x7fc00000), g.y, bf.x, bf.y);
    v.b = make_float4(f.x, f.y, 0.f, 0.f);
  }
  out[(((size_t)b * nb + band) * nsteps + s) * kTRows + r] = v;
}

// Skewed results back to the flow image: thread = pixel.
__global__ __launch_bounds__(256) void k_unskew_flow(const float2* __restrict__ outS, float2* __restrict__ flow, int w,
                                                     int h, size_t bs, int dir, int nb, int nsteps) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
  if (x >= w) return;
  const int xi = dir > 0 ? x : w - 1 - x, yi = dir > 0 ? y : h - 1 - y;
  const int band = yi >> 4, r = yi & 15, s = xi + r;
  flow[bs * b + (size_t)y * w + x] = outS[(((size_t)b * nb + band) * nsteps + s) * kTRows + r];
}

template <bool FAST>
__global__ __launch_bounds__(64) void k_sweep_tile(const SkewRec* __restrict__ recS, const float2* __restrict__ G,
                                                   float2* __restrict__ outS, unsigned long long* __restrict__ H,
                                                   unsigned* __restrict__ hdr, int w, int h, size_t bs, FlowIdx idx,
                                                   int dir, SweepConst c, SweepFast fc, int nb, int B,
                                                   unsigned* __restrict__ errflag) {
  __shared__ float2 s_g[kTWinRows][kTWinCols];        // I1 gradient window, indexed [row - ybase][image column & 127]
  __shared__ SkewRec s_rec[kTRecRing][kTRows];        // skewed records of steps [.., ..+32)
  __shared__ float2 s_up[kTUpRing];
  __shared__ unsigned s_ticket;
  const int lane = threadIdx.x;
  if (lane == 0) s_ticket = atomicAdd(hdr, 1u) + 1u;  // the counter starts at 0xFFFFFFFF (memset 0xFF)
  __syncthreads();
  const unsigned tk = s_ticket;
  const int band = (int)(tk / (unsigned)B), b = (int)(tk - (unsigned)band * (unsigned)B);
  if (band >= nb) return;
  const int nsteps = w + kTRows - 1;
  const float2* __restrict__ G1 = G + bs * idx.i1[b];
  const char* __restrict__ G1b0 = reinterpret_cast<const char*>(G1);
  const char* __restrict__ G1b1 = reinterpret_cast<const char*>(G1 + w);
  recS += ((size_t)b * nb + band) * nsteps * kTRows;
  outS += ((size_t)b * nb + band) * nsteps * kTRows;
  H += (size_t)b * nb * w;
  const unsigned long long* Hin = H + (size_t)band * w;
  unsigned long long* Hout = H + (size_t)(band + 1) * w;
  const int r = lane >> 2, q = lane & 3;
  const int yi = band * kTRows + r;
  const bool rowValid = yi < h;
  const int yic = rowValid ? yi : h - 1;
  const int y = dir > 0 ? yic : h - 1 - yic;
  const bool hasUp = yi > 0;
  const bool hasUpBand = band > 0;
  const bool publishLane = band + 1 < nb && r == kTRows - 1 && q == 0;
  const float fy = (float)y;
  const float kEps = 0.001f, kInf = __int_as_float(0x7f800000);
  // image rows of the band: [ylo, ylo + 15]; window rows [ybase, ybase + kTWinRows)
  const int ylo = dir > 0 ? band * kTRows : h - 1 - (band * kTRows + kTRows - 1);
  const int ybase = ylo - kTFy;

  // ---- I1-gradient window: chunk g = virtual columns [16g, 16g+15] = image columns [xs, xs+15] ----
  // lane -> (row group rg = lane >> 3 handles window rows rg, rg+8, ...; pair pj = lane & 7 -> columns xs+2pj, +1)
  constexpr int kWinIter = (kTWinRows + 7) / 8;
  const int wrg = lane >> 3, wpj = lane & 7;
  float2 gA[kWinIter], gB[kWinIter];
  auto win_issue = [&](int g) {
    const int xs = dir > 0 ? 16 * g : w - 16 - 16 * g;
    const int xa = min(max(xs + 2 * wpj, 0), w - 1), xb = min(max(xs + 2 * wpj + 1, 0), w - 1);
#pragma unroll
    for (int i = 0; i < kWinIter; ++i) {
      const int yy = min(max(ybase + wrg + 8 * i, 0), h - 1);
      const float2* row = G1 + (size_t)yy * w;
      gA[i] = row[xa];
      gB[i] = row[xb];
    }
  };
  auto win_write = [&](int g) {
    const int xs = dir > 0 ? 16 * g : w - 16 - 16 * g;
    const int ca = (xs + 2 * wpj) & (kTWinCols - 1), cb = (xs + 2 * wpj + 1) & (kTWinCols - 1);
#pragma unroll
    for (int i = 0; i < kWinIter; ++i) {
      const int wr = wrg + 8 * i;
      if (wr < kTWinRows) {
        s_g[wr][ca] = gA[i];
        s_g[wr][cb] = gB[i];
      }
    }
  };
  // ---- skewed record stream: chunk c = steps [16c, 16c+16) = 16*16 records of 32 B = 8 KB = 8 x (64 lanes x 16 B) ----
  float4 rq[8];
  const float4* recQ = reinterpret_cast<const float4*>(recS);
  const int nchunks = (nsteps + 15) >> 4;
  const size_t recQuads = (size_t)nsteps * kTRows * 2;  // float4 elements in this band's stream
  auto rec_issue = [&](int cidx) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const size_t e = (size_t)cidx * 512 + i * 64 + lane;
      rq[i] = recQ[e < recQuads ? e : recQuads - 1];
    }
  };
  auto rec_write = [&](int cidx) {
    float4* dst = reinterpret_cast<float4*>(&s_rec[(cidx & 1) * 16][0]);
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[i * 64 + lane] = rq[i];
  };

  // errorFunction at (x + ax, y + ay) for this lane's pixel (PixFlow.h:493-534); evalLane: the value is used
  const int xvalidBase = dir > 0 ? 0 : w - 1;
  auto evaluate = [&](int x, float4 rc, float ax, float ay, bool evalLane, int gmax) -> float {
    const float mx = __builtin_amdgcn_fmed3f((float)x + ax, 0.0f, c.wm2);
    const float my = __builtin_amdgcn_fmed3f(fy + ay, 0.0f, c.hm2);
    const int x0 = (int)mx, y0 = (int)my;
    const float xR = __builtin_amdgcn_fractf(mx), yR = __builtin_amdgcn_fractf(my);
    // window test in virtual columns: valid [16(gmax-7), 16 gmax + 15]
    const int xv0 = dir > 0 ? x0 : xvalidBase - x0, xv1 = dir > 0 ? x0 + 1 : xv0 - 1;
    const int wy = y0 - ybase;
    const int vlo = 16 * (gmax - 7), vhi = 16 * gmax + 15;
    const bool inWin = wy >= 0 && wy <= kTWinRows - 2 && min(xv0, xv1) >= vlo && max(xv0, xv1) <= vhi;
    Texels tt;
    if (__builtin_expect(__ballot(evalLane && !inWin) == 0ull, 1)) {
      const int wyc = min(max(wy, 0), kTWinRows - 2);
      const int c0 = x0 & (kTWinCols - 1), c1 = (x0 + 1) & (kTWinCols - 1);
      const float2 t00 = s_g[wyc][c0], t10 = s_g[wyc][c1], t01 = s_g[wyc + 1][c0], t11 = s_g[wyc + 1][c1];
      tt.r0 = make_float4(t00.x, t00.y, t10.x, t10.y);
      tt.r1 = make_float4(t01.x, t01.y, t11.x, t11.y);
    } else {
      const unsigned boff = (unsigned)(__umul24(y0, w) + x0) << 3;
      const f4a8 ta = *reinterpret_cast<const f4a8*>(G1b0 + boff);
      const f4a8 tb = *reinterpret_cast<const f4a8*>(G1b1 + boff);
      tt.r0 = make_float4(ta.x, ta.y, ta.z, ta.w);
      tt.r1 = make_float4(tb.x, tb.y, tb.z, tb.w);
    }
    Foot ft;
    ft.off = 0; ft.xR = xR; ft.yR = yR;
    float e;
    if (FAST) {
      bool tiny;
      e = error_fast(tt, xR, yR, rc.x, rc.y, rc.z, rc.w, ax, ay, c, fc, tiny);
      if (__builtin_expect(__ballot(tiny) != 0ull, 0)) e = error_from(tt, ft, rc.x, rc.y, rc.z, rc.w, ax, ay, c);
    } else {
      e = error_from(tt, ft, rc.x, rc.y, rc.z, rc.w, ax, ay, c);
    }
    return e;
  };

  // ---- granules of the band above -> s_up ring (columns [upFilled - 64, upFilled) are valid) ----
  int upFilled = hasUpBand ? 0 : 0x3fffffff, pendS = -100;
  bool pending = false, dead = false;
  unsigned long long pv = kEmptyGranuleT;
  auto issue = [&](int s) {
    const int xi = upFilled + lane;
    pv = kEmptyGranuleT;
    if (xi < w) pv = __hip_atomic_load(Hin + xi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    pending = true;
    pendS = s;
  };
  auto process = [&](int s) {
    const int xi = upFilled + lane;
    const unsigned long long bad = __ballot(xi >= w || (pv == kEmptyGranuleT && !dead));
    int n = bad ? (int)__ffsll((long long)bad) - 1 : 64;
    n = min(n, s + kTUpRing - upFilled);
    if (n > 0) {
      if (lane < n)
        s_up[xi & (kTUpRing - 1)] = make_float2(__uint_as_float((unsigned)pv), __uint_as_float((unsigned)(pv >> 32)));
      upFilled = __builtin_amdgcn_readfirstlane(upFilled + n);
    }
    pending = false;
  };

  // ---- prologue: window chunks 0..2 and record chunks 0..1 in LDS, the next ones in flight ----
  for (int g = 0; g < 3; ++g) {
    win_issue(g);
    win_write(g);
  }
  win_issue(3);
  rec_issue(0);
  rec_write(0);
  if (nchunks > 1) rec_issue(1);

  float2 fl = make_float2(0.f, 0.f);  // result of the previous pixel of this row (same in the 4 lanes of the quad)
  for (int s = 0; s < nsteps; ++s) {
    if ((s & 15) == 1) {
      // What was issued 16 steps ago goes into LDS (it has long landed), then the next chunks are issued.
      const int cc = s >> 4;
      win_write(cc + 3);
      if (cc + 1 < nchunks) rec_write(cc + 1);
      __builtin_amdgcn_sched_barrier(0);
      win_issue(cc + 4);
      if (cc + 2 < nchunks) rec_issue(cc + 2);
    }
    const int gmax = ((s + 15) >> 4) + 2;
    if (hasUpBand && s < w) {
      if (pending && (s - pendS >= 2 || upFilled <= s)) process(s);
      unsigned spins = 0;
      while (upFilled <= s) {  // row 0 needs column s now
        if (!pending) issue(s);
        process(s);
        if (upFilled <= s) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > (1u << 21) ||
              ((spins & 1023u) == 0 && __hip_atomic_load(errflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            dead = true;  // the band above is gone: stop waiting, flag the result invalid, keep draining
            if (lane == 0) atomicExch(errflag, 1u);
          }
        }
      }
      if (!pending && upFilled < w && upFilled - s < 40) issue(s);
    }
    const SkewRec in = s_rec[s & (kTRecRing - 1)][r];
    const float4 rc = in.a;
    const float2 fo = make_float2(in.b.x, in.b.y);
    const float2 upl = s_up[s & (kTUpRing - 1)];
    const int xi = s - r;
    const bool active = rowValid && xi >= 0 && xi < w;
    const int x = dir > 0 ? xi : w - 1 - xi;  // unclamped: out-of-range columns are inactive
    const bool upd = rc.x == rc.x;
    float2 up;
    up.x = from_row_above_t(upl.x, fl.x);
    up.y = from_row_above_t(upl.y, fl.y);
    // round 1: the three proposals (PixFlow.h:390-393 / 403-406), lanes 0..2 of the quad
    const float2 cand = q == 0 ? fo : (q == 2 ? up : fl);
    const float e = evaluate(x, rc, cand.x + 0.0f, cand.y + 0.0f, active && upd && q < 3, gmax);
    const float e0 = quad_bcast_t<0>(e);
    float e1 = quad_bcast_t<1>(e), e2 = quad_bcast_t<2>(e);
    if (!(xi > 0)) e1 = kInf;  // no left proposal in the first column
    if (!hasUp) e2 = kInf;     // no up proposal in the first row
    float2 f = fo;
    float cur = e0;
    if (e1 < cur) { f = fl; cur = e1; }
    if (e2 < cur) { f = up; cur = e2; }
    // round 2: errorGradient's probes of the winner (PixFlow.h:195-217), lanes 0..1
    const float pe = evaluate(x, rc, f.x + (q == 0 ? kEps : 0.0f), f.y + (q == 1 ? kEps : 0.0f), active && upd && q < 2, gmax);
    const float ex = quad_bcast_t<0>(pe), ey = quad_bcast_t<1>(pe);
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
    const bool take = active && upd;
    const float2 alt = active ? fo : fl;
    res.x = take ? res.x : alt.x;
    res.y = take ? res.y : alt.y;
    fl = res;
    if (q == 0) outS[(size_t)s * kTRows + r] = res;  // one 128-byte line per step
    if (publishLane && active)
      __hip_atomic_store(Hout + xi, ((unsigned long long)__float_as_uint(res.y) << 32) | __float_as_uint(res.x),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ==========================================================================================
int sweep_tile_num_bands(int h) { return (h + kTRows - 1) / kTRows; }
size_t sweep_tile_rec_bytes(int w, int h, int B) {
  return (size_t)B * sweep_tile_num_bands(h) * (w + kTRows - 1) * kTRows * sizeof(SkewRec);
}
size_t sweep_tile_out_bytes(int w, int h, int B) {
  return (size_t)B * sweep_tile_num_bands(h) * (w + kTRows - 1) * kTRows * sizeof(float2);
}
size_t sweep_tile_handoff_bytes(int w, int h, int B) {
  return 256 + (size_t)B * sweep_tile_num_bands(h) * w * sizeof(unsigned long long);
}
// One sweep of level (w, h): skew the inputs, run the bands, put the results back into `flow`.
void launch_sweep_tile(hipStream_t st, const float2* G, const float* A, const float2* blurred, float2* flow,
                       void* recS, void* outS, void* handoff, unsigned* errflag, int w, int h, size_t bs, int B,
                       const FlowIdx& idx, int dir, const PixFlowConsts& pc, bool fast) {
  const SweepConst c = make_sweep_const(pc, w, h);
  SweepFast fc;
  fc.rcCols = 1.0f / c.fcols;
  fc.rcRows = 1.0f / c.frows;
  fc.rcEps = 1.0f / 0.001f;
  fc.dbg = 0;
  const int nb = sweep_tile_num_bands(h), nsteps = w + kTRows - 1;
  hipLaunchKernelGGL(k_make_records_skew, dim3((nsteps + 15) / 16, nb, B), dim3(256), 0, st, G, A, blurred, flow,
                     reinterpret_cast<SkewRec*>(recS), w, h, bs, idx, dir, nb, nsteps);
  (void)hipMemsetAsync(handoff, 0xFF, sweep_tile_handoff_bytes(w, h, B), st);
  unsigned* hdr = reinterpret_cast<unsigned*>(handoff);
  unsigned long long* H = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(handoff) + 256);
  if (fast)
    hipLaunchKernelGGL((k_sweep_tile<true>), dim3(nb * B), dim3(64), 0, st, reinterpret_cast<const SkewRec*>(recS), G,
                       reinterpret_cast<float2*>(outS), H, hdr, w, h, bs, idx, dir, c, fc, nb, B, errflag);
  else
    hipLaunchKernelGGL((k_sweep_tile<false>), dim3(nb * B), dim3(64), 0, st, reinterpret_cast<const SkewRec*>(recS), G,
                       reinterpret_cast<float2*>(outS), H, hdr, w, h, bs, idx, dir, c, fc, nb, B, errflag);
  hipLaunchKernelGGL(k_unskew_flow, dim3((w + 255) / 256, h, B), dim3(256), 0, st, reinterpret_cast<const float2*>(outS),
                     flow, w, h, bs, dir, nb, nsteps);
}

}  // namespace s360
