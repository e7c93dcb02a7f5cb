// isp_kernels.hip — the soft ISP of Surround360 on gfx950: 16-bit Bayer raw -> BGR.
//
// Reference: surround360_render/source/camera_isp/CameraIsp.h (the non-accelerated path of Raw2Rgb,
// Raw2Rgb.cpp:441-456): loadImage/resizeInput (:338-358, 831-854), blackLevelAdjust (:1106-1126), antiVignette
// (:1145-1154), whiteBalance (:1005-1021), clampAndStretch (:1128-1143), demosaicBilinearFilter (:89-148),
// demosaicEdgeAware (:181-335), colorCorrect (:1214-1242), sharpen (:1244-1259; util/Filter.h:38-126), getImage
// (:1275-1299). float32 in the reference's operation order (the library is built with -ffp-contract=off).
//
// Kernels (all HBM-bound streaming / small-stencil passes over planes of 4 B per pixel):
//   k_isp_front        raw16 -> normalised Bayer plane (box resize, black level, vignette curves, white balance, clamp)
//   k_isp_stuck        removeStuckPixels where it changes pixels (a serial in-place pass: one workgroup, see the kernel)
//   k_isp_green_cand   vertical / horizontal green candidates and the "horizontal is smoother" flag
//   k_isp_green_pick   9x9 vote over the flags -> the green plane
//   k_isp_color<DM>    red / blue by constant-hue interpolation (or the bilinear demosaic), composite CCM, tone LUT
//   k_isp_iir_rows / k_isp_iir_cols   the two-tap IIR low pass (causal + anticausal, reflected ends), one chain per
//                      (row, channel) / (column, channel) as in the reference — the recurrence cannot be re-associated
//   k_isp_finish       unsharp mask with noise coring (exact expf, below) and the float -> 8/16-bit BGR store
#include "isp.hpp"

#include "devmath.hpp"

namespace s360 {

namespace {

__device__ __forceinline__ int reflecti(int x, int r) { return x < 0 ? -x : x >= r ? 2 * r - x - 2 : x; }  // MathUtil.h:43-46
__device__ __forceinline__ float clampf(float x, float a, float b) { return x < a ? a : x > b ? b : x; }     // MathUtil.h:38-41
__device__ __forceinline__ float lerpf(float x0, float x1, float a) { return x0 * (1.0f - a) + x1 * a; }     // MathUtil.h:58-61
__device__ __forceinline__ float bilerpf(float x00, float x01, float x10, float x11, float a, float b) {
  return lerpf(lerpf(x00, x01, a), lerpf(x10, x11, a), b);
}
__device__ __forceinline__ bool is_red(const IspDev& d, int i, int j) { return (d.redMask >> ((i & 1) * 2 + (j & 1))) & 1; }
__device__ __forceinline__ bool is_green(const IspDev& d, int i, int j) { return (d.greenMask >> ((i & 1) * 2 + (j & 1))) & 1; }

// expf as glibc 2.35 computes it on x86-64 with FMA (sysdeps/ieee754/flt-32/e_expf.c through the multiarch FMA
// build): double arithmetic, 32-entry 2^(i/32) table, cubic in double, with the contractions gcc -mfma makes.
// tools/expf_check.c compares this sequence with the host's expf for every float <= 0 (2 139 095 041 values): no
// difference. Only arguments <= 0 occur (noise coring). The table is built on the host with exp2().
__device__ __forceinline__ float expf_glibc_neg(float x, const unsigned long long* __restrict__ tab) {
  const unsigned ux = __float_as_uint(x);
  const unsigned abstop = (ux >> 20) & 0x7ff;
  if (abstop >= (0x42b00000u >> 20)) {  // |x| >= 88 or NaN
    if (ux == 0xff800000u) return 0.0f;
    if (abstop >= (0x7f800000u >> 20)) return x + x;
    if (x < -0x1.9fe368p6f) return 0.0f;      // underflow
    if (x < -0x1.9d1d9ep6f) return 0x1p-149f;  // __math_may_uflowf
  }
  const double kInvLn2N = 0x1.71547652b82fep+0 * 32, kShift = 0x1.8p+52;
  const double C0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32, C1 = 0x1.ebfce50fac4f3p-3 / 32 / 32, C2 = 0x1.62e42ff0c52d6p-1 / 32;
  const double xd = (double)x;
  double kd = __builtin_fma(kInvLn2N, xd, kShift);
  const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
  kd -= kShift;
  const double r = __builtin_fma(kInvLn2N, xd, -kd);
  const unsigned long long t = tab[ki & 31] + (ki << (52 - 5));
  const double s = __longlong_as_double((long long)t);
  const double z = __builtin_fma(C0, r, C1);
  const double r2 = r * r;
  double y = __builtin_fma(C2, r, 1.0);
  y = __builtin_fma(z, r2, y);
  y = y * s;
  return (float)y;
}

}  // namespace

// ---- RawConverter::convert8Frame / convert12Frame (RawConverter.cpp:15-59): packed sensor bytes -> 16-bit samples ------
// 12-bit: three bytes hold two pixels (b0 << 4 | b1 & 15, then b2 << 4 | b1 >> 4); the value is widened by replicating
// its top bits (v << 4 | v >> 8). Even widths only (a row is then 3 w / 2 bytes; the reference's running byte pointer
// reads past the row for odd widths).
__global__ __launch_bounds__(256) void k_isp_unpack(const unsigned char* __restrict__ frame, int bits, int w, int h,
                                                    unsigned short* __restrict__ out) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  unsigned v;
  if (bits == 8) {
    v = (unsigned)frame[(size_t)y * w + x] * 0x101u;
  } else {
    const unsigned char* p = frame + (size_t)y * (3 * (size_t)w / 2) + (size_t)(x >> 1) * 3 + (x & 1);
    const unsigned lo = p[0], hi = p[1];
    const unsigned u = (x & 1) ? (hi << 4 | lo >> 4) : (lo << 4 | (hi & 0xFu));
    v = (u << 4 | u >> 8) & 0xFFFFu;
  }
  out[(size_t)y * w + x] = (unsigned short)v;
}
void isp_launch_unpack(hipStream_t st, const unsigned char* frame, int bits, int w, int h, unsigned short* out) {
  hipLaunchKernelGGL(k_isp_unpack, dim3((w + 255) / 256, h), dim3(256), 0, st, frame, bits, w, h, out);
}

// ---- front end: one thread per output pixel of the (resized) Bayer plane -------------------------------------------
__global__ __launch_bounds__(256) void k_isp_front(const unsigned short* __restrict__ raw, int inW, int inH,
                                                   float* __restrict__ plane, int w, int h, IspDev d,
                                                   const float* __restrict__ curveH, const float* __restrict__ curveV) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (j >= w) return;
  // resizeInput<uint16_t>: sum of resize x resize samples of the same Bayer colour
  const int rs = d.resize, rr = rs > 1 ? 2 : 1;
  float sum = 0.0f;
  for (int k = 0; k < rs; ++k) {
    const int ipp = reflecti(i * rs + k * 2 + (i % rr), inH);
    for (int l = 0; l < rs; ++l) {
      const int jpp = reflecti(j * rs + l * 2 + (j % rr), inW);
      sum += (float)raw[(size_t)ipp * inW + jpp];
    }
  }
  float v = sum * d.areaRecip;
  const int ch = is_red(d, i, j) ? 0 : is_green(d, i, j) ? 1 : 2;
  if (v < 1.0f) v = (v - d.black[ch]) * d.blackScale[ch];            // blackLevelAdjust
  v *= curveH[j * 3 + ch] * curveV[i * 3 + ch];                       // antiVignette
  v = clampf(v * d.wb[ch], 0.0f, 1.0f);                               // whiteBalance(clampOutput)
  const float lo = d.clampMin[ch], hi = d.clampMax[ch];               // clampAndStretch
  v = clampf(v, lo, hi);
  plane[(size_t)i * w + j] = (v - lo) / (hi - lo);
}

// ---- edge-aware demosaic, green channel ----------------------------------------------------------------------------
// gV / gH: the green value interpolated along the column / the row (Laplacian-corrected with the pixel's own colour);
// flag = (dH <= dV): the horizontal direction is at least as smooth.
__global__ __launch_bounds__(256) void k_isp_green_cand(const float* __restrict__ p, int w, int h, IspDev d,
                                                        float* __restrict__ gV, float* __restrict__ gH,
                                                        unsigned char* __restrict__ flag) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (j >= w) return;
  const int i_1 = reflecti(i - 1, h), i1 = reflecti(i + 1, h), i_2 = reflecti(i - 2, h), i2 = reflecti(i + 2, h);
  const int j_1 = reflecti(j - 1, w), j1 = reflecti(j + 1, w), j_2 = reflecti(j - 2, w), j2 = reflecti(j + 2, w);
  auto P = [&](int y, int x) { return p[(size_t)y * w + x]; };
  const float c = P(i, j);
  float v, hz, dv, dh;
  if (is_green(d, i, j)) {
    v = c;
    hz = c;
    dv = (fabsf(P(i2, j) - c) + fabsf(c - P(i_2, j))) / 2.0f;
    dh = (fabsf(P(i, j2) - c) + fabsf(c - P(i, j_2))) / 2.0f;
  } else {  // green neighbours above / below and left / right; the pixel's own colour two steps away
    v = (P(i_1, j) + P(i1, j)) / 2.0f;
    hz = (P(i, j_1) + P(i, j1)) / 2.0f;
    dv = (fabsf(P(i_1, j) - P(i1, j))) / 2.0f;
    dh = (fabsf(P(i, j_1) - P(i, j1))) / 2.0f;
    v += (2.0f * c - P(i_2, j) - P(i2, j)) / 4.0f;
    hz += (2.0f * c - P(i, j_2) - P(i, j2)) / 4.0f;
    dv += fabsf(-2.0f * c + P(i_2, j) + P(i2, j)) / 2.0f;
    dh += fabsf(-2.0f * c + P(i, j_2) + P(i, j2)) / 2.0f;
  }
  const size_t o = (size_t)i * w + j;
  gV[o] = v;
  gH[o] = hz;
  flag[o] = dh <= dv ? 1 : 0;
}
// homogeneity vote over the 9x9 neighbourhood (reflected): fewer than 40 of 81 "horizontal" votes -> vertical
constexpr int GP_T = 32;
__global__ __launch_bounds__(GP_T * GP_T / 4) void k_isp_green_pick(const unsigned char* __restrict__ flag,
                                                                    const float* __restrict__ gV,
                                                                    const float* __restrict__ gH, int w, int h,
                                                                    float* __restrict__ green) {
  __shared__ unsigned char s_f[GP_T + 8][GP_T + 8];
  __shared__ unsigned char s_r[GP_T + 8][GP_T];  // horizontal 9-sums
  const int x0 = blockIdx.x * GP_T, y0 = blockIdx.y * GP_T, tid = threadIdx.x;
  for (int t = tid; t < (GP_T + 8) * (GP_T + 8); t += GP_T * GP_T / 4) {
    const int ly = t / (GP_T + 8), lx = t - ly * (GP_T + 8);
    s_f[ly][lx] = flag[(size_t)reflecti(min(y0 + ly - 4, h + 3), h) * w + reflecti(min(x0 + lx - 4, w + 3), w)];
  }
  __syncthreads();
  for (int t = tid; t < (GP_T + 8) * GP_T; t += GP_T * GP_T / 4) {
    const int ly = t / GP_T, lx = t - ly * GP_T;
    int s = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) s += s_f[ly][lx + k];
    s_r[ly][lx] = (unsigned char)s;
  }
  __syncthreads();
  for (int t = tid; t < GP_T * GP_T; t += GP_T * GP_T / 4) {
    const int ly = t / GP_T, lx = t - ly * GP_T;
    const int x = x0 + lx, y = y0 + ly;
    if (x >= w || y >= h) continue;
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) cnt += s_r[ly + k][lx];
    const size_t o = (size_t)y * w + x;
    green[o] = cnt < 40 ? gV[o] : gH[o];  // hCount < diameterSquared / 2
  }
}

// ---- removeStuckPixels (CameraIsp.h:1024-1104) where it changes pixels -----------------------------------------------
// For 2 <= stuckPixelThreshold <= the region's population the reference's pass is a no-op (isp.cpp) and this kernel is not
// launched. Otherwise every pixel whose same-colour neighbourhood is dark takes that neighbourhood's median — IN PLACE and in
// boustrophedon order, so a pixel sees the new values of the pixels before it: a recurrence without a parallel order (row
// i + 1 is walked against the direction of row i; its first pixel's window holds the pixels row i finished last). What can be
// taken out of the chain is everything that does not depend on it. k_isp_stuck_pre evaluates every pixel against the ORIGINAL
// image with the whole chip; then ONE workgroup (k_isp_stuck) walks the rows. For a row, all its threads first settle every
// pixel's verdict against the image as it stands — the pre-pass's where no changed pixel lies within R rows and columns (a
// per-column "last changed row" is kept), a fresh evaluation otherwise — phase A —, then thread 0 takes the row in scan order
// in stretches of 64 positions — phase B —: a stretch in which phase A found nothing to write and which lies more than R
// columns behind the row's last changed pixel is skipped, a pixel farther than R columns from it takes phase A's verdict
// (its window has not changed), any other pixel is evaluated again on the current values. Exact for every configuration; fast
// where few pixels change (the use the pass was written for), a serial walk where every pixel does (a dark image with a
// threshold of 0 or 1).
constexpr int kStuckMaxRegion = 113;  // same-colour sites of a 15 x 15 window (R <= 7)
struct StuckEval { bool write; float value; };
__device__ inline StuckEval stuck_evaluate(const float* __restrict__ plane, int w, int h, const IspDev& d, int R, int thr,
                                           float dark, int i, int j) {
  const bool tr = is_red(d, i, j), tg = is_green(d, i, j);
  float v[kStuckMaxRegion];
  int n = 0;
  float mean = 0.0f;
  for (int y = -R; y <= R; ++y) {
    const int ip = reflecti(i + y, h);
    for (int x = -R; x <= R; ++x) {
      const int jp = reflecti(j + x, w);
      const bool pr = is_red(d, ip, jp), pg = is_green(d, ip, jp);
      if ((pr && tr) || (pg && tg) || (!pr && !pg && !tr && !tg)) {
        const float p = plane[(size_t)ip * w + jp];
        mean += p;
        if (n < kStuckMaxRegion) v[n] = p;
        ++n;
      }
    }
  }
  mean /= float(n);
  StuckEval e;
  e.write = false;
  e.value = 0.0f;
  // `for (k = n - 1; k <= n - threshold; k--)` in size_t arithmetic: entered (and then always ending at the centre pixel) iff
  // threshold <= 1 or threshold > n
  if (!(mean < dark) || (thr >= 2 && thr <= n)) return e;
  // the median VALUE sorted[n / 2]: n / 2 + 1 rounds of selecting the smallest remaining value
  float m = 0.0f;
  for (int r = 0; r <= n / 2; ++r) {
    int at = r;
    for (int k = r + 1; k < n; ++k)
      if (v[k] < v[at]) at = k;
    m = v[at];
    v[at] = v[r];
    v[r] = m;
  }
  e.write = true;
  e.value = m;
  return e;
}
// every pixel against the ORIGINAL image, all workgroups at once: the verdict holds for every pixel that no changed pixel comes near
__global__ __launch_bounds__(256) void k_isp_stuck_pre(const float* __restrict__ plane, int w, int h, IspDev d, int R, int thr,
                                                        float dark, float* __restrict__ cand0, unsigned char* __restrict__ act0,
                                                        unsigned* __restrict__ count) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (j >= w) return;
  StuckEval e;
  e.write = false;
  e.value = 0.0f;
  if (j != ((i & 1) == 0 ? w - 1 : 0)) e = stuck_evaluate(plane, w, h, d, R, thr, dark, i, j);  // (`j != jEnd`, CameraIsp.h:1054)
  cand0[(size_t)i * w + j] = e.value;
  act0[(size_t)i * w + j] = e.write ? 1 : 0;
  if (e.write) atomicAdd(count, 1u);  // (the pixels the serial pass will at least have to rewrite: what it is budgeted on)
}
__global__ __launch_bounds__(1024) void k_isp_stuck(float* __restrict__ plane, int w, int h, IspDev d, int R, int thr, float dark,
                                                     const float* __restrict__ cand0, const unsigned char* __restrict__ act0,
                                                     float* __restrict__ cand, unsigned char* __restrict__ act,
                                                     int* __restrict__ dirtyRow, unsigned* __restrict__ count, unsigned budget) {
  __shared__ unsigned char s_any[1024];  // per stretch of 64 scan positions: phase A found something to write (w <= 65536)
  const int tid = threadIdx.x;
  // The walk below is ONE workgroup and costs 1 - 5 us per pixel it rewrites (profiles/r04_v8_stuck_pixel_time.txt: 46 k pixels
  // 45 ms, 1.9 M pixels 2.2 s; a dark 2048 x 2048 frame with a threshold of 0 would hold the GPU for half a minute in this one
  // launch). Above the budget the pass is NOT run and the host refuses the frame (isp.cpp): bounded, and said aloud.
  if (budget && count[0] > budget) {
    if (tid == 0) count[1] = 1u;
    return;
  }
  for (int j = tid; j < w; j += blockDim.x) dirtyRow[j] = -0x3fffffff;  // the last row in which column j changed
  __syncthreads();
  for (int i = 0; i < h; ++i) {
    const bool even = (i & 1) == 0;
    for (int c = tid; c < 1024; c += blockDim.x) s_any[c] = 0;
    __syncthreads();
    // ---- phase A: the pre-pass's verdict, or — within R rows and columns of a changed pixel — the pixel against the current image
    for (int j = tid; j < w; j += blockDim.x) {
      bool near = false;
      for (int x = -R; x <= R; ++x) near = near || dirtyRow[min(max(j + x, 0), w - 1)] >= i - R;
      StuckEval e;
      e.write = act0[(size_t)i * w + j] != 0;
      e.value = cand0[(size_t)i * w + j];
      if (near && j != (even ? w - 1 : 0)) e = stuck_evaluate(plane, w, h, d, R, thr, dark, i, j);
      cand[j] = e.value;
      act[j] = e.write ? 1 : 0;
      if (e.write) s_any[(even ? j : w - 1 - j) >> 6] = 1;  // (every writer stores the same value)
    }
    __syncthreads();
    // ---- phase B: thread 0 walks the row in scan order — a serial chain, more threads could only repeat it
    if (tid == 0) {
      int lastDirty = -0x3fffffff;  // scan position of the last pixel of this row whose value changed
      for (int base = 0; base < w - 1; base += 64) {
        if (!s_any[base >> 6] && base - lastDirty > R) continue;
        const int end = min(base + 64, w - 1);
        for (int t = base; t < end; ++t) {
          const int j = even ? t : w - 1 - t;
          StuckEval e;
          if (t - lastDirty <= R) {
            e = stuck_evaluate(plane, w, h, d, R, thr, dark, i, j);
          } else {
            e.write = act[j] != 0;
            e.value = cand[j];
          }
          if (e.write) {
            float* px = plane + (size_t)i * w + j;
            if (__float_as_uint(*px) != __float_as_uint(e.value)) {
              *px = e.value;
              lastDirty = t;
              dirtyRow[j] = i;
            }
          }
        }
      }
    }
    __syncthreads();
  }
}

// ---- red / blue, colour correction ---------------------------------------------------------------------------------
// DM 2: constant-hue interpolation of (colour - green) around the pixel; DM 0: the bilinear demosaic on the Bayer plane.
template <int DM>
__global__ __launch_bounds__(256) void k_isp_color(const float* __restrict__ p, const float* __restrict__ green, int w,
                                                   int h, IspDev d, const float* __restrict__ lut,
                                                   float* __restrict__ img /*[h][w][3] r,g,b*/) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (j >= w) return;
  const int i_1 = reflecti(i - 1, h), i1 = reflecti(i + 1, h), j_1 = reflecti(j - 1, w), j1 = reflecti(j + 1, w);
  auto P = [&](int y, int x) { return p[(size_t)y * w + x]; };
  const bool redGreenRow = (is_red(d, i, 0) && is_green(d, i, 1)) || (is_red(d, i, 1) && is_green(d, i, 0));
  const bool red = is_red(d, i, j), grn = is_green(d, i, j);
  float r, g, b;
  if (DM == 0) {
    const float c = P(i, j);
    if (red) {
      r = c;
      g = bilerpf(P(i_1, j), P(i1, j), P(i, j_1), P(i, j1), 0.5f, 0.5f);
      b = bilerpf(P(i_1, j_1), P(i1, j_1), P(i_1, j1), P(i1, j1), 0.5f, 0.5f);
    } else if (grn) {
      g = c;
      if (redGreenRow) {
        b = (P(i_1, j) + P(i1, j)) / 2.0f;
        r = (P(i, j_1) + P(i, j1)) / 2.0f;
      } else {
        r = (P(i_1, j) + P(i1, j)) / 2.0f;
        b = (P(i, j_1) + P(i, j1)) / 2.0f;
      }
    } else {
      b = c;
      g = bilerpf(P(i_1, j), P(i1, j), P(i, j_1), P(i, j1), 0.5f, 0.5f);
      r = bilerpf(P(i_1, j_1), P(i1, j_1), P(i_1, j1), P(i1, j1), 0.5f, 0.5f);
    }
  } else {
    const int i_2 = reflecti(i - 2, h), i2 = reflecti(i + 2, h), j_2 = reflecti(j - 2, w), j2 = reflecti(j + 2, w);
    auto D = [&](int y, int x) { return P(y, x) - green[(size_t)y * w + x]; };  // colour minus green at a red / blue site
    const float pg = green[(size_t)i * w + j];
    g = pg;
    if (red || !grn) {
      const float diag = (D(i_1, j_1) + D(i1, j_1) + D(i_1, j1) + D(i1, j1)) / 4.0f + pg;
      const float own = (D(i, j) + D(i_2, j) + D(i2, j) + D(i, j_2) + D(i, j2)) / 5.0f + pg;
      r = red ? own : diag;
      b = red ? diag : own;
    } else {
      // (the reference adds (i+1, j+2) twice and never (i+1, j): CameraIsp.h:298-304)
      const float c1 = (D(i_1, j_2) + D(i_1, j) + D(i_1, j2) + D(i1, j_2) + D(i1, j2) + D(i1, j2)) / 6.0f + pg;
      const float c2 = (D(i_2, j_1) + D(i, j_1) + D(i2, j_1) + D(i_2, j1) + D(i, j1) + D(i2, j1)) / 6.0f + pg;
      b = redGreenRow ? c1 : c2;
      r = redGreenRow ? c2 : c1;
    }
  }
  // colorCorrect: composite CCM (already scaled by 4095), clamp, truncate to the LUT index
  float* o = img + ((size_t)i * w + j) * 3;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float v = d.ccm[k * 3] * r + d.ccm[k * 3 + 1] * g + d.ccm[k * 3 + 2] * b;
    const int idx = (int)clampf(v, 0.0f, 4095.0f);
    o[k] = lut[idx * 3 + k];
  }
}

// ---- IIR low pass (Filter.h:38-90): v = ip * (1 - alpha) + v * alpha along the row, then back; reflected ends -------
// A first-order recurrence per (row, channel) — then per (column, channel) — that bit-exactness forbids re-associating: 6144
// chains of 2048 dependent steps for a 2048^2 image. What can be taken away is everything that is not the chain. Each
// direction is two kernels (causal, then anticausal continuing from the causal pass's last value, handed over in `state`:
// a kernel boundary makes the intermediate image visible).
//   * rows: the first version walked a row with one thread per (row, channel): every load of a wave touched 64 different
//     rows, 0.97 ms per image. Now a wave owns 21 rows (63 chains) and takes them in tiles of 64 positions: the tile — 21
//     segments of 192 consecutive floats — comes in as coalesced loads (the next tile's are requested before this one is
//     walked), is transposed through LDS (row stride 195 floats: the 63 lanes of a step fall on 63 different banks), walked
//     in place, and leaves as coalesced stores;
//   * columns: a chain per thread is already coalesced (consecutive threads = consecutive floats of a row), but every step
//     waited for its own load, 1.28 ms per image; now the loads of the next 32 rows are in flight while 32 steps run.
// Measured per 2048^2 image: profiles/r04_v3_isp_kernel_stats.txt.
constexpr int IR_ROWS = 21, IR_T = 64, IR_STRIDE = 195, IR_Q = IR_ROWS * 3;  // IR_Q wave-wide loads per tile
// causal (ANTI false): in = image, out[m] = state after consuming in[reflect(m + 1)], m = 0 .. w-1, starting from in[0];
// anticausal (ANTI true): in = the causal pass's output, out[m] = clamp(state after consuming in[reflect(m - 1)]), m = w-1 .. 0
// PIPE: the accelerated pipeline's variant of the same recurrence (CameraIspGen.cpp:451-500, `lp`): the chain starts AT the
// first element (which keeps its value), consumes in[m] instead of in[reflect(m +- 1)], and nothing is clamped.
template <bool ANTI, bool PIPE>
__global__ __launch_bounds__(64) void k_isp_iir_rows_t(const float* __restrict__ in, float* __restrict__ out,
                                                       float* __restrict__ state, int w, int h, float alpha, float maxVal) {
  __shared__ float s_t[IR_ROWS * IR_STRIDE];
  const int lane = threadIdx.x;
  const int r = lane / 3, k = lane - 3 * r;  // this lane's chain (lane 63 has none)
  const int row0 = blockIdx.x * IR_ROWS;
  const int i = row0 + r;
  const bool chain = lane < IR_Q && i < h;
  const float ia = 1.0f - alpha;
  const size_t pitch = (size_t)w * 3;
  float v = 0.0f;
  if (chain) v = ANTI ? state[(size_t)i * 3 + k] : in[(size_t)i * pitch + k];
  const int ntiles = (w + IR_T - 1) / IR_T;
  float regs[IR_Q];
  // element q * 64 + lane of a tile = row q / 3, float (q % 3) * 64 + lane of its 192-float segment
  auto request = [&](int tile) {
    const int m0 = tile * IR_T;
#pragma unroll
    for (int q = 0; q < IR_Q; ++q) {
      const int rr = q / 3, e = (q % 3) * 64 + lane;
      const int p = e / 3, ch = e - 3 * p;
      int pos = m0 + p + (PIPE ? 0 : ANTI ? -1 : 1);
      pos = reflecti(pos, w);
      pos = min(max(pos, 0), w - 1);  // (positions behind the image's end are never walked)
      const int ii = min(row0 + rr, h - 1);
      regs[q] = in[(size_t)ii * pitch + (size_t)pos * 3 + ch];
    }
  };
  request(ANTI ? ntiles - 1 : 0);
  for (int tt = 0; tt < ntiles; ++tt) {
    const int tile = ANTI ? ntiles - 1 - tt : tt;
    const int m0 = tile * IR_T;
    __syncthreads();  // (the previous tile has left LDS)
#pragma unroll
    for (int q = 0; q < IR_Q; ++q) s_t[(q / 3) * IR_STRIDE + (q % 3) * 64 + lane] = regs[q];
    __syncthreads();
    if (tt + 1 < ntiles) request(ANTI ? tile - 1 : tile + 1);  // in flight while this tile is walked
    if (chain) {
      float* t = &s_t[r * IR_STRIDE + k];
      const int n = min(IR_T, w - m0);
      if (n == IR_T && !(PIPE && tt == 0)) {
#pragma unroll 8
        for (int pp = 0; pp < IR_T; ++pp) {
          const int p = ANTI ? IR_T - 1 - pp : pp;
          v = t[3 * p] * ia + v * alpha;
          t[3 * p] = ANTI && !PIPE ? clampf(v, 0.0f, maxVal) : v;
        }
      } else {
        for (int pp = 0; pp < n; ++pp) {
          const int p = ANTI ? n - 1 - pp : pp;
          if (!(PIPE && m0 + p == (ANTI ? w - 1 : 0))) v = t[3 * p] * ia + v * alpha;  // (PIPE: the chain's first element stays)
          t[3 * p] = ANTI && !PIPE ? clampf(v, 0.0f, maxVal) : v;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < IR_Q; ++q) {
      const int rr = q / 3, e = (q % 3) * 64 + lane;
      const int p = e / 3;
      if (row0 + rr < h && m0 + p < w) out[(size_t)(row0 + rr) * pitch + (size_t)m0 * 3 + e] = s_t[rr * IR_STRIDE + e];
    }
  }
  if (!ANTI && chain) state[(size_t)i * 3 + k] = v;
}
// One thread per (column, channel); IC_U rows per batch of loads, two register banks: while one batch is walked the
// next one is in flight. A batch's loads are requested in the order the walk consumes them (loads retire in order through one
// counter: the anticausal pass walks upwards, and a first version that requested its rows top-down waited for the whole batch
// — and, through a register copy, for the NEXT batch — at every step: 0.47 ms per image against 0.16 for the causal pass).
// 32 rows per batch: the counter has six bits, and a slot's wait must name the 31 + 32 younger requests behind it exactly.
constexpr int IC_U = 32;
template <bool ANTI, bool PIPE>
__global__ __launch_bounds__(64) void k_isp_iir_cols_t(const float* __restrict__ in, float* __restrict__ out,
                                                       float* __restrict__ state, int w, int h, float alpha, float maxVal) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= w * 3) return;
  const size_t pitch = (size_t)w * 3;
  const float ia = 1.0f - alpha;
  float v = ANTI ? state[t] : in[t];
  float bankA[IC_U], bankB[IC_U];
  const int nb = (h + IC_U - 1) / IC_U;
  auto block_of = [&](int bb) { return ANTI ? nb - 1 - bb : bb; };
  // slot uu of a bank = the uu-th row the walk takes: row m0 + uu (causal) / m0 + IC_U - 1 - uu (anticausal)
  auto request = [&](int b, float* dst) {
    const int m0 = b * IC_U;
#pragma unroll
    for (int uu = 0; uu < IC_U; ++uu) {
      const int row = m0 + (ANTI ? IC_U - 1 - uu : uu);
      int pos = reflecti(row + (PIPE ? 0 : ANTI ? -1 : 1), h);
      pos = min(max(pos, 0), h - 1);  // (rows behind the image's end are never walked)
      dst[uu] = in[(size_t)pos * pitch + t];
    }
  };
  auto walk = [&](int b, const float* src) {
    const int m0 = b * IC_U;
    if (m0 + IC_U <= h && !(PIPE && (ANTI ? m0 + IC_U == h : m0 == 0))) {
      float res[IC_U];
#pragma unroll
      for (int uu = 0; uu < IC_U; ++uu) {
        v = src[uu] * ia + v * alpha;
        res[uu] = ANTI && !PIPE ? clampf(v, 0.0f, maxVal) : v;
      }
      // (stores behind the walk: issued between the steps they would count as younger requests and push the waits
      // beyond what the counter can name)
#pragma unroll
      for (int uu = 0; uu < IC_U; ++uu) out[(size_t)(m0 + (ANTI ? IC_U - 1 - uu : uu)) * pitch + t] = res[uu];
    } else {
#pragma unroll
      for (int uu = 0; uu < IC_U; ++uu) {
        const int row = m0 + (ANTI ? IC_U - 1 - uu : uu);
        if (row < h) {
          if (!(PIPE && row == (ANTI ? h - 1 : 0))) v = src[uu] * ia + v * alpha;  // (PIPE: the chain's first element stays)
          out[(size_t)row * pitch + t] = ANTI && !PIPE ? clampf(v, 0.0f, maxVal) : v;
        }
      }
    }
  };
  request(block_of(0), bankA);
  for (int bb = 0; bb < nb; bb += 2) {
    if (bb + 1 < nb) request(block_of(bb + 1), bankB);
    walk(block_of(bb), bankA);
    if (bb + 2 < nb) request(block_of(bb + 2), bankA);
    if (bb + 1 < nb) walk(block_of(bb + 1), bankB);
  }
  if (!ANTI) state[t] = v;
}

// ---- unsharp mask with noise coring (Filter.h:92-126) + output conversion (CameraIsp.h:1275-1299) ----------------------
template <bool SHARPEN, typename OUT>
__global__ __launch_bounds__(256) void k_isp_finish(const float* __restrict__ img, const float* __restrict__ lp,
                                                    size_t n /*pixels*/, IspDev d,
                                                    const unsigned long long* __restrict__ exptab,
                                                    OUT* __restrict__ out) {
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float v = img[p * 3 + k];
    if (SHARPEN) {
      const float l = lp[p * 3 + k];
      const float hp = v - l;
      const float ng = 1.0f - expf_glibc_neg(-((hp * hp) * d.noiseCore), exptab);
      v = clampf(l + hp * ng * d.amount[k], 0.0f, d.maxVal);
    }
    out[p * 3 + (2 - k)] = (OUT)(int)v;  // float -> uchar / short by truncation; swizzled to B,G,R
  }
}

// ======================================================================================================================
void isp_launch(hipStream_t st, const IspDev& d, const unsigned short* raw, int inW, int inH, const IspFrameBufs& B,
                void* out) {
  const int w = inW / d.resize, h = inH / d.resize;
  const size_t n = (size_t)w * h;
  const dim3 row(256), grd((w + 255) / 256, h);
  hipLaunchKernelGGL(k_isp_front, grd, row, 0, st, raw, inW, inH, B.plane, w, h, d, B.curveH, B.curveV);
  if (d.stuckR > 0)  // removeStuckPixels where it changes pixels (isp_derive leaves stuckR 0 where it is the reference's no-op)
  {
    (void)hipMemsetAsync(B.stuckCount, 0, 2 * sizeof(unsigned), st);
    hipLaunchKernelGGL(k_isp_stuck_pre, grd, row, 0, st, B.plane, w, h, d, d.stuckR, d.stuckThr, d.stuckDark, B.stuckCand0, B.stuckAct0,
                       B.stuckCount);
    hipLaunchKernelGGL(k_isp_stuck, dim3(1), dim3(1024), 0, st, B.plane, w, h, d, d.stuckR, d.stuckThr, d.stuckDark, B.stuckCand0,
                       B.stuckAct0, B.stuckCand, B.stuckAct, B.stuckDirty, B.stuckCount, B.stuckBudget);
  }
  if (d.demosaic == 0) {
    hipLaunchKernelGGL((k_isp_color<0>), grd, row, 0, st, B.plane, nullptr, w, h, d, B.lut, B.img);
  } else {
    hipLaunchKernelGGL(k_isp_green_cand, grd, row, 0, st, B.plane, w, h, d, B.gV, B.gH, B.flag);
    hipLaunchKernelGGL(k_isp_green_pick, dim3((w + GP_T - 1) / GP_T, (h + GP_T - 1) / GP_T), dim3(GP_T * GP_T / 4), 0, st,
                       B.flag, B.gV, B.gH, w, h, B.green);
    hipLaunchKernelGGL((k_isp_color<2>), grd, row, 0, st, B.plane, B.green, w, h, d, B.lut, B.img);
  }
  const unsigned gp = (unsigned)((n + 255) / 256);
  if (d.sharpen) {
    // rows: img -> scratch (causal) -> lp (anticausal, clamped); columns: lp -> scratch -> lp
    const dim3 gr((h + IR_ROWS - 1) / IR_ROWS), gc((w * 3 + 63) / 64);
    hipLaunchKernelGGL((k_isp_iir_rows_t<false, false>), gr, dim3(64), 0, st, B.img, B.scratch, B.state, w, h, d.alpha, d.maxVal);
    hipLaunchKernelGGL((k_isp_iir_rows_t<true, false>), gr, dim3(64), 0, st, B.scratch, B.lp, B.state, w, h, d.alpha, d.maxVal);
    hipLaunchKernelGGL((k_isp_iir_cols_t<false, false>), gc, dim3(64), 0, st, B.lp, B.scratch, B.state, w, h, d.alpha, d.maxVal);
    hipLaunchKernelGGL((k_isp_iir_cols_t<true, false>), gc, dim3(64), 0, st, B.scratch, B.lp, B.state, w, h, d.alpha, d.maxVal);
    if (d.outputBpp == 8)
      hipLaunchKernelGGL((k_isp_finish<true, unsigned char>), dim3(gp), dim3(256), 0, st, B.img, B.lp, n, d, B.exptab,
                         (unsigned char*)out);
    else
      hipLaunchKernelGGL((k_isp_finish<true, unsigned short>), dim3(gp), dim3(256), 0, st, B.img, B.lp, n, d, B.exptab,
                         (unsigned short*)out);
  } else {
    if (d.outputBpp == 8)
      hipLaunchKernelGGL((k_isp_finish<false, unsigned char>), dim3(gp), dim3(256), 0, st, B.img, nullptr, n, d,
                         B.exptab, (unsigned char*)out);
    else
      hipLaunchKernelGGL((k_isp_finish<false, unsigned short>), dim3(gp), dim3(256), 0, st, B.img, nullptr, n, d,
                         B.exptab, (unsigned short*)out);
  }
}

// ======================================================================================================================
// The ACCELERATED ISP's arithmetic: CameraIspPipe (camera_isp/CameraIspPipe.h), i.e. the Halide pipeline that
// camera_isp/CameraIspGen.cpp generates — what the reference's Unpacker always runs (Unpacker.cpp:24,176-183) and Raw2Rgb runs
// with --accelerate (Raw2Rgb.cpp:427-440). Written from the generator's source, function by function; Halide cannot be built
// here, so this arithmetic is NOT pinned against the reference (include/s360.h and DESIGN.md section 8 say what may differ at
// rounding level). Same structure as the soft ISP above, other details: the image is extended by mirroring WITHOUT boundary
// logic in the stencils — the site plane is computed for 8 pixels beyond every edge (raw: mirror_interior, vignette tables:
// mirror_image, site colour from the virtual coordinate: CameraIspGen.cpp:674-680), the flags for 6, green for 2 —; black
// level, white balance and clamp are one A (x - B); the horizontal vignette table has its green and blue columns swapped
// (CameraIspPipe.h:88-89); the low pass runs along y first, starts at the first element and clamps nothing; the noise gain of
// output channel c is read at c while everything else is read at the swizzled channel (CameraIspGen.cpp:529-537).
//   k_pipe_site<FAST>   raw16 -> the extended Bayer plane in [0, 1]
//   k_pipe_flag         dH <= dV on the plane extended by 6
//   k_pipe_green        9 x 9 vote (LDS, separable sums) -> green on the plane extended by 2
//   k_pipe_color<FAST>  red / blue (through r - g, b - g; or the bilinear demosaic), CCM, 12-bit index, tone table
//   k_isp_iir_cols_t / k_isp_iir_rows_t <.., PIPE>   the low pass (the kernels above with the pipeline's chain ends)
//   k_pipe_finish       unsharp mask with noise coring, or (FAST) the swizzled store
constexpr int PP = 8, PF = 6, PG = 2;  // how far beyond the image the site plane / the flags / green are computed
__device__ __forceinline__ int mirror_interior(int x, int n) { return x < 0 ? -x : x >= n ? 2 * n - 2 - x : x; }
__device__ __forceinline__ int mirror_image(int x, int n) { return x < 0 ? -x - 1 : x >= n ? 2 * n - 1 - x : x; }
// site colours (CameraIspGen.cpp:44-67; pattern 0 = GBRG, 1 = RGGB): 0 red, 1 green, 2 blue. (x, y) may be negative.
__device__ __forceinline__ int pipe_colour(int pattern, int x, int y) {
  const int px = x & 1, py = y & 1;
  if (pattern == 0) return px == py ? 1 : (px == 0 ? 0 : 2);  // G B / R G
  return px != py ? 1 : (px == 0 ? 0 : 2);                     // R G / G B
}
// green on a red row (greenRedPixel): the site's horizontal neighbours are red
__device__ __forceinline__ bool pipe_green_red(int pattern, int x, int y) {
  return pattern == 0 ? ((x & 1) == 1 && (y & 1) == 1) : ((x & 1) == 1 && (y & 1) == 0);
}
__device__ __forceinline__ float pavg(float a, float b) { return (a + b) * 0.5f; }
__device__ __forceinline__ float pabsd(float a, float b) { return a > b ? a - b : b - a; }

template <bool FAST>
__global__ __launch_bounds__(256) void k_pipe_site(const unsigned short* __restrict__ raw, int w, int h, IspPipeDev d,
                                                   const float* __restrict__ vigH, const float* __restrict__ vigV,
                                                   float* __restrict__ site) {
  const int X = blockIdx.x * blockDim.x + threadIdx.x, Y = blockIdx.y;  // coordinates in the extended plane
  if (X >= w + 2 * PP) return;
  const int x = X - PP, y = Y - PP;
  const int k = pipe_colour(d.pattern, x, y);
  float v = ((float)raw[(size_t)mirror_interior(y, h) * w + mirror_interior(x, w)] - d.bias[k]) * d.invRange[k];
  if (!FAST) v *= vigH[mirror_image(x, w) * 3 + k] * vigV[mirror_image(y, h) * 3 + k];
  site[(size_t)Y * (w + 2 * PP) + X] = fmaxf(fminf(v, 1.0f), 0.0f);
}

__global__ __launch_bounds__(256) void k_pipe_flag(const float* __restrict__ site, int w, int h, IspPipeDev d,
                                                   unsigned char* __restrict__ flag) {
  const int X = blockIdx.x * blockDim.x + threadIdx.x, Y = blockIdx.y;  // coordinates in the plane extended by PF
  if (X >= w + 2 * PF) return;
  const int x = X - PF, y = Y - PF;
  const int sp = w + 2 * PP;
  const float* c = site + (size_t)(y + PP) * sp + (x + PP);
  const float s0 = c[0];
  float dV, dH;
  if (pipe_colour(d.pattern, x, y) == 1) {
    dV = pavg(pabsd(c[2 * sp], s0), pabsd(c[-2 * sp], s0));
    dH = pavg(pabsd(c[2], s0), pabsd(c[-2], s0));
  } else {
    dV = pavg(pabsd(c[sp], c[-sp]), pabsd(c[2 * sp] + c[-2 * sp], 2.0f * s0));
    dH = pavg(pabsd(c[1], c[-1]), pabsd(c[2] + c[-2], 2.0f * s0));
  }
  flag[(size_t)Y * (w + 2 * PF) + X] = dH <= dV ? 1 : 0;
}

__global__ __launch_bounds__(GP_T * GP_T / 4) void k_pipe_green(const unsigned char* __restrict__ flag,
                                                                const float* __restrict__ site, int w, int h, IspPipeDev d,
                                                                float* __restrict__ green) {
  __shared__ unsigned char s_f[GP_T + 8][GP_T + 8];
  __shared__ unsigned char s_r[GP_T + 8][GP_T];  // horizontal 9-sums
  // output (X, Y) of the plane extended by PG = image pixel (X - PG, Y - PG); its window starts at flag-plane (X, Y)
  const int X0 = blockIdx.x * GP_T, Y0 = blockIdx.y * GP_T, tid = threadIdx.x;
  const int fw = w + 2 * PF, fh = h + 2 * PF;
  for (int t = tid; t < (GP_T + 8) * (GP_T + 8); t += GP_T * GP_T / 4) {
    const int ly = t / (GP_T + 8), lx = t - ly * (GP_T + 8);
    s_f[ly][lx] = flag[(size_t)min(Y0 + ly, fh - 1) * fw + min(X0 + lx, fw - 1)];  // (clamped entries feed no output)
  }
  __syncthreads();
  for (int t = tid; t < (GP_T + 8) * GP_T; t += GP_T * GP_T / 4) {
    const int ly = t / GP_T, lx = t - ly * GP_T;
    int s = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) s += s_f[ly][lx + k];
    s_r[ly][lx] = (unsigned char)s;
  }
  __syncthreads();
  const int sp = w + 2 * PP;
  for (int t = tid; t < GP_T * GP_T; t += GP_T * GP_T / 4) {
    const int ly = t / GP_T, lx = t - ly * GP_T;
    const int X = X0 + lx, Y = Y0 + ly;
    if (X >= w + 2 * PG || Y >= h + 2 * PG) continue;
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) cnt += s_r[ly + k][lx];
    const int x = X - PG, y = Y - PG;
    const float* c = site + (size_t)(y + PP) * sp + (x + PP);
    float g = c[0];
    if (pipe_colour(d.pattern, x, y) != 1) {
      if (cnt < 81 / 2) g = pavg(c[sp], c[-sp]) + (2.0f * c[0] - c[2 * sp] - c[-2 * sp]) * 0.25f;  // gV
      else g = pavg(c[1], c[-1]) + (2.0f * c[0] - c[2] - c[-2]) * 0.25f;                           // gH
    }
    green[(size_t)Y * (w + 2 * PG) + X] = g;
  }
}

template <bool FAST>
__global__ __launch_bounds__(256) void k_pipe_color(const float* __restrict__ site, const float* __restrict__ green, int w,
                                                    int h, IspPipeDev d, const unsigned short* __restrict__ toneTab,
                                                    float* __restrict__ tone /*[h][w][3] r,g,b*/) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  const int sp = w + 2 * PP, gp = w + 2 * PG;
  const float* c = site + (size_t)(y + PP) * sp + (x + PP);
  const int col = pipe_colour(d.pattern, x, y);
  const bool gr = pipe_green_red(d.pattern, x, y);
  float r, g, b;
  if (FAST) {  // bilinearDemosaic (CameraIspGen.cpp:112-160)
    const float vert = pavg(c[-sp], c[sp]), horz = pavg(c[-1], c[1]);
    const float cross = pavg(pavg(c[-1], c[1]), pavg(c[-sp], c[sp]));
    const float diag = pavg(pavg(c[-sp - 1], c[-sp + 1]), pavg(c[sp - 1], c[sp + 1]));
    if (col == 1) {
      g = c[0];
      r = gr ? horz : vert;
      b = gr ? vert : horz;
    } else {
      g = cross;
      r = col == 0 ? c[0] : diag;
      b = col == 0 ? diag : c[0];
    }
  } else {  // edgeAwareDemosaic's r and b (CameraIspGen.cpp:235-262): box filters of (colour - green), green added back
    const float* q = green + (size_t)(y + PG) * gp + (x + PG);
    auto D = [&](int dx, int dy) { return c[dy * sp + dx] - q[dy * gp + dx]; };
    g = q[0];
    const float fifth = 1.0f / 5.0f, sixth = 1.0f / 6.0f;  // (a float division by a constant is a multiplication in Halide)
    if (col != 1) {
      const float own = (D(0, 0) + D(0, 2) + D(0, -2) + D(2, 0) + D(-2, 0)) * fifth + g;
      const float dia = (D(1, 1) + D(1, -1) + D(-1, 1) + D(-1, -1)) * 0.25f + g;
      r = col == 0 ? own : dia;
      b = col == 0 ? dia : own;
    } else {
      const float cols = (D(-1, -2) + D(-1, 0) + D(-1, 2) + D(1, -2) + D(1, 0) + D(1, 2)) * sixth + g;
      const float rows = (D(-2, -1) + D(0, -1) + D(2, -1) + D(-2, 1) + D(0, 1) + D(2, 1)) * sixth + g;
      r = gr ? cols : rows;
      b = gr ? rows : cols;
    }
  }
  float* o = tone + ((size_t)y * w + x) * 3;
#pragma unroll
  for (int k = 0; k < 3; ++k) {  // applyCCM (:429-449), toneTable (:599)
    const float v = d.ccm[k * 3] * r + d.ccm[k * 3 + 1] * g + d.ccm[k * 3 + 2] * b;
    const int idx = (int)fmaxf(fminf(v, 4095.0f), 0.0f);
    o[k] = (float)toneTab[idx * 3 + k];
  }
}

// applyUnsharpMask (CameraIspGen.cpp:502-547) and the output cast; FAST: ispOutput = toneCorrected at the swizzled channel
template <bool FAST, typename OUT>
__global__ __launch_bounds__(256) void k_pipe_finish(const float* __restrict__ tone, const float* __restrict__ low, size_t n,
                                                     IspPipeDev d, const unsigned long long* __restrict__ exptab,
                                                     OUT* __restrict__ out) {
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  float t[3], l[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    t[k] = tone[p * 3 + k];
    l[k] = FAST ? 0.0f : low[p * 3 + k];
  }
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const int cp = d.swizzle ? 2 - ch : ch;
    float v = t[cp];
    if (!FAST) {
      const float hpC = t[ch] - l[ch];
      const float ng = 1.0f - expf_glibc_neg(-((hpC * hpC) * d.noiseCore), exptab);
      const float hp = t[cp] - l[cp];
      v = fmaxf(fminf(l[cp] + hp * ng * d.amount[cp], d.maxVal), 0.0f);
    }
    out[p * 3 + ch] = (OUT)(int)v;
  }
}

void isp_pipe_launch(hipStream_t st, const IspPipeDev& d, const unsigned short* raw, int w, int h, const IspPipeBufs& B,
                     void* out) {
  const size_t n = (size_t)w * h;
  const dim3 row(256);
  if (d.fast) hipLaunchKernelGGL((k_pipe_site<true>), dim3((w + 2 * PP + 255) / 256, h + 2 * PP), row, 0, st, raw, w, h, d, B.vigH, B.vigV, B.site);
  else hipLaunchKernelGGL((k_pipe_site<false>), dim3((w + 2 * PP + 255) / 256, h + 2 * PP), row, 0, st, raw, w, h, d, B.vigH, B.vigV, B.site);
  const dim3 grd((w + 255) / 256, h);
  const unsigned gp = (unsigned)((n + 255) / 256);
  if (d.fast) {
    hipLaunchKernelGGL((k_pipe_color<true>), grd, row, 0, st, B.site, nullptr, w, h, d, B.toneTab, B.tone);
    if (d.outputBpp == 8) hipLaunchKernelGGL((k_pipe_finish<true, unsigned char>), dim3(gp), row, 0, st, B.tone, nullptr, n, d, B.exptab, (unsigned char*)out);
    else hipLaunchKernelGGL((k_pipe_finish<true, unsigned short>), dim3(gp), row, 0, st, B.tone, nullptr, n, d, B.exptab, (unsigned short*)out);
    return;
  }
  hipLaunchKernelGGL(k_pipe_flag, dim3((w + 2 * PF + 255) / 256, h + 2 * PF), row, 0, st, B.site, w, h, d, B.flag);
  hipLaunchKernelGGL(k_pipe_green, dim3((w + 2 * PG + GP_T - 1) / GP_T, (h + 2 * PG + GP_T - 1) / GP_T), dim3(GP_T * GP_T / 4), 0,
                     st, B.flag, B.site, w, h, d, B.green);
  hipLaunchKernelGGL((k_pipe_color<false>), grd, row, 0, st, B.site, B.green, w, h, d, B.toneTab, B.tone);
  // lowPass0 = lp along y, lowPass = lp along x (CameraIspGen.cpp:515-518): tone -> scratch (causal) -> low (anticausal), twice
  const dim3 gr((h + IR_ROWS - 1) / IR_ROWS), gc((w * 3 + 63) / 64);
  hipLaunchKernelGGL((k_isp_iir_cols_t<false, true>), gc, dim3(64), 0, st, B.tone, B.scratch, B.state, w, h, d.alpha, d.maxVal);
  hipLaunchKernelGGL((k_isp_iir_cols_t<true, true>), gc, dim3(64), 0, st, B.scratch, B.low, B.state, w, h, d.alpha, d.maxVal);
  hipLaunchKernelGGL((k_isp_iir_rows_t<false, true>), gr, dim3(64), 0, st, B.low, B.scratch, B.state, w, h, d.alpha, d.maxVal);
  hipLaunchKernelGGL((k_isp_iir_rows_t<true, true>), gr, dim3(64), 0, st, B.scratch, B.low, B.state, w, h, d.alpha, d.maxVal);
  if (d.outputBpp == 8) hipLaunchKernelGGL((k_pipe_finish<false, unsigned char>), dim3(gp), row, 0, st, B.tone, B.low, n, d, B.exptab, (unsigned char*)out);
  else hipLaunchKernelGGL((k_pipe_finish<false, unsigned short>), dim3(gp), row, 0, st, B.tone, B.low, n, d, B.exptab, (unsigned short*)out);
}

}  // namespace s360
