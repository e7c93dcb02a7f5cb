// png.hip — the finished equirect as a PNG file, encoded on the device.
//
// What it replaces: imwriteExceptionOnFail(FLAGS_output_equirect_path, ...) at TRSP:938-961 (cv::imwrite's PngEncoder: 8-bit
// RGB, the Sub filter on every scanline, zlib at Z_BEST_SPEED with Z_RLE — grfmt_png.cpp as the reference's OpenCV sets it).
// The file-to-file program was bound by exactly that step (host threads filtering and deflating 201 MB per 8K frame while
// the GPU waited, VERDICT r05 "What's weak" 4); here the scanlines never leave the device uncompressed.
//
// The file: the layout host/png_io.hpp's banded writer produces (its reader inflates such files band by band in parallel) —
// signature, IHDR, the private ancillary chunk "sbNd" (rows per band), an IDAT chunk with the two zlib header bytes, ONE
// IDAT CHUNK PER BAND of scanlines, an IDAT chunk with the Adler-32 of all filtered scanlines, IEND. Every band is a
// byte-aligned raw-deflate segment (a dynamic-Huffman block closed by an empty stored block — zlib's sync flush — and a
// final block in the last band), so the concatenation is one valid zlib stream for any PNG reader.
//
// The work per band (one workgroup of 512 threads; a band is ~196 KB of filtered scanlines, 1024 bands in an 8K frame):
//   pass 1  the band's B,G,R rows tile by tile through LDS; A LANE PER BYTE: a wave takes 64 consecutive filtered bytes, every
//           lane makes its byte (Sub), one ballot of "equals my left neighbour" gives the group's repeat flags, and the tokens —
//           literals and distance-1 matches, Z_RLE's token set, runs cut at the 64-byte group — follow from the flags with a
//           shift and a find-first per lane; they go into the band's histogram, the Adler-32 pieces of the bytes on the way
//   build   length-limited canonical Huffman code of the literal/length alphabet: rank sort by all threads, the
//           two-queue merge and the 15-bit limit by one thread, canonical codes by all threads; the exact coded size is
//           then known, and a band that would not shrink goes out as stored blocks instead
//   pass 2  the same tiles again, ONE token pass: every wave writes its groups' tokens from bit 0 of a bit-buffer region of its own
//           in LDS (a DPP scan gives every token-starting lane its offset, the lane ORs its tokens — one value of up to 46 bits),
//           then the workgroup concatenates the regions behind the bits carried over from the tile before: every thread assembles
//           whole output dwords from the dwords of the regions that overlap them, and they go to HBM coalesced
// then a layout kernel (prefix sum of the bands' sizes = where each IDAT chunk starts in the file) and a gather kernel that
// moves every band to its place with the chunk's length and type in front. The host adds what needs no pixel: signature,
// IHDR, CRC-32 of every chunk (threads), the combined Adler-32, IEND.
// Algorithmic bytes per 8K frame: 201 MB read twice (the second time mostly from L2) + the compressed size written twice —
// 0.1 ms of HBM time; the kernel is bound by instructions and LDS latency per byte. History of the band kernel on an 8K frame of the
// bench (tools/png_time.py, round 6): a THREAD per 16 pixels walking its 48 bytes (byte loops, sinks per token) 1.95 ms — phase
// cuts: pass 1 0.55, code build 0.18, pass 2 1.32; privatised histograms, a two-barrier scan, tile loads in flight together: nothing
// measurable; bytes made four at a time with funnel shifts + v_perm_b32 and a token loop over repeat flags: 2.1–2.3 ms (the chains
// per token stayed); 512 threads x 8 pixels, twice the waves per CU: 1.49 ms; a lane per byte with a pass that only counted bits in
// front of the emitting pass: 1.44 ms at 741 M VALU wave-instructions per frame and 0.89 VALU busy; this form — per-wave regions
// and a concatenation instead of the counting pass — 1.41 ms at 577 M VALU instructions (2.9 per byte), VALU busy 0.71, waves
// waiting 0.49 of their time (profiles/r06_v9_png_pmc.txt); phase cuts: pass 1 0.43 ms, code build 0.14, pass 2 0.82. The
// instruction count fell by 22 % and the time did not: what is left is the workgroup's own critical path — four barriers per tile in
// pass 2, the one-thread code build, waves of unequal token counts meeting at every barrier — not issue slots and not bytes.
#include "png.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>

#include "../../include/s360.h"

namespace s360 {

namespace {
constexpr int kT = 512;                         // threads per workgroup
constexpr int kTilePx = 4096;                   // pixels of one row per workgroup iteration
constexpr int kRawWords = kTilePx * 3 / 4 + 4;  // a tile's bytes + the pixel to its left + alignment slack
constexpr int kMaxBits = 15;                    // deflate's longest code
constexpr int kWaves = kT / 64;
constexpr int kGroupsPerWave = (kTilePx * 3 / 64 + kWaves - 1) / kWaves;  // 64-byte groups of a tile a wave takes
constexpr int kRegionWords = kGroupsPerWave * 64 * kMaxBits / 32 + 4;     // a wave's bits of one tile at the longest code (+ the row's filter-type byte, + slack)
constexpr int kOutWords = kWaves * kRegionWords;
constexpr int kSyms = 288;                      // literal/length alphabet, padded (286 symbols exist)
constexpr int kNumLit = 286;
constexpr int kHistCopies = 8;
constexpr unsigned kAdler = 65521u;
constexpr unsigned kStoredMax = 65535u;

struct Smem {
  unsigned raw[kRawWords];
  unsigned out[kOutWords];
  unsigned hist[kSyms];
  // pass 1 counts into kHistCopies copies, a lane into copy lane % kHistCopies: the Sub-filtered bytes of a smooth image are mostly a
  // handful of values, and a wave's 64 atomics on one bin are served one after the other (measured on an 8K frame: 1.93 ms for the
  // band kernel with one copy)
  unsigned histp[kHistCopies][kSyms];
  unsigned code[kSyms];  // bit-reversed code | length << 16
  unsigned scan[kT];
  unsigned nodeW[2 * kSyms];
  unsigned short sorted[kSyms];
  unsigned short parent[2 * kSyms];
  unsigned char depth[2 * kSyms];
  unsigned blc[kMaxBits + 2], next[kMaxBits + 2];
  // [0] sum of bytes, [1] weighted sum (both < 65521 per add), [2] extra bits of the matches, [3] matches, [4] coded bits of the
  // symbols, [5] used symbols, [6] highest used symbol + 1
  unsigned misc[8];
};

struct Geo {
  int w, h, rows_per_band, nbands;
  unsigned line;
  unsigned long long band_stride, total_bytes;
};

// ---- deflate's length codes (RFC 1951, 3.2.5): match length 3..258 -> symbol, extra bits, their value ----
__device__ inline void length_code(int L, int& sym, int& ebits, int& eval) {
  if (L <= 10) { sym = 254 + L; ebits = 0; eval = 0; return; }
  if (L == 258) { sym = 285; ebits = 0; eval = 0; return; }
  const int m = L - 3;  // 8 .. 254
  int k = 3;
  while ((m >> (k + 1)) != 0) ++k;  // floor(log2 m): 3 .. 7
  ebits = k - 2;
  const int r = m - (1 << k);
  sym = 265 + 4 * (k - 3) + (r >> ebits);
  eval = r & ((1 << ebits) - 1);
}
__device__ inline unsigned rev_bits(unsigned c, int n) {
  unsigned r = 0;
  for (int i = 0; i < n; ++i) { r = (r << 1) | (c & 1u); c >>= 1; }
  return r;
}

// Row y's pixels [px0, px0 + kTilePx) and the pixel to their left into S.raw (dwords, coalesced); returns the byte offset
// of pixel px0 in it. `bgr` is 4-byte aligned, `total` its size.
__device__ inline int load_tile(Smem& S, const uint8_t* bgr, unsigned long long total, int w, int y, int px0) {
  const unsigned long long rowb = (unsigned long long)y * w * 3;
  const unsigned long long a0 = rowb + (unsigned long long)px0 * 3 - (px0 > 0 ? 3 : 0);
  const int npx = min(kTilePx, w - px0);
  const unsigned long long a1 = rowb + (unsigned long long)(px0 + npx) * 3;
  const unsigned long long base = a0 & ~3ull;
  const int nwords = (int)((a1 - base + 3) >> 2);
  // every dword of the thread is requested before the first goes to LDS (as a load-store loop these were up to 13 serialised
  // memory round trips per tile and pass: most of the kernel's time)
  constexpr int kIt = (kRawWords + kT - 1) / kT;
  unsigned v[kIt];
#pragma unroll
  for (int it = 0; it < kIt; ++it) {
    const int i = threadIdx.x + it * kT;
    const unsigned long long a = base + 4ull * i;
    v[it] = (i < nwords && a + 4 <= total) ? *reinterpret_cast<const unsigned*>(bgr + a) : 0u;
  }
#pragma unroll
  for (int it = 0; it < kIt; ++it) {
    const int i = threadIdx.x + it * kT;
    if (i >= nwords) continue;
    const unsigned long long a = base + 4ull * i;
    unsigned w = v[it];
    if (a + 4 > total) for (int k = 0; k < 4 && a + k < total; ++k) w |= (unsigned)bgr[a + k] << (8 * k);  // (the image's last bytes)
    S.raw[i] = w;
  }
  return (int)(rowb + (unsigned long long)px0 * 3 - base);
}

// ---- a lane per byte -------------------------------------------------------------------------------------------------------
// A wave takes 64 consecutive filtered bytes of the tile at a time. Every lane makes its byte (two LDS byte reads: the pixel's
// channel and the left pixel's), compares it with its left neighbour's (ds_bpermute), and ONE ballot gives the group's repeat
// flags. Z_RLE's tokens follow from the flags without a serial walk: a lane whose byte does not repeat its predecessor starts a
// token group — its literal, then the r repeats behind it as one distance-1 match (r >= 3) or as r more literals; r = the run of
// set flags right behind the lane = a shift and a find-first of the ballot. Runs are cut at the group (63 repeats at most).
struct Tok {
  unsigned v;  // the lane's filtered byte
  int r;       // start lane: repeats behind it; any other lane: -1 (nothing to emit)
};
__device__ inline Tok group_tok(const uint8_t* rawb, int off, int px0, int nbytes, int j) {
  const int lane = threadIdx.x & 63;
  const bool valid = j < nbytes;
  const int jj = valid ? j : 0;
  const int p = jj / 3, c = jj - 3 * p;  // pixel of the tile, channel in PNG order (R,G,B out of B,G,R)
  const int i = off + 3 * p;
  const unsigned cur = rawb[i + 2 - c];
  const unsigned left = (px0 + p == 0) ? 0u : rawb[i - 1 - c];  // Sub: the same channel one pixel to the left, 0 at the row's start
  Tok t;
  t.v = (cur - left) & 255u;
  // the left neighbour's byte: row_bcast:15 brings lanes 15 / 31 / 47 to the first lanes of the next DPP row, row_shr:1 the rest
  int pvi = __builtin_amdgcn_update_dpp((int)t.v, (int)t.v, 0x142, 0xE, 0x1, false);
  pvi = __builtin_amdgcn_update_dpp(pvi, (int)t.v, 0x111, 0xF, 0xF, false);
  const unsigned pv = (unsigned)pvi;
  const bool eq = valid && lane > 0 && t.v == pv;
  const unsigned long long E = __ballot(eq);
  const unsigned long long behind = lane == 63 ? 0ull : E >> (lane + 1);
  t.r = (valid && !eq) ? __ffsll((long long)~behind) - 1 : -1;
  return t;
}
// bits of a start lane's tokens (its literal, then a match or up to two more literals)
__device__ inline unsigned tok_bits(const Smem& S, const Tok& t) {
  if (t.r < 0) return 0u;
  const unsigned len = S.code[t.v] >> 16;
  if (t.r < 3) return len * (unsigned)(1 + t.r);
  int sym, eb, ev;
  length_code(t.r, sym, eb, ev);
  return len + (S.code[sym] >> 16) + (unsigned)eb + 1u;  // + the one-bit distance code
}
// inclusive prefix sum over the wave's lanes: four row_shr steps inside the 16-lane DPP rows, then row_bcast:15 / :31 across them
// (as six ds_bpermute steps — LDS round trips, each waiting for the one before — the scan was most of a group's time)
__device__ inline unsigned wave_scan(unsigned v) {
  int x = (int)v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);
  return (unsigned)x;
}
// up to 64 bits ORed into S.out from bit `o` on
__device__ inline void or_bits(Smem& S, unsigned o, unsigned long long val) {
  const unsigned w = o >> 5, sh = o & 31u;
  const unsigned long long lo = val << sh;
  const unsigned hi = sh ? (unsigned)(val >> (64u - sh)) : 0u;
  if ((unsigned)lo) atomicOr(&S.out[w], (unsigned)lo);
  if ((unsigned)(lo >> 32)) atomicOr(&S.out[w + 1], (unsigned)(lo >> 32));
  if (hi) atomicOr(&S.out[w + 2], hi);
}

struct BitWriter {  // ORs bits into S.out from bit `pos` on (neighbouring threads share the first and the last dword)
  Smem& S;
  unsigned long long acc = 0;
  int nacc, word;
  __device__ BitWriter(Smem& s, unsigned pos) : S(s), nacc((int)(pos & 31)), word((int)(pos >> 5)) {}
  __device__ void put(unsigned v, int n) {
    acc |= (unsigned long long)v << nacc;
    nacc += n;
    if (nacc >= 32) {
      atomicOr(&S.out[word++], (unsigned)acc);
      acc >>= 32;
      nacc -= 32;
    }
  }
  __device__ void finish() { if (nacc) atomicOr(&S.out[word], (unsigned)acc); }
  __device__ unsigned pos() const { return (unsigned)word * 32u + (unsigned)nacc; }
};
// The bit buffer is one region per wave (S.out[w * kRegionWords ..], S.scan[w] bits in it): a wave writes its tokens from bit 0 of its
// region without knowing where the waves before it end — no pass that only counts. Here the regions are concatenated behind the
// `pend` (< 32) bits carried over in `carry`: every thread assembles whole output dwords from the (at most two) dwords of each region
// that overlaps them, they go to gout[*gw ..], the last partial dword becomes the new carry, the regions are cleared.
__device__ inline unsigned take_bits(const unsigned* R, unsigned q, unsigned n) {  // n (1..32) bits of R from bit q
  const unsigned w = q >> 5, sh = q & 31u;
  const unsigned v = (unsigned)(((((unsigned long long)R[w + 1]) << 32) | R[w]) >> sh);
  return n < 32u ? v & ((1u << n) - 1u) : v;
}
__device__ inline unsigned compact_flush(Smem& S, unsigned* gout, unsigned* gw, unsigned pend, unsigned* carry) {
  unsigned start[kWaves + 1];
  start[0] = pend;
#pragma unroll
  for (int k = 0; k < kWaves; ++k) start[k + 1] = start[k] + S.scan[k];
  const unsigned total = start[kWaves], full = total >> 5;
  for (unsigned i = threadIdx.x; i <= full; i += kT) {
    const unsigned lo = 32u * i, hi = lo + 32u;
    unsigned d = i == 0 ? *carry : 0u;
#pragma unroll
    for (int k = 0; k < kWaves; ++k) {
      const unsigned a = max(lo, start[k]), b = min(hi, start[k + 1]);
      if (a < b) d |= take_bits(S.out + k * kRegionWords, a - start[k], b - a) << (a - lo);
    }
    if (i < full) gout[*gw + i] = d;
    else S.misc[7] = d;  // the bits behind the last whole dword
  }
  __syncthreads();
  *carry = S.misc[7];
  {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int used = (int)(S.scan[wv] >> 5) + 3;
    for (int i = lane; i < used; i += 64) S.out[wv * kRegionWords + i] = 0u;
  }
  *gw += full;
  __syncthreads();
  return total & 31u;
}

// filtered byte i of the band starting at row y0, straight from the image (the stored-block path)
__device__ inline unsigned filtered_byte(const uint8_t* bgr, const Geo& G, int y0, unsigned i) {
  const unsigned r = i / G.line, k = i - r * G.line;
  if (k == 0) return 1u;
  const unsigned c = k - 1, p = c / 3, ch = c - 3 * p;
  const uint8_t* px = bgr + ((unsigned long long)(y0 + r) * G.w + p) * 3;
  return (unsigned)(px[2 - ch] - (p ? px[-1 - (int)ch] : 0)) & 255u;
}

__global__ __launch_bounds__(kT) void k_png_band(const uint8_t* __restrict__ bgr, Geo G, uint8_t* __restrict__ scratch,
                                                  PngBandMeta* __restrict__ meta) {
  __shared__ Smem S;
  const int t = threadIdx.x, b = blockIdx.x;
  const int y0 = b * G.rows_per_band, rows = min(G.rows_per_band, G.h - y0);
  const unsigned n = (unsigned)rows * G.line;
  const bool last = b == G.nbands - 1;
  const int tiles = (G.w + kTilePx - 1) / kTilePx;
  const uint8_t* rawb = reinterpret_cast<const uint8_t*>(S.raw);
  unsigned* gout = reinterpret_cast<unsigned*>(scratch + (unsigned long long)b * G.band_stride);

  for (int i = t; i < kSyms; i += kT) { S.hist[i] = 0u; S.code[i] = 0u; }
  for (int i = t; i < kHistCopies * kSyms; i += kT) (&S.histp[0][0])[i] = 0u;
  for (int i = t; i < kOutWords; i += kT) S.out[i] = 0u;
  if (t < 8) S.misc[t] = 0u;
  __syncthreads();

  // ---- pass 1: histogram of the tokens, Adler-32 pieces of the filtered bytes ----
  const int lane = t & 63, wv = t >> 6;
  {
    unsigned long long sum = 0, wsum = 0;  // this lane's bytes: their sum, and the sum of (n - index in the band) x byte
    unsigned eb = 0, nm = 0;
    unsigned* H = S.histp[t % kHistCopies];
    for (int r = 0; r < rows; ++r)
      for (int tl = 0; tl < tiles; ++tl) {
        const int px0 = tl * kTilePx, npx = min(kTilePx, G.w - px0), nbytes = 3 * npx;
        const int off = load_tile(S, bgr, G.total_bytes, G.w, y0 + r, px0);
        __syncthreads();
        const int ng = (nbytes + 63) >> 6, per = (ng + kWaves - 1) / kWaves;
        const unsigned pos0 = (unsigned)r * G.line + 1u + 3u * (unsigned)px0;  // index in the band of the tile's first byte
        for (int g = wv * per; g < min((wv + 1) * per, ng); ++g) {
          const int j = 64 * g + lane;
          const Tok k = group_tok(rawb, off, px0, nbytes, j);
          if (j < nbytes) {
            sum += k.v;
            wsum += (unsigned long long)(n - (pos0 + (unsigned)j)) * k.v;
          }
          if (k.r >= 0) {
            atomicAdd(&H[k.v], k.r < 3 ? 1u + (unsigned)k.r : 1u);
            if (k.r >= 3) {
              int sym, e, ev;
              length_code(k.r, sym, e, ev);
              atomicAdd(&H[sym], 1u);
              eb += (unsigned)e;
              ++nm;
            }
          }
        }
        if (tl == 0 && t == 0) {  // the row's filter-type byte (1 = Sub): a literal of its own in front of the row
          atomicAdd(&H[1], 1u);
          sum += 1;
          wsum += n - (unsigned)r * G.line;
        }
        __syncthreads();
      }
    atomicAdd(&S.misc[0], (unsigned)(sum % kAdler));
    atomicAdd(&S.misc[1], (unsigned)(wsum % kAdler));
    atomicAdd(&S.misc[2], eb);
    atomicAdd(&S.misc[3], nm);
  }
  __syncthreads();
  for (int i = t; i < kSyms; i += kT) {
    unsigned c = i == 256 ? 1u : 0u;  // (256: the end-of-block symbol)
    for (int k = 0; k < kHistCopies; ++k) c += S.histp[k][i];
    S.hist[i] = c;
  }
  __syncthreads();

  // ---- the literal/length code: rank sort (all threads), two-queue Huffman merge + 15-bit limit (one thread) ----
  for (int s = t; s < kNumLit; s += kT) {
    const unsigned c = S.hist[s];
    if (!c) continue;
    int rank = 0;
    for (int j = 0; j < kNumLit; ++j) {
      const unsigned cj = S.hist[j];
      rank += (cj != 0u && (cj < c || (cj == c && j < s))) ? 1 : 0;
    }
    S.sorted[rank] = (unsigned short)s;
    atomicAdd(&S.misc[5], 1u);
    atomicMax(&S.misc[6], (unsigned)s + 1u);
  }
  __syncthreads();
  if (t == 0) {
    const int m = (int)S.misc[5];  // >= 2: the filter-type byte's literal and the end-of-block symbol
    for (int i = 0; i < m; ++i) S.nodeW[i] = S.hist[S.sorted[i]];
    int a = 0, q = m, nxt = m;
    while (nxt < 2 * m - 1) {
      int x, y;
      if (a < m && (q >= nxt || S.nodeW[a] <= S.nodeW[q])) x = a++; else x = q++;
      if (a < m && (q >= nxt || S.nodeW[a] <= S.nodeW[q])) y = a++; else y = q++;
      S.nodeW[nxt] = S.nodeW[x] + S.nodeW[y];
      S.parent[x] = S.parent[y] = (unsigned short)nxt;
      ++nxt;
    }
    const int root = 2 * m - 2;
    for (int l = 0; l <= kMaxBits + 1; ++l) S.blc[l] = 0u;
    S.depth[root] = 0;
    for (int i = root - 1; i >= 0; --i) {
      const int d = min(S.depth[S.parent[i]] + 1, 250);
      S.depth[i] = (unsigned char)d;
      if (i < m) S.blc[min(d, kMaxBits)] += 1u;
    }
    // codes longer than 15 bits were counted at 15: give back what that overfills, one 2^-15 at a time — a 15-bit code goes,
    // the deepest shorter code becomes two codes one bit longer
    unsigned kraft = 0;
    for (int l = 1; l <= kMaxBits; ++l) kraft += S.blc[l] << (kMaxBits - l);
    while (kraft > (1u << kMaxBits)) {
      S.blc[kMaxBits] -= 1u;
      for (int l = kMaxBits - 1; l > 0; --l)
        if (S.blc[l]) { S.blc[l] -= 1u; S.blc[l + 1] += 2u; break; }
      --kraft;
    }
    // lengths by frequency: the rarest symbols take the longest codes
    int i = 0;
    for (int l = kMaxBits; l >= 1; --l)
      for (unsigned j = 0; j < S.blc[l]; ++j) S.code[S.sorted[i++]] = (unsigned)l << 16;
    unsigned c = 0;
    S.blc[0] = 0u;
    for (int l = 1; l <= kMaxBits; ++l) { c = (c + S.blc[l - 1]) << 1; S.next[l] = c; }
  }
  __syncthreads();
  // canonical codes: within one length in symbol order (all threads), sent most-significant bit first = stored reversed
  unsigned mycode[2] = {0u, 0u};
  for (int k = 0, s = t; s < kNumLit; s += kT, ++k) {
    const unsigned l = S.code[s] >> 16;
    if (!l) continue;
    unsigned before = 0;
    for (int j = 0; j < s; ++j) before += (S.code[j] >> 16) == l ? 1u : 0u;
    mycode[k] = rev_bits(S.next[l] + before, (int)l) | (l << 16);
    atomicAdd(&S.misc[4], l * S.hist[s]);
  }
  __syncthreads();
  for (int k = 0, s = t; s < kNumLit; s += kT, ++k)
    if (mycode[k]) S.code[s] = mycode[k];
  __syncthreads();

  // ---- dynamic block or stored blocks: whichever is smaller (the coded size is exact) ----
  const unsigned nlit = max(257u, S.misc[6]);
  const unsigned hdr_bits = 3 + 5 + 5 + 4 + 19 * 3 + 4 * (nlit + 1);
  const unsigned long long coded_bits = (unsigned long long)hdr_bits + S.misc[4] + S.misc[2] + S.misc[3];
  const unsigned long long dyn_bytes = (coded_bits + (last ? 0 : 3) + 7) / 8 + (last ? 0 : 4);
  const unsigned pieces = (n + kStoredMax - 1) / kStoredMax;
  const unsigned long long stored_bytes = (unsigned long long)n + 5ull * pieces;
  const bool stored = dyn_bytes >= stored_bytes;
  if (t == 0) {
    meta[b].s1 = S.misc[0] % kAdler;
    meta[b].s2 = S.misc[1] % kAdler;
    meta[b].stored = stored ? 1u : 0u;
    meta[b].bytes = (unsigned)(stored ? stored_bytes : dyn_bytes);
  }
  if (stored) {
    uint8_t* ob = reinterpret_cast<uint8_t*>(gout);
    for (unsigned i = t; i < n; i += kT) ob[5u * (i / kStoredMax + 1u) + i] = (uint8_t)filtered_byte(bgr, G, y0, i);
    for (unsigned q = t; q < pieces; q += kT) {
      uint8_t* hp = ob + (unsigned long long)q * (kStoredMax + 5u);
      const unsigned len = min(kStoredMax, n - q * kStoredMax);
      hp[0] = (last && q == pieces - 1) ? 1 : 0;  // BFINAL, BTYPE 00, padding
      hp[1] = (uint8_t)len; hp[2] = (uint8_t)(len >> 8);
      hp[3] = (uint8_t)~len; hp[4] = (uint8_t)(~len >> 8);
    }
    return;
  }

  // ---- pass 2: the block header, then the tokens at their bit offsets ----
  // header (RFC 1951, 3.2.7): BFINAL, BTYPE = 2, HLIT, HDIST = 0 (one distance code), HCLEN = 15: all 19 code-length codes —
  // the lengths 0..15 get 4 bits each (a complete code), the repeat codes 16..18 none, so every length below is 4 bits
  if (t == 0) {
    BitWriter W(S, 0);
    W.put(last ? 1u : 0u, 1);
    W.put(2u, 2);
    W.put(nlit - 257u, 5);
    W.put(0u, 5);
    W.put(15u, 4);
    for (int i = 0; i < 19; ++i) W.put(i < 3 ? 0u : 4u, 3);  // order 16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15
    W.finish();
  }
  __syncthreads();
  {
    const unsigned at = 3 + 5 + 5 + 4 + 19 * 3;
    for (unsigned s = t; s <= nlit; s += kT) {  // s == nlit: the distance code's length (1 if any match exists)
      const unsigned l = s < nlit ? S.code[s] >> 16 : (S.misc[3] ? 1u : 0u);
      BitWriter W(S, at + 4 * s);
      W.put(rev_bits(l, 4), 4);
      W.finish();
    }
  }
  __syncthreads();
  unsigned gw = 0, carry = 0;
  if (t < kWaves) S.scan[t] = t == 0 ? hdr_bits : 0u;  // (the header sits in region 0)
  __syncthreads();
  unsigned pend = compact_flush(S, gout, &gw, 0u, &carry);
  for (int r = 0; r < rows; ++r)
    for (int tl = 0; tl < tiles; ++tl) {
      const int px0 = tl * kTilePx, npx = min(kTilePx, G.w - px0), nbytes = 3 * npx;
      const int off = load_tile(S, bgr, G.total_bytes, G.w, y0 + r, px0);
      __syncthreads();
      const int ng = (nbytes + 63) >> 6, per = (ng + kWaves - 1) / kWaves;
      const int g0 = wv * per, g1 = min(g0 + per, ng);
      const bool lead = tl == 0 && wv == 0;  // this wave writes the row's filter-type byte in front of its groups
      const unsigned base = (unsigned)wv * kRegionWords * 32u;  // the wave's region, in bits of S.out
      unsigned at = 0;
      if (lead) {
        const unsigned leadc = S.code[1];
        if (lane == 0) or_bits(S, base, leadc & 0xffffu);
        at = leadc >> 16;
      }
      for (int g = g0; g < g1; ++g) {
        const Tok k = group_tok(rawb, off, px0, nbytes, 64 * g + lane);
        const unsigned bits = tok_bits(S, k);
        const unsigned incl = wave_scan(bits);
        if (k.r >= 0) {  // this lane's tokens as one value: the literal, then the match or the literal again
          const unsigned cl = S.code[k.v], len = cl >> 16;
          unsigned long long val = cl & 0xffffu;
          unsigned n2 = len;
          if (k.r < 3) {
            for (int q = 0; q < k.r; ++q) { val |= (unsigned long long)(cl & 0xffffu) << n2; n2 += len; }
          } else {
            int sym, e, ev;
            length_code(k.r, sym, e, ev);
            const unsigned cm = S.code[sym];
            val |= (unsigned long long)(cm & 0xffffu) << n2;
            n2 += cm >> 16;
            val |= (unsigned long long)(unsigned)ev << n2;  // the length's extra bits; the distance symbol behind them is a 0 bit
          }
          or_bits(S, base + at + incl - bits, val);
        }
        at += (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
      }
      if (lane == 0) S.scan[wv] = at;
      __syncthreads();
      pend = compact_flush(S, gout, &gw, pend, &carry);
    }
  // end of block; every band but the last is closed like zlib's sync flush: an empty stored block, which byte-aligns
  if (t == 0) {
    BitWriter W(S, 0);  // (region 0 from its first bit: compact_flush puts it behind the `pend` bits carried over)
    W.put(S.code[256] & 0xffffu, (int)(S.code[256] >> 16));
    if (!last) {
      W.put(0u, 3);
      const unsigned p = pend + W.pos();  // position in the band's stream, modulo 32
      if (p & 7u) W.put(0u, (int)(8u - (p & 7u)));
      W.put(0x0000u, 16);
      W.put(0xffffu, 16);
    }
    W.finish();
    S.scan[0] = W.pos();
  } else if (t < kWaves) {
    S.scan[t] = 0u;
  }
  __syncthreads();
  pend = compact_flush(S, gout, &gw, pend, &carry);
  if (t == 0 && pend) gout[gw] = carry;  // (the last band's final bits; the band's reserve covers the dword)
}

// where every band's IDAT chunk starts in the file: prefix sum of 12 + bytes behind the preamble (one workgroup)
__global__ __launch_bounds__(kT) void k_png_layout(PngBandMeta* __restrict__ meta, int nbands) {
  __shared__ unsigned long long part[kT];
  const int t = threadIdx.x, per = (nbands + kT - 1) / kT;
  const int i0 = min(t * per, nbands), i1 = min(i0 + per, nbands);
  unsigned long long s = 0;
  for (int i = i0; i < i1; ++i) s += 12ull + meta[i].bytes;
  part[t] = s;
  __syncthreads();
  if (t == 0) {
    unsigned long long run = kPngPreamble;
    for (int k = 0; k < kT; ++k) { const unsigned long long v = part[k]; part[k] = run; run += v; }
    meta[nbands].file_off = run;
    meta[nbands].bytes = 0;
  }
  __syncthreads();
  unsigned long long at = part[t];
  for (int i = i0; i < i1; ++i) { meta[i].file_off = at; at += 12ull + meta[i].bytes; }
}

// band b to its place in the file: length (big-endian) + "IDAT" + the bytes; the chunk's CRC stays for the host
__global__ __launch_bounds__(kT) void k_png_gather(const uint8_t* __restrict__ scratch, unsigned long long band_stride,
                                                    const PngBandMeta* __restrict__ meta, uint8_t* __restrict__ file) {
  const int t = threadIdx.x, b = blockIdx.x;
  const unsigned bytes = meta[b].bytes;
  uint8_t* dst = file + meta[b].file_off;
  if (t < 8) {
    const uint8_t hdr[8] = {(uint8_t)(bytes >> 24), (uint8_t)(bytes >> 16), (uint8_t)(bytes >> 8), (uint8_t)bytes, 'I', 'D', 'A', 'T'};
    dst[t] = hdr[t];
  }
  dst += 8;
  const uint8_t* src = scratch + (unsigned long long)b * band_stride;  // 16-byte aligned
  const unsigned head = min(bytes, (unsigned)((4u - (unsigned)(reinterpret_cast<uintptr_t>(dst) & 3u)) & 3u));
  if ((unsigned)t < head) dst[t] = src[t];
  const unsigned nw = (bytes - head) >> 2;
  const unsigned* sw = reinterpret_cast<const unsigned*>(src);
  unsigned* dw = reinterpret_cast<unsigned*>(dst + head);
  const unsigned sh = 8u * head;
  for (unsigned i = t; i < nw; i += kT) dw[i] = sh ? (sw[i] >> sh) | (sw[i + 1] << (32u - sh)) : sw[i];
  const unsigned done = head + 4u * nw;
  if (done + (unsigned)t < bytes) dst[done + t] = src[done + t];
}

inline size_t cdivz(size_t a, size_t b) { return (a + b - 1) / b; }
}  // namespace

PngPlan PngPlan::make(int w, int h) {
  if (w < 1 || h < 1 || w > 65535 || h > 65535) throw Error(S360_ERR_INVALID_ARG, "png: unsupported image size");
  PngPlan p;
  p.w = w;
  p.h = h;
  p.line = 1 + 3 * (size_t)w;
  // ~192 KB of scanlines per band, at least ~64 bands in a tall image (small frames still spread over the chip)
  const size_t by_size = std::max<size_t>(1, ((size_t)192 << 10) / p.line), by_count = cdivz((size_t)h, 64);
  p.rows_per_band = (int)std::max<size_t>(1, std::min(std::min(by_size, by_count), (size_t)h));
  if (const char* e = std::getenv("S360_PNG_BAND_ROWS"))  // (developer switch: band height)
    if (std::atoi(e) > 0) p.rows_per_band = std::min(std::atoi(e), h);
  p.nbands = (int)cdivz((size_t)h, (size_t)p.rows_per_band);
  const size_t n = (size_t)p.rows_per_band * p.line;
  p.band_stride = (n + 5 * cdivz(n, kStoredMax) + 64 + 15) & ~(size_t)15;
  p.file_bound = kPngPreamble + (size_t)p.nbands * (12 + n + 5 * cdivz(n, kStoredMax) + 8) + 16 + 12 + 64;
  return p;
}

void png_encode_enqueue(hipStream_t st, const uint8_t* bgr, const PngPlan& P, DevBuf& scratch, DevBuf& meta, uint8_t* file) {
  if (reinterpret_cast<uintptr_t>(bgr) & 3) throw Error(S360_ERR_INVALID_ARG, "png: image not 4-byte aligned");
  scratch.ensure((size_t)P.nbands * P.band_stride);
  meta.ensure(((size_t)P.nbands + 1) * sizeof(PngBandMeta));
  Geo G;
  G.w = P.w; G.h = P.h; G.rows_per_band = P.rows_per_band; G.nbands = P.nbands;
  G.line = (unsigned)P.line;
  G.band_stride = P.band_stride;
  G.total_bytes = (unsigned long long)P.w * P.h * 3;
  hipLaunchKernelGGL(k_png_band, dim3(P.nbands), dim3(kT), 0, st, bgr, G, scratch.as<uint8_t>(), meta.as<PngBandMeta>());
  hipLaunchKernelGGL(k_png_layout, dim3(1), dim3(kT), 0, st, meta.as<PngBandMeta>(), P.nbands);
  hipLaunchKernelGGL(k_png_gather, dim3(P.nbands), dim3(kT), 0, st, scratch.as<uint8_t>(), (unsigned long long)P.band_stride,
                     meta.as<PngBandMeta>(), file);
  S360_HIP(hipGetLastError());
}

// ---- host side: CRC-32 (the PNG / zlib polynomial 0xEDB88320, eight tables, eight bytes per step) ----
namespace {
struct CrcTables {
  uint32_t t[8][256];
  CrcTables() {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      t[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int k = 1; k < 8; ++k) t[k][i] = t[0][t[k - 1][i] & 255u] ^ (t[k - 1][i] >> 8);
  }
};
const CrcTables& crc_tables() {
  static const CrcTables T;
  return T;
}
inline void be32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }
size_t put_chunk(uint8_t* at, const char* type, const uint8_t* data, size_t len) {
  be32(at, (uint32_t)len);
  std::memcpy(at + 4, type, 4);
  if (len) std::memcpy(at + 8, data, len);
  be32(at + 8 + len, crc32_update(0u, at + 4, 4 + len));
  return 12 + len;
}
}  // namespace

uint32_t crc32_update(uint32_t crc, const uint8_t* p, size_t n) {
  const CrcTables& T = crc_tables();
  uint32_t c = ~crc;
  while (n && (reinterpret_cast<uintptr_t>(p) & 7)) { c = T.t[0][(c ^ *p++) & 255u] ^ (c >> 8); --n; }
  while (n >= 8) {
    uint32_t lo, hi;
    std::memcpy(&lo, p, 4);
    std::memcpy(&hi, p + 4, 4);
    lo ^= c;
    c = T.t[7][lo & 255u] ^ T.t[6][(lo >> 8) & 255u] ^ T.t[5][(lo >> 16) & 255u] ^ T.t[4][lo >> 24] ^
        T.t[3][hi & 255u] ^ T.t[2][(hi >> 8) & 255u] ^ T.t[1][(hi >> 16) & 255u] ^ T.t[0][hi >> 24];
    p += 8;
    n -= 8;
  }
  while (n--) c = T.t[0][(c ^ *p++) & 255u] ^ (c >> 8);
  return ~c;
}

size_t png_finish_host(uint8_t* file, size_t cap, const PngPlan& P, const PngBandMeta* meta, int crc_threads) {
  const size_t end_bands = (size_t)meta[P.nbands].file_off;
  if (end_bands + 16 + 12 > cap) throw Error(S360_ERR_INVALID_ARG, "png: output buffer too small");
  static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
  std::memcpy(file, sig, 8);
  size_t at = 8;
  uint8_t ihdr[13];
  be32(ihdr, (uint32_t)P.w);
  be32(ihdr + 4, (uint32_t)P.h);
  ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;  // 8-bit RGB, deflate, adaptive filtering, not interlaced
  at += put_chunk(file + at, "IHDR", ihdr, 13);
  uint8_t br[4];
  be32(br, (uint32_t)P.rows_per_band);
  at += put_chunk(file + at, "sbNd", br, 4);  // (host/png_io.hpp: independent Sub-filtered bands of this many rows)
  static const uint8_t zhdr[2] = {0x78, 0x01};
  at += put_chunk(file + at, "IDAT", zhdr, 2);
  if (at != kPngPreamble) throw Error(S360_ERR_STATE, "png: preamble size");
  // CRC of every band chunk (type + data)
  const int nt = std::max(1, std::min(crc_threads, P.nbands));
  auto work = [&](int k) {
    for (int b = k; b < P.nbands; b += nt) {
      uint8_t* c = file + meta[b].file_off;
      be32(c + 8 + meta[b].bytes, crc32_update(0u, c + 4, 4 + (size_t)meta[b].bytes));
    }
  };
  std::vector<std::thread> th;
  for (int k = 1; k < nt; ++k) th.emplace_back(work, k);
  work(0);
  for (auto& x : th) x.join();
  // Adler-32 of all filtered scanlines from the bands' pieces: A = 1 + sum of bytes, B = N + sum over bytes of (N - index) * byte
  unsigned long long A = 1, Bs = 0, after = 0;
  const unsigned long long N = (unsigned long long)P.h * P.line;
  for (int b = P.nbands - 1; b >= 0; --b) {  // `after`: filtered bytes behind band b
    const unsigned long long nb = (unsigned long long)std::min(P.rows_per_band, P.h - b * P.rows_per_band) * P.line;
    A = (A + meta[b].s1) % kAdler;
    Bs = (Bs + meta[b].s2 + (after % kAdler) * meta[b].s1) % kAdler;
    after += nb;
  }
  Bs = (Bs + N % kAdler) % kAdler;
  uint8_t ad[4];
  be32(ad, (uint32_t)((Bs << 16) | A));
  at = end_bands;
  at += put_chunk(file + at, "IDAT", ad, 4);
  at += put_chunk(file + at, "IEND", nullptr, 0);
  return at;
}

}  // namespace s360
