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
//   pass 1  the band's B,G,R rows tile by tile through LDS; every thread filters (Sub) and tokenises 8 pixels — literals
//           and distance-1 matches, i.e. Z_RLE's token set, runs cut at the 24-byte chunk — into the band's histogram;
//           the Adler-32 pieces of the filtered bytes on the way
//   build   length-limited canonical Huffman code of the literal/length alphabet: rank sort by all threads, the
//           two-queue merge and the 15-bit limit by one thread, canonical codes by all threads; the exact coded size is
//           then known, and a band that would not shrink goes out as stored blocks instead
//   pass 2  the same tiles again: bits per chunk, a workgroup prefix sum, every thread ORs its tokens into an LDS bit
//           buffer at its offset, whole dwords go to HBM coalesced
// then a layout kernel (prefix sum of the bands' sizes = where each IDAT chunk starts in the file) and a gather kernel that
// moves every band to its place with the chunk's length and type in front. The host adds what needs no pixel: signature,
// IHDR, CRC-32 of every chunk (threads), the combined Adler-32, IEND.
// Algorithmic bytes per 8K frame: 201 MB read twice (the second time mostly from L2) + the compressed size written twice —
// 0.1 ms of HBM time. What the band kernel is actually bound by is the latency of its per-token chains (a table read, a 64-bit
// shift-and-or, an LDS atomic per token) at the occupancy its 53 KB of LDS allow. Measured on an 8K frame of the bench
// (tools/png_time.py, round 6): 1.95 ms with 256 threads x 16 pixels (phase cuts: pass 1 0.55, code build 0.18, pass 2 1.32);
// privatised histograms, a two-barrier scan and tile loads in flight together changed nothing measurable; 512 threads x 8
// pixels — twice the waves per CU for the same LDS — 1.49 ms, files 1 % larger (runs cut at 24 bytes instead of 48). The design that
// would remove the chains — a lane per byte, tokens from ballots, bit offsets from a wave scan — was not built.
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
constexpr int kChunkPx = 8;                     // pixels one thread tokenises per tile (a multiple of 4: three dwords)
constexpr int kChunkBytes = kChunkPx * 3 + 1;   // ... and the most filtered bytes that is (a row's first chunk carries the filter-type byte)
constexpr int kTilePx = kT * kChunkPx;          // pixels of one row per workgroup iteration
constexpr int kRawWords = kTilePx * 3 / 4 + 4;  // a tile's bytes + the pixel to its left + alignment slack
constexpr int kMaxBits = 15;                    // deflate's longest code
constexpr int kOutWords = (kT * kChunkBytes * kMaxBits + 31) / 32 + 8;
constexpr int kSyms = 288;                      // literal/length alphabet, padded (286 symbols exist)
constexpr int kNumLit = 286;
constexpr int kChunkDw = kChunkPx * 3 / 4;       // dwords of a whole chunk
constexpr int kChunkB = kChunkPx * 3;            // ... and its bytes
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

// The tokens of one chunk — `lead`: the row's filter-type byte (1 = Sub) first, then pixels [cpx, cpx + npx) as R,G,B bytes
// minus the same channel of the pixel to the left (0 at the row's start) — in stream order. Z_RLE's token set: a byte equal to
// its predecessor extends a run; a run of >= 3 repeats leaves as ONE distance-1 match, a shorter one as literals.
template <class Sink>
__device__ inline void chunk_tokens(const uint8_t* rawb, int idx0 /* byte of pixel cpx's B in rawb */, int cpx, int npx, bool lead,
                                    Sink& sink) {
  int prev = -1, run = 0;
  if (lead) { sink.byte(1); sink.lit(1); prev = 1; }
  for (int p = 0; p < npx; ++p) {
    const int i = idx0 + 3 * p;
    const bool first = cpx + p == 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {  // PNG order R,G,B out of B,G,R
      const int v = (rawb[i + 2 - c] - (first ? 0 : rawb[i - 1 - c])) & 255;
      sink.byte(v);
      if (v == prev) { ++run; continue; }
      if (run >= 3) sink.match(run); else for (int r = 0; r < run; ++r) sink.lit(prev);
      run = 0;
      sink.lit(v);
      prev = v;
    }
  }
  if (run >= 3) sink.match(run); else for (int r = 0; r < run; ++r) sink.lit(prev);
}

// ---- the fast path of a chunk: 16 whole pixels behind the row's first one, no run of three repeats --------------------------
// (the chunk's filtered bytes are made 4 at a time instead of by byte loops with two LDS byte reads each: the chunk's LDS dwords,
// funnel shifts to the chunk's alignment, v_perm_b32 from B,G,R to the stream's R,G,B, the left pixel = the same
// stream three bytes earlier, a byte-wise subtraction in a dword; the run logic works on one repeat flag per byte. A chunk none of
// whose bytes repeats three times is literals only — one byte-field extract + one table access per byte and pass —, any other takes
// one loop turn per literal-and-run from the flags.)
__device__ inline unsigned funnel(unsigned lo, unsigned hi, unsigned sh) {  // the dword at byte offset sh (0..3) of the 8 bytes lo, hi
  return (unsigned)((((unsigned long long)hi << 32) | lo) >> (8u * sh));
}
__device__ inline unsigned sub_bytes(unsigned a, unsigned b) {  // a - b per byte, modulo 256
  const unsigned H = 0x80808080u;
  return ((a | H) - (b & ~H)) ^ ((a ^ ~b) & H);
}
__device__ inline unsigned zero_bytes(unsigned v) { return (v - 0x01010101u) & ~v & 0x80808080u; }  // != 0 iff v has a zero byte
// F = the chunk's filtered bytes in stream order; idx0 = byte offset of the chunk's first pixel in raw (its left neighbour's three
// bytes in front of it). Returns the repeat flags: bit j set = byte j equals byte j - 1 (bit 0 never).
__device__ inline unsigned long long chunk_filtered(const unsigned* __restrict__ raw, int idx0, unsigned (&F)[kChunkDw]) {
  const int q0 = (idx0 >> 2) - 1;
  const unsigned sh = (unsigned)idx0 & 3u;
  unsigned M[kChunkDw + 2], Sd[kChunkDw + 1];  // Sd[k + 1] = the bytes idx0 + 4 k .. + 3, k = -1 .. kChunkDw - 1
#pragma unroll
  for (int q = 0; q < kChunkDw + 2; ++q) M[q] = raw[max(q0 + q, 0)];
#pragma unroll
  for (int k = 0; k < kChunkDw + 1; ++k) Sd[k] = funnel(M[k], M[k + 1], sh);
  unsigned T[kChunkDw + 1];  // T[k + 1] = stream bytes 4 k .. 4 k + 3; T[0]: the left pixel's R,G,B in bytes 1..3
  T[0] = __builtin_amdgcn_perm(Sd[0], Sd[0], 0x01020300u);
#pragma unroll
  for (int m = 0; m < kChunkPx / 4; ++m) {  // four pixels = 12 bytes = three dwords at a time
    const unsigned a = Sd[3 * m + 1], b = Sd[3 * m + 2], c = Sd[3 * m + 3];
    const unsigned X = funnel(a, b, 3), Y = funnel(b, c, 3);
    T[3 * m + 1] = __builtin_amdgcn_perm(b, a, 0x05000102u);
    T[3 * m + 2] = __builtin_amdgcn_perm(Y, X, 0x04050001u);
    T[3 * m + 3] = __builtin_amdgcn_perm(c, b, 0x05060702u);
  }
#pragma unroll
  for (int k = 0; k < kChunkDw; ++k) F[k] = sub_bytes(T[k + 1], funnel(T[k], T[k + 1], 1));
  // x = every byte xor its predecessor (the first byte has none: made to differ): a zero byte = a repeat. The repeat flags are
  // gathered into one bit per byte (exact per-byte zero test, then the four flag bits of a dword multiplied together into a nibble).
  unsigned long long eq = 0;
#pragma unroll
  for (int k = 0; k < kChunkDw; ++k) {
    const unsigned x = F[k] ^ funnel(k ? F[k - 1] : ~F[0] << 24, F[k], 3);
    const unsigned z = ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu);  // 0x80 in every byte of x that is zero
    eq |= (unsigned long long)((((z >> 7) * 0x00204081u) >> 21) & 15u) << (4 * k);
  }
  return eq;
}
// byte s (not known at compile time) of the chunk's dwords: a chain of selects (registers cannot be indexed by a lane's value)
__device__ inline unsigned chunk_byte(const unsigned (&F)[kChunkDw], int s) {
  const int k = s >> 2;
  unsigned d = F[0];
#pragma unroll
  for (int q = 1; q < kChunkDw; ++q) d = k == q ? F[q] : d;
  return (d >> (8 * (s & 3))) & 255u;
}
// The tokens of a whole chunk from its repeat flags: every byte that does not repeat its predecessor is a literal, the r repeats
// behind it one distance-1 match (r >= 3) or r literals — chunk_tokens' sequence, with one loop turn per literal-and-run instead of
// one per byte, and no LDS read on the way.
template <class Sink>
__device__ inline void chunk_tokens_eq(const unsigned (&F)[kChunkDw], unsigned long long eq, Sink& sink) {
  unsigned long long starts = ~eq & ((1ull << kChunkB) - 1ull);
  while (starts) {
    const int s = __ffsll((long long)starts) - 1;
    starts &= starts - 1;
    const int r = (starts ? __ffsll((long long)starts) - 1 : kChunkB) - s - 1;
    const int v = (int)chunk_byte(F, s);
    sink.lit(v);
    if (r >= 3) sink.match(r);
    else for (int i = 0; i < r; ++i) sink.lit(v);
  }
}
#define S360_PNG_BYTE(F, j) (((F)[(j) >> 2] >> (8 * ((j)&3))) & 255u)

struct HistSink {
  unsigned* H;  // this lane's copy of the histogram
  unsigned sum = 0, wsum = 0, j = 0, ebits = 0, nmatch = 0;
  __device__ void byte(int v) { sum += v; wsum += j * v; ++j; }
  __device__ void lit(int v) { atomicAdd(&H[v], 1u); }
  __device__ void match(int L) {
    int sym, eb, ev;
    length_code(L, sym, eb, ev);
    atomicAdd(&H[sym], 1u);
    ebits += eb;
    ++nmatch;
  }
};
struct CountSink {
  const Smem& S;
  unsigned bits = 0;
  __device__ void byte(int) {}
  __device__ void lit(int v) { bits += S.code[v] >> 16; }
  __device__ void match(int L) {
    int sym, eb, ev;
    length_code(L, sym, eb, ev);
    bits += (S.code[sym] >> 16) + eb + 1;  // + the one-bit distance code
  }
};
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
struct EmitSink {
  BitWriter& W;
  __device__ void byte(int) {}
  __device__ void lit(int v) { const unsigned c = W.S.code[v]; W.put(c & 0xffffu, (int)(c >> 16)); }
  __device__ void match(int L) {
    int sym, eb, ev;
    length_code(L, sym, eb, ev);
    const unsigned c = W.S.code[sym];
    W.put(c & 0xffffu, (int)(c >> 16));
    if (eb) W.put((unsigned)ev, eb);
    W.put(0u, 1);  // distance symbol 0 (distance 1), the only distance code: one bit
  }
};

// exclusive prefix sum over the workgroup's threads; *total = the sum. Inside a wave by lane shuffles (ds_bpermute: lane - d), the four
// waves' totals through LDS: two barriers (as a Hillis-Steele scan in LDS it was seventeen, sixteen times per band).
__device__ inline unsigned block_scan(Smem& S, unsigned v, unsigned* total) {
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  unsigned incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned up = (unsigned)__builtin_amdgcn_ds_bpermute(4 * max(lane - d, 0), (int)incl);
    if (lane >= d) incl += up;
  }
  if (lane == 63) S.scan[wv] = incl;
  __syncthreads();
  unsigned base = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < kT / 64; ++k) {
    const unsigned w = S.scan[k];
    if (k < wv) base += w;
    tot += w;
  }
  *total = tot;
  __syncthreads();
  return base + incl - v;
}

// S.out holds `total` bits from bit 0: the whole dwords go to gout[*gw ..], the rest moves to the buffer's front. (The callers'
// next barrier — behind the next tile's load, or the explicit one in front of the epilogue — orders the refill after the zeroing.)
__device__ inline unsigned flush_words(Smem& S, unsigned* gout, unsigned* gw, unsigned total) {
  const int full = (int)(total >> 5);
  for (int i = threadIdx.x; i < full; i += kT) gout[*gw + i] = S.out[i];
  const unsigned carry = S.out[full];
  __syncthreads();
  for (int i = threadIdx.x; i <= full; i += kT) S.out[i] = i == 0 ? carry : 0u;
  *gw += (unsigned)full;
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
  {
    unsigned s1 = 0, s2 = 0, eb = 0, nm = 0;
    for (int r = 0; r < rows; ++r)
      for (int tl = 0; tl < tiles; ++tl) {
        const int off = load_tile(S, bgr, G.total_bytes, G.w, y0 + r, tl * kTilePx);
        __syncthreads();
        const int cpx = tl * kTilePx + t * kChunkPx, npx = min(kChunkPx, G.w - cpx);
        unsigned F[kChunkDw];
        if (npx == kChunkPx && cpx > 0) {  // a whole chunk behind the row's first pixel: from registers
          const unsigned long long eq = chunk_filtered(S.raw, off + 3 * t * kChunkPx, F);
          unsigned* H = S.histp[t % kHistCopies];
          unsigned sum = 0, wsum = 0;
#pragma unroll
          for (int k = 0; k < kChunkDw; ++k) {
            sum = __builtin_amdgcn_sad_u8(F[k], 0u, sum);  // + the four bytes
            wsum = __builtin_amdgcn_udot4(F[k], 0x03020100u + 0x04040404u * (unsigned)k, wsum, false);  // + sum of index x byte
          }
          if (!(eq & (eq >> 1) & (eq >> 2))) {  // no three repeats in a row: literals only
#pragma unroll
            for (int j = 0; j < kChunkB; ++j) atomicAdd(&H[S360_PNG_BYTE(F, j)], 1u);
          } else {
            HistSink hs{H};
            chunk_tokens_eq(F, eq, hs);
            eb += hs.ebits;
            nm += hs.nmatch;
          }
          const unsigned g = (unsigned)r * G.line + 1u + 3u * (unsigned)cpx;
          s1 = (s1 + sum) % kAdler;
          const unsigned long long wgt = (unsigned long long)(n - g) * sum - wsum;
          s2 = (unsigned)((s2 + wgt % kAdler) % kAdler);
        } else if (npx > 0) {
          HistSink hs{S.histp[t % kHistCopies]};
          chunk_tokens(rawb, off + 3 * t * kChunkPx, cpx, npx, cpx == 0, hs);
          // bytes [g, g + j) of the band: sum += d, weighted sum += (n - (g + i)) d_i
          const unsigned g = (unsigned)r * G.line + (cpx ? 1u + 3u * (unsigned)cpx : 0u);
          s1 = (s1 + hs.sum) % kAdler;
          const unsigned long long wgt = (unsigned long long)(n - g) * hs.sum - hs.wsum;
          s2 = (unsigned)((s2 + wgt % kAdler) % kAdler);
          eb += hs.ebits;
          nm += hs.nmatch;
        }
        __syncthreads();
      }
    atomicAdd(&S.misc[0], s1);
    atomicAdd(&S.misc[1], s2);
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
  unsigned gw = 0;
  unsigned pend = flush_words(S, gout, &gw, hdr_bits);
  for (int r = 0; r < rows; ++r)
    for (int tl = 0; tl < tiles; ++tl) {
      const int off = load_tile(S, bgr, G.total_bytes, G.w, y0 + r, tl * kTilePx);
      __syncthreads();
      const int cpx = tl * kTilePx + t * kChunkPx, npx = min(kChunkPx, G.w - cpx);
      unsigned bits = 0;
      unsigned F[kChunkDw];
      const bool whole = npx == kChunkPx && cpx > 0;
      const unsigned long long eq = whole ? chunk_filtered(S.raw, off + 3 * t * kChunkPx, F) : 0ull;
      const bool fast = whole && !(eq & (eq >> 1) & (eq >> 2));  // literals only
      if (fast) {
#pragma unroll
        for (int j = 0; j < kChunkB; ++j) bits += S.code[S360_PNG_BYTE(F, j)] >> 16;
      } else if (whole) {
        CountSink cs{S};
        chunk_tokens_eq(F, eq, cs);
        bits = cs.bits;
      } else if (npx > 0) {
        CountSink cs{S};
        chunk_tokens(rawb, off + 3 * t * kChunkPx, cpx, npx, cpx == 0, cs);
        bits = cs.bits;
      }
      unsigned tot;
      const unsigned excl = block_scan(S, bits, &tot);
      if (fast) {
        BitWriter W(S, pend + excl);
#pragma unroll
        for (int j = 0; j < kChunkB; ++j) {
          const unsigned c = S.code[S360_PNG_BYTE(F, j)];
          W.put(c & 0xffffu, (int)(c >> 16));
        }
        W.finish();
      } else if (whole) {
        BitWriter W(S, pend + excl);
        EmitSink es{W};
        chunk_tokens_eq(F, eq, es);
        W.finish();
      } else if (npx > 0) {
        BitWriter W(S, pend + excl);
        EmitSink es{W};
        chunk_tokens(rawb, off + 3 * t * kChunkPx, cpx, npx, cpx == 0, es);
        W.finish();
      }
      __syncthreads();
      pend = flush_words(S, gout, &gw, pend + tot);
    }
  // end of block; every band but the last is closed like zlib's sync flush: an empty stored block, which byte-aligns
  __syncthreads();  // (the last flush has finished clearing the bit buffer)
  if (t == 0) {
    BitWriter W(S, pend);
    W.put(S.code[256] & 0xffffu, (int)(S.code[256] >> 16));
    if (!last) {
      W.put(0u, 3);
      const unsigned p = W.pos();
      if (p & 7u) W.put(0u, (int)(8u - (p & 7u)));
      W.put(0x0000u, 16);
      W.put(0xffffu, 16);
    }
    W.finish();
    S.misc[7] = W.pos();
  }
  __syncthreads();
  const unsigned endbits = S.misc[7];
  for (int i = t; i < (int)((endbits + 31) >> 5); i += kT) gout[gw + i] = S.out[i];  // (the band's reserve covers the last dword)
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
