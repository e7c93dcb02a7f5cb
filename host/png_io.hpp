// png_io.hpp — minimal PNG codec on zlib for the host binary (the reference uses cv::imread / cv::imwrite,
// RigDescription.cpp:87-105, TRSP:961). Reads every PNG colour type and bit depth (1..16 bits, Adam7 interlace) into
// 8-bit B,G,R(,A) like imread's 8-bit decode; writes 8-bit RGB / RGBA (and the ISP's 16-bit RGB) and fails loudly on a short write; read_gray keeps the depth of 8-/16-bit greyscale raw images. Pixel order at this interface is OpenCV's: B,G,R(,A).
#pragma once
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <new>
#include <thread>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace pngio {

// Where decoded pixels live. A streaming host that feeds a GPU sets the two hooks ONCE, before its first image, to a
// page-locked allocator (s360_host_alloc / s360_host_free): the decoders then write straight into memory the upload can
// DMA from, and the library's staging copy (a memcpy of 214 MB per 8K frame on the thread that feeds the GPU) is gone.
inline void* (*g_pixel_alloc)(size_t) = nullptr;
inline void (*g_pixel_free)(void*) = nullptr;
template <class T>
struct PixelAlloc {
  using value_type = T;
  PixelAlloc() = default;
  template <class U> PixelAlloc(const PixelAlloc<U>&) {}
  T* allocate(size_t n) {
    void* p = g_pixel_alloc ? g_pixel_alloc(n * sizeof(T)) : std::malloc(n * sizeof(T));
    if (!p) throw std::bad_alloc();
    return static_cast<T*>(p);
  }
  void deallocate(T* p, size_t) { if (g_pixel_free) g_pixel_free(p); else std::free(p); }
  template <class U> bool operator==(const PixelAlloc<U>&) const { return true; }
  template <class U> bool operator!=(const PixelAlloc<U>&) const { return false; }
};
typedef std::vector<uint8_t, PixelAlloc<uint8_t>> Pixels;
struct Image {
  int w = 0, h = 0, c = 0;  // c = 3 (BGR) or 4 (BGRA)
  Pixels px;
};

inline uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }
inline void put32(uint8_t* p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; }
inline int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// Paeth-filtered scanline of 3- or 4-byte pixels, undone in place. A pixel's bytes are predicted together in 16-bit SSE2
// lanes (the chain from pixel to pixel stays serial): 2048 x 2048 RGB in 14 ms instead of 45. cur and prev must be
// readable for 4 bytes from their last pixel (the callers' buffers are padded).
#if defined(__SSE2__)
}  // namespace pngio
#include <emmintrin.h>
namespace pngio {
template <int BPP>
inline void unfilter_paeth_px(uint8_t* cur, const uint8_t* prev, size_t stride) {
  const __m128i zero = _mm_setzero_si128(), low = _mm_set1_epi16(0x00ff);
  __m128i a = zero, c = zero;
  for (size_t i = 0; i + BPP <= stride; i += BPP) {
    int xb, pbits;
    std::memcpy(&xb, cur + i, 4);
    std::memcpy(&pbits, prev + i, 4);
    const __m128i b = _mm_unpacklo_epi8(_mm_cvtsi32_si128(pbits), zero), x = _mm_unpacklo_epi8(_mm_cvtsi32_si128(xb), zero);
    const __m128i da = _mm_sub_epi16(b, c), db = _mm_sub_epi16(a, c), dc = _mm_add_epi16(da, db);  // p - a, p - b, p - c
    const __m128i pa = _mm_max_epi16(da, _mm_sub_epi16(zero, da)), pb = _mm_max_epi16(db, _mm_sub_epi16(zero, db)),
                  pc = _mm_max_epi16(dc, _mm_sub_epi16(zero, dc));
    const __m128i smallest = _mm_min_epi16(pc, _mm_min_epi16(pa, pb));
    const __m128i isa = _mm_cmpeq_epi16(smallest, pa), isb = _mm_cmpeq_epi16(smallest, pb);
    const __m128i bc = _mm_or_si128(_mm_and_si128(isb, b), _mm_andnot_si128(isb, c));
    const __m128i pred = _mm_or_si128(_mm_and_si128(isa, a), _mm_andnot_si128(isa, bc));
    const __m128i d = _mm_and_si128(_mm_add_epi16(x, pred), low);
    const int v = _mm_cvtsi128_si32(_mm_packus_epi16(d, d));
    cur[i] = (uint8_t)v; cur[i + 1] = (uint8_t)(v >> 8); cur[i + 2] = (uint8_t)(v >> 16);
    if (BPP == 4) cur[i + 3] = (uint8_t)(v >> 24);
    a = d;
    c = b;
  }
}
#else
template <int BPP>
inline void unfilter_paeth_px(uint8_t* cur, const uint8_t* prev, size_t stride) {
  for (size_t i = 0; i < std::min<size_t>(BPP, stride); ++i) cur[i] = (uint8_t)(cur[i] + prev[i]);
  for (size_t i = BPP; i < stride; ++i) cur[i] = (uint8_t)(cur[i] + paeth(cur[i - BPP], prev[i], prev[i - BPP]));
}
#endif

// keep_alpha == false mirrors CV_LOAD_IMAGE_COLOR (3 channels); true mirrors flag -1 (unchanged: 3 or 4 channels).
// Every PNG the format defines is accepted, converted the way cv::imread's 8-bit decode does (grfmt_png.cpp):
// bit depths 1/2/4 are expanded (grey scaled to 0..255, palette looked up), 16-bit samples keep their high byte
// (png_set_strip_16), grey becomes B=G=R, Adam7-interlaced files are de-interlaced.
// `im` is overwritten; its pixel buffer is reused when it is large enough (a stream hands the previous frames' images
// back to its decoders: 17 x 12.6 MB mapped, page-faulted and unmapped per frame cost more than the decoding).
inline int g_read_threads = 0;  // threads of the parallel band reader (0: by the image's size, up to 8; < 0: sequential reader only)
inline void read_into(const std::string& path, bool keep_alpha, Image& im) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("failed to load image: " + path);
  std::vector<uint8_t> file;
  if (std::fseek(f, 0, SEEK_END) == 0) {  // the whole file in one read (state images are up to 85 MB)
    const long sz = std::ftell(f);
    std::rewind(f);
    if (sz > 0) {
      file.resize((size_t)sz);
      file.resize(std::fread(file.data(), 1, (size_t)sz, f));
    }
  }
  uint8_t buf[1 << 16];
  size_t n;
  while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + n);  // (a source that cannot seek, or grew)
  std::fclose(f);
  static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
  if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0) throw std::runtime_error("not a PNG file: " + path);
  size_t pos = 8;
  int w = 0, h = 0, depth = 0, ctype = -1, interlace = 0;
  bool have_ihdr = false;
  std::vector<uint8_t> idat, plte, trns;
  std::vector<std::pair<size_t, size_t>> idat_chunks;  // (offset, length) in `file`
  int band_rows = 0;                                   // "sbNd": this writer's independent bands (write_rows)
  while (pos + 12 <= file.size()) {
    const uint32_t len = be32(&file[pos]);
    const char* type = (const char*)&file[pos + 4];
    const uint8_t* data = &file[pos + 8];
    if (len > file.size() || pos + 12 + len > file.size()) break;
    if (!std::memcmp(type, "IHDR", 4)) {
      if (len < 13) throw std::runtime_error("corrupt PNG header: " + path);
      const uint32_t uw = be32(data), uh = be32(data + 4);
      if (uw == 0 || uh == 0 || uw > 65535u || uh > 65535u) throw std::runtime_error("unsupported PNG dimensions: " + path);
      w = (int)uw; h = (int)uh; depth = data[8]; ctype = data[9]; interlace = data[12];
      if (data[10] != 0 || data[11] != 0 || interlace > 1) throw std::runtime_error("unsupported PNG coding: " + path);
      have_ihdr = true;
    } else if (!std::memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
    else if (!std::memcmp(type, "tRNS", 4)) trns.assign(data, data + len);
    else if (!std::memcmp(type, "sbNd", 4) && len == 4) band_rows = (int)std::min<uint32_t>(be32(data), 65535u);
    else if (!std::memcmp(type, "IDAT", 4)) idat_chunks.emplace_back(pos + 8, (size_t)len);
    else if (!std::memcmp(type, "IEND", 4)) break;
    pos += 12 + len;
  }
  if (!have_ihdr) throw std::runtime_error("corrupt PNG (no header): " + path);
  const int sc = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;  // samples per pixel
  const bool depth_ok = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
                        (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                        ((ctype == 2 || ctype == 4 || ctype == 6) && (depth == 8 || depth == 16));
  if (!sc || !depth_ok) throw std::runtime_error("unsupported PNG colour type / bit depth: " + path);
  const int bpp_bits = sc * depth, fbpp = std::max(1, bpp_bits / 8);  // filter distance in bytes
  auto row_bytes = [&](int pw) { return ((size_t)pw * bpp_bits + 7) / 8; };
  // passes: {x0, y0, dx, dy}; non-interlaced = one pass covering everything
  static const int adam7[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
  static const int whole[1][4] = {{0, 0, 1, 1}};
  const int (*passes)[4] = interlace ? adam7 : whole;
  const int npass = interlace ? 7 : 1;
  size_t total = 0;
  for (int p = 0; p < npass; ++p) {
    const int pw = (w - passes[p][0] + passes[p][2] - 1) / passes[p][2], ph = (h - passes[p][1] + passes[p][3] - 1) / passes[p][3];
    if (pw > 0 && ph > 0) total += (row_bytes(pw) + 1) * ph;
  }
  const bool has_alpha = ctype == 4 || ctype == 6 || (ctype == 3 && !trns.empty());
  im.w = w; im.h = h; im.c = (keep_alpha && has_alpha) ? 4 : 3;
  im.px.resize((size_t)w * h * im.c);
  // A file of this writer (write_rows with the Sub filter): the zlib header in an IDAT chunk of its own, then one IDAT chunk per
  // band of `band_rows` scanlines, each an independent sync-flushed raw-deflate segment, every scanline filtered with Sub — said
  // by the private ancillary chunk "sbNd". Such bands are inflated, unfiltered and converted IN PARALLEL (a frame's state images
  // are up to 85 MB of scanlines each, and one zlib stream inflates on one thread). Anything unexpected — another band count, a
  // filter that needs the row above, a segment that does not inflate to its size — falls back to the sequential reader below.
  if (g_read_threads >= 0 && band_rows > 0 && !interlace && depth == 8 && (ctype == 2 || ctype == 6) && idat_chunks.size() >= 3 &&
      idat_chunks[0].second == 2 && idat_chunks.back().second == 4 && (int)idat_chunks.size() - 2 == (h + band_rows - 1) / band_rows) {
    const int nb = (int)idat_chunks.size() - 2;  // (zlib header, the bands, Adler-32)
    const size_t stride = row_bytes(w), line = stride + 1;
    const int want = g_read_threads > 0 ? g_read_threads : (int)std::min<size_t>(8, std::max<size_t>(1, total / ((size_t)8 << 20)));
    const int nthreads = std::max(1, std::min(want, nb));
    std::atomic<int> next(0);
    std::atomic<bool> bad(false);
    // (the bands are raw deflate: the stream's Adler-32 — the trailing 4-byte IDAT — is checked from per-band sums like the
    // sequential reader's zlib does it, so a state image whose bytes are damaged but still inflate is refused here too)
    std::vector<uLong> band_adler((size_t)nb, 1);
    std::vector<size_t> band_len((size_t)nb, 0);
    auto worker = [&] {
      std::vector<uint8_t> buf;
      for (int bi = next.fetch_add(1); bi < nb && !bad.load(); bi = next.fetch_add(1)) {
        const int y0 = bi * band_rows, rows = std::min(band_rows, h - y0);
        buf.resize((size_t)rows * line);
        z_stream z2;
        std::memset(&z2, 0, sizeof z2);
        if (inflateInit2(&z2, -15) != Z_OK) { bad = true; return; }
        z2.next_in = &file[idat_chunks[bi + 1].first];
        z2.avail_in = (uInt)idat_chunks[bi + 1].second;
        z2.next_out = buf.data();
        z2.avail_out = (uInt)buf.size();
        const int rc = inflate(&z2, Z_SYNC_FLUSH);
        const bool ok = (rc == Z_OK || rc == Z_STREAM_END) && z2.avail_out == 0 && (z2.avail_in == 0 || rc == Z_STREAM_END);
        inflateEnd(&z2);
        if (!ok) { bad = true; return; }
        band_adler[bi] = adler32(1L, buf.data(), (uInt)buf.size());
        band_len[bi] = buf.size();
        for (int r = 0; r < rows; ++r) {
          uint8_t* cur = buf.data() + (size_t)r * line + 1;
          if (cur[-1] == 1) { for (size_t i = (size_t)fbpp; i < stride; ++i) cur[i] = (uint8_t)(cur[i] + cur[i - fbpp]); }
          else if (cur[-1] != 0) { bad = true; return; }
          uint8_t* o = &im.px[(size_t)(y0 + r) * w * im.c];
          const uint8_t* sp = cur;
          if (im.c == 3) for (int px = 0; px < w; ++px, sp += sc, o += 3) { o[0] = sp[2]; o[1] = sp[1]; o[2] = sp[0]; }
          else for (int px = 0; px < w; ++px, sp += 4, o += 4) { o[0] = sp[2]; o[1] = sp[1]; o[2] = sp[0]; o[3] = sp[3]; }
        }
      }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; ++t) th.emplace_back(worker);
    worker();
    for (auto& t : th) t.join();
    if (!bad.load()) {
      uLong adler = 1;
      for (int bi = 0; bi < nb; ++bi) adler = adler32_combine(adler, band_adler[bi], (z_off_t)band_len[bi]);
      if ((uint32_t)adler == be32(&file[idat_chunks.back().first])) return;
      // (a mismatch falls through to the sequential reader, whose zlib reports the stream as corrupt)
    }
  }
  for (const auto& ch : idat_chunks) idat.insert(idat.end(), file.begin() + ch.first, file.begin() + ch.first + ch.second);
  if (idat.empty()) throw std::runtime_error("corrupt PNG data: " + path);
  // The zlib stream is inflated a band of scanlines at a time and each band is unfiltered and converted while it is
  // still in cache (a 2048 x 2048 camera image is 12.6 MB of scanlines; 17 of them are decoded at once per frame).
  z_stream zs;
  std::memset(&zs, 0, sizeof zs);
  if (inflateInit(&zs) != Z_OK) throw std::runtime_error("corrupt PNG data: " + path);
  struct ZEnd { z_stream* z; ~ZEnd() { inflateEnd(z); } } zend{&zs};
  zs.next_in = idat.data();
  zs.avail_in = (uInt)std::min<size_t>(idat.size(), 0xffffffffu);
  if (idat.size() > 0xffffffffu) throw std::runtime_error("corrupt PNG data: " + path);
  bool stream_end = false;
  size_t produced = 0;
  auto inflate_into = [&](uint8_t* dst, size_t n) {  // exactly n more bytes of scanline data, or the file is corrupt
    zs.next_out = dst;
    zs.avail_out = (uInt)n;
    while (zs.avail_out != 0) {
      if (stream_end) throw std::runtime_error("corrupt PNG data: " + path);
      const int rc = inflate(&zs, Z_NO_FLUSH);
      if (rc == Z_STREAM_END) stream_end = true;
      else if (rc != Z_OK) throw std::runtime_error("corrupt PNG data: " + path);
    }
    produced += n;
  };
  for (int p = 0; p < npass; ++p) {
    const int x0 = passes[p][0], y0 = passes[p][1], dx = passes[p][2], dy = passes[p][3];
    const int pw = (w - x0 + dx - 1) / dx, ph = (h - y0 + dy - 1) / dy;
    if (pw <= 0 || ph <= 0) continue;
    const size_t stride = row_bytes(pw), line = stride + 1;
    const int band = (int)std::max<size_t>(1, std::min<size_t>((size_t)ph, ((size_t)256 << 10) / line));
    std::vector<uint8_t> buf((size_t)band * line + 8), above(stride + 8, 0);  // (+8: the pixel-wide Paeth loads)
    for (int py0 = 0; py0 < ph; py0 += band) {
      const int rows = std::min(band, ph - py0);
      inflate_into(buf.data(), (size_t)rows * line);
      for (int r = 0; r < rows; ++r) {
        uint8_t* cur = buf.data() + (size_t)r * line + 1;
        const uint8_t* prev = r ? cur - line : above.data();
        const size_t bp = (size_t)fbpp;
        switch (cur[-1]) {  // the scanline's filter type, undone in place
          case 0: break;
          case 1:
            for (size_t i = bp; i < stride; ++i) cur[i] = (uint8_t)(cur[i] + cur[i - bp]);
            break;
          case 2:
            for (size_t i = 0; i < stride; ++i) cur[i] = (uint8_t)(cur[i] + prev[i]);
            break;
          case 3:
            for (size_t i = 0; i < std::min(bp, stride); ++i) cur[i] = (uint8_t)(cur[i] + (prev[i] >> 1));
            for (size_t i = bp; i < stride; ++i) cur[i] = (uint8_t)(cur[i] + ((cur[i - bp] + prev[i]) >> 1));
            break;
          case 4:
            if (bp == 3 && stride % 3 == 0) { unfilter_paeth_px<3>(cur, prev, stride); break; }
            if (bp == 4 && stride % 4 == 0) { unfilter_paeth_px<4>(cur, prev, stride); break; }
            for (size_t i = 0; i < std::min(bp, stride); ++i) cur[i] = (uint8_t)(cur[i] + prev[i]);  // paeth(0, b, 0) = b
            for (size_t i = bp; i < stride; ++i) cur[i] = (uint8_t)(cur[i] + paeth(cur[i - bp], prev[i], prev[i - bp]));
            break;
          default: throw std::runtime_error("corrupt PNG filter: " + path);
        }
        uint8_t* orow = &im.px[(size_t)(y0 + (py0 + r) * dy) * w * im.c];
        if (depth == 8 && (ctype == 2 || ctype == 6) && dx == 1) {  // the camera images: R,G,B(,A) bytes -> B,G,R(,A)
          const uint8_t* sp = cur;
          uint8_t* o = orow;
          if (im.c == 3) for (int px = 0; px < pw; ++px, sp += sc, o += 3) { o[0] = sp[2]; o[1] = sp[1]; o[2] = sp[0]; }
          else for (int px = 0; px < pw; ++px, sp += 4, o += 4) { o[0] = sp[2]; o[1] = sp[1]; o[2] = sp[0]; o[3] = sp[3]; }
          continue;
        }
        auto sample = [&](int px, int k) -> int {  // sample k of pixel px as 8 bits (16-bit: high byte; <8-bit grey: scaled)
          if (depth == 8) return cur[(size_t)px * sc + k];
          if (depth == 16) return cur[((size_t)px * sc + k) * 2];
          const int bit = px * depth, v = (cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1);
          return ctype == 3 ? v : v * 255 / ((1 << depth) - 1);
        };
        for (int px = 0; px < pw; ++px) {
          uint8_t r8, g, b, a = 255;
          if (ctype == 0) { r8 = g = b = (uint8_t)sample(px, 0); }
          else if (ctype == 4) { r8 = g = b = (uint8_t)sample(px, 0); a = (uint8_t)sample(px, 1); }
          else if (ctype == 2) { r8 = (uint8_t)sample(px, 0); g = (uint8_t)sample(px, 1); b = (uint8_t)sample(px, 2); }
          else if (ctype == 6) { r8 = (uint8_t)sample(px, 0); g = (uint8_t)sample(px, 1); b = (uint8_t)sample(px, 2); a = (uint8_t)sample(px, 3); }
          else {
            const size_t k = (size_t)sample(px, 0);
            if (3 * k + 2 >= plte.size()) throw std::runtime_error("corrupt PNG palette: " + path);
            r8 = plte[3 * k]; g = plte[3 * k + 1]; b = plte[3 * k + 2];
            if (k < trns.size()) a = trns[k];
          }
          uint8_t* o = orow + (size_t)(x0 + px * dx) * im.c;
          o[0] = b; o[1] = g; o[2] = r8;
          if (im.c == 4) o[3] = a;
        }
      }
      std::memcpy(above.data(), buf.data() + (size_t)(rows - 1) * line + 1, stride);
    }
  }
  // like uncompress() on a buffer of exactly the expected size: a stream that holds more scanline data, or does not
  // end, is corrupt
  if (produced != total) throw std::runtime_error("corrupt PNG data: " + path);
  if (!stream_end) {
    uint8_t extra;
    zs.next_out = &extra;
    zs.avail_out = 1;
    const int rc = inflate(&zs, Z_NO_FLUSH);
    if (rc != Z_STREAM_END || zs.avail_out != 1) throw std::runtime_error("corrupt PNG data: " + path);
  }
}
inline Image read(const std::string& path, bool keep_alpha) {
  Image im;
  read_into(path, keep_alpha, im);
  return im;
}

// Output file that remembers the first failed write: a full disk must not leave a truncated PNG behind a zero exit
// code (the reference's imwriteExceptionOnFail aborts, CvUtil.h).
struct OutFile {
  FILE* f;
  bool ok = true;
  std::string path;
  explicit OutFile(const std::string& p) : f(std::fopen(p.c_str(), "wb")), path(p) {
    if (!f) throw std::runtime_error("failed to write image: " + p);
  }
  void put(const void* d, size_t n) {
    if (n && std::fwrite(d, 1, n, f) != n) ok = false;
  }
  void close() {
    const bool closed = std::fclose(f) == 0;
    f = nullptr;
    if (!ok || !closed) {
      std::remove(path.c_str());
      throw std::runtime_error("failed to write image (short write): " + path);
    }
  }
  ~OutFile() { if (f) { std::fclose(f); std::remove(path.c_str()); } }  // not closed = not complete: no truncated file stays behind
};

inline void chunk(OutFile& f, const char* type, const uint8_t* data, size_t len) {
  uint8_t hdr[8];
  put32(hdr, (uint32_t)len);
  std::memcpy(hdr + 4, type, 4);
  f.put(hdr, 8);
  f.put(data, len);
  uLong crc = crc32(0L, (const Bytef*)type, 4);
  if (len) crc = crc32(crc, data, (uInt)len);
  uint8_t c[4];
  put32(c, (uint32_t)crc);
  f.put(c, 4);
}

// px: B,G,R(,A) rows. Encode speed matters more than size here: the 8192x8192 equirect is 201 MB of scanlines, and a stream
// has one to encode every ~110 ms beside 17 camera images to decode. The scanlines are deflated in parallel (pigz-style):
// bands of rows become independent raw-deflate streams that end with a sync flush (byte-aligned, no final block) —
// concatenated they are ONE valid deflate stream; the zlib header and the Adler-32 of the whole image (adler32_combine of
// the bands) are added around them. Filter and deflate settings are those of cv::imwrite's PngEncoder at its defaults
// (Sub on every row, Z_BEST_SPEED, Z_RLE): measured on an 8K frame of the synthetic world's kind of content with 4 threads,
// Sub + Z_RLE 0.60 s / 90 MB against 1.6 s / 107 MB for Up + the default strategy (tools/host_io_time) — the end-to-end
// stream was bound by exactly this CPU time (profiles/r04_v2_end_to_end_*).
// fill_row(y, dst) writes the (w * c * depth / 8) bytes of scanline y in PNG order (R,G,B(,A); 16-bit samples big-endian).
// CPUs this process may use: the hardware threads, cut down to the control group's quota (cgroup v2 cpu.max, v1
// cpu.cfs_quota_us) — a container on a 256-thread host that is given 16 CPUs runs 100 deflate threads SLOWER than 8
// (measured: an 8K stream through files at 121 ms per frame with 32 threads per encoder, 108 with 8).
inline int available_cpus() {
  int n = (int)std::thread::hardware_concurrency();
  if (n < 1) n = 1;
  auto cut = [&](double cpus) { if (cpus >= 1.0 && cpus < n) n = (int)cpus; };
  if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[64] = {0};
    long long period = 0;
    if (std::fscanf(f, "%63s %lld", q, &period) == 2 && period > 0 && std::strcmp(q, "max") != 0) cut((double)std::atoll(q) / (double)period);
    std::fclose(f);
  } else {
    long long quota = -1, period = 0;
    if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (std::fscanf(g, "%lld", &quota) != 1) quota = -1; std::fclose(g); }
    if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(g, "%lld", &period) != 1) period = 0; std::fclose(g); }
    if (quota > 0 && period > 0) cut((double)quota / (double)period);
  }
  return n;
}
// (experiment switches of tools/host_io_time; the writer's defaults are what the measurements chose)
inline int g_write_strategy = Z_RLE;  // zlib strategy
inline int g_write_filter = 1;        // 1 Sub on every row (what cv::imwrite's PngEncoder sets), 2 Up (Sub on the first row)
template <typename FillRow>
inline void write_rows(const std::string& path, FillRow fill_row, int w, int h, int c, int depth, int level, int max_threads) {
  if (c != 3 && c != 4) throw std::runtime_error("png write: 3 or 4 channels only");
  const size_t stride = (size_t)w * c * (depth / 8) + 1;
  // bands of ~2 MB of scanlines
  const int rows_per_band = (int)std::max<size_t>(1, std::min<size_t>((size_t)h, ((size_t)2 << 20) / stride + 1));
  const int nbands = (h + rows_per_band - 1) / rows_per_band;
  struct Band { std::vector<uint8_t> z; uLong adler = 1, crc = 0; size_t raw = 0; bool ok = true; };
  std::vector<Band> bands(nbands);
  auto compress_band = [&](int bi) {
    Band& B = bands[bi];
    const int y0 = bi * rows_per_band, y1 = std::min(h, y0 + rows_per_band);
    // Scanlines are filtered before they are deflated (Sub: each byte minus the one a pixel to its left; g_write_filter 2:
    // Up, whose first row of a band needs the row above the band, which fill_row gives).
    std::vector<uint8_t> raw((size_t)(y1 - y0) * stride);
    const size_t nb = stride - 1, px_bytes = (size_t)c * (depth / 8);
    std::vector<uint8_t> rowA(nb), rowB(nb);
    uint8_t *cur = rowA.data(), *prev = rowB.data();
    const bool sub = g_write_filter == 1;
    if (y0 > 0 && !sub) fill_row(y0 - 1, prev);
    uint8_t* o = raw.data();
    for (int y = y0; y < y1; ++y) {
      fill_row(y, cur);
      if (y == 0 || sub) {
        *o++ = 1;  // Sub
        for (size_t i = 0; i < std::min(px_bytes, nb); ++i) o[i] = cur[i];
        for (size_t i = px_bytes; i < nb; ++i) o[i] = (uint8_t)(cur[i] - cur[i - px_bytes]);
      } else {
        *o++ = 2;  // Up
        for (size_t i = 0; i < nb; ++i) o[i] = (uint8_t)(cur[i] - prev[i]);
      }
      o += nb;
      std::swap(cur, prev);
    }
    B.raw = raw.size();
    B.adler = adler32(1L, raw.data(), (uInt)raw.size());
    z_stream zs;
    std::memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, g_write_strategy) != Z_OK) { B.ok = false; return; }
    B.z.resize(deflateBound(&zs, (uLong)raw.size()) + 64);
    zs.next_in = raw.data();
    zs.avail_in = (uInt)raw.size();
    zs.next_out = B.z.data();
    zs.avail_out = (uInt)B.z.size();
    const bool last = bi == nbands - 1;
    const int rc = deflate(&zs, last ? Z_FINISH : Z_SYNC_FLUSH);
    B.ok = last ? rc == Z_STREAM_END : (rc == Z_OK && zs.avail_in == 0 && zs.avail_out != 0);
    B.z.resize(B.z.size() - zs.avail_out);
    deflateEnd(&zs);
    B.crc = crc32(crc32(0L, (const Bytef*)"IDAT", 4), B.z.data(), (uInt)B.z.size());  // the chunk's CRC, also in parallel
  };
  int nthreads = max_threads > 0 ? max_threads : available_cpus();
  nthreads = std::max(1, std::min(nthreads, nbands));
  // The file is written band by band WHILE later bands still compress: this thread writes band i as soon as it is there
  // (bands are handed out in order, so they finish nearly in order) instead of after the last one — on a file system that
  // takes 0.2 s for an 8K equirect's 40-80 MB the write used to start when the deflate threads had all finished.
  std::mutex mu;
  std::condition_variable cv;
  std::vector<char> ready(nbands, 0);
  std::atomic<int> next(0);
  std::vector<std::thread> th;
  auto worker = [&] {
    for (int bi = next.fetch_add(1); bi < nbands; bi = next.fetch_add(1)) {
      compress_band(bi);
      { std::lock_guard<std::mutex> lk(mu); ready[bi] = 1; }
      cv.notify_all();
    }
  };
  if (nthreads > 1)
    for (int t = 0; t < nthreads; ++t) th.emplace_back(worker);
  struct Joiner { std::vector<std::thread>& t; ~Joiner() { for (auto& x : t) if (x.joinable()) x.join(); } } joiner{th};
  OutFile f(path);
  static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
  f.put(sig, 8);
  uint8_t ihdr[13];
  put32(ihdr, (uint32_t)w); put32(ihdr + 4, (uint32_t)h);
  ihdr[8] = (uint8_t)depth; ihdr[9] = c == 3 ? 2 : 6; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
  chunk(f, "IHDR", ihdr, 13);
  if (g_write_filter == 1 && depth == 8) {  // what lets read_into take the bands in parallel (ancillary, private, safe to copy)
    uint8_t br[4];
    put32(br, (uint32_t)rows_per_band);
    chunk(f, "sbNd", br, 4);
  }
  // zlib stream = header, the bands' deflate data, Adler-32; one IDAT chunk per piece
  static const uint8_t zhdr[2] = {0x78, 0x01};
  chunk(f, "IDAT", zhdr, 2);
  uLong adler = 1;
  bool failed = false;
  for (int bi = 0; bi < nbands; ++bi) {
    if (nthreads == 1) compress_band(bi);
    else {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return ready[bi] != 0; });
    }
    Band& B = bands[bi];
    if (!B.ok) { failed = true; continue; }  // (keep draining: the workers must finish before `bands` goes away)
    if (failed) continue;
    uint8_t hdr[8], crc[4];
    put32(hdr, (uint32_t)B.z.size());
    std::memcpy(hdr + 4, "IDAT", 4);
    put32(crc, (uint32_t)B.crc);
    f.put(hdr, 8);
    f.put(B.z.data(), B.z.size());
    f.put(crc, 4);
    adler = adler32_combine(adler, B.adler, (z_off_t)B.raw);
    std::vector<uint8_t>().swap(B.z);  // written: give the band's memory back while the others compress
  }
  if (failed) throw std::runtime_error("png write: deflate failed");
  uint8_t tail[4];
  put32(tail, (uint32_t)adler);
  chunk(f, "IDAT", tail, 4);
  chunk(f, "IEND", nullptr, 0);
  f.close();
}
// 8-bit B,G,R(,A) -> RGB / RGBA PNG
inline void write(const std::string& path, const uint8_t* px, int w, int h, int c, int level = 1, int max_threads = 0) {
  write_rows(path, [=](int y, uint8_t* o) {
    const uint8_t* s = px + (size_t)y * w * c;
    for (int x = 0; x < w; ++x) {
      o[0] = s[2]; o[1] = s[1]; o[2] = s[0];
      if (c == 4) o[3] = s[3];
      o += c; s += c;
    }
  }, w, h, c, 8, level, max_threads);
}
// 16-bit B,G,R (CV_16UC3) -> 16-bit RGB PNG (big-endian samples), what imwrite does for the ISP's 16-bit output
inline void write16(const std::string& path, const uint16_t* px, int w, int h, int level = 1, int max_threads = 0) {
  write_rows(path, [=](int y, uint8_t* o) {
    const uint16_t* s = px + (size_t)y * w * 3;
    for (int x = 0; x < w; ++x, s += 3, o += 6) {
      o[0] = s[2] >> 8; o[1] = (uint8_t)s[2]; o[2] = s[1] >> 8; o[3] = (uint8_t)s[1]; o[4] = s[0] >> 8; o[5] = (uint8_t)s[0];
    }
  }, w, h, 3, 16, level, max_threads);
}

// Greyscale PNG, 8 or 16 bits, unchanged depth (imread(path, GRAYSCALE | ANYDEPTH) on a raw Bayer image, Raw2Rgb.cpp:402-404):
// `depth` receives 8 or 16; samples are returned as uint16 (8-bit ones unscaled). Other colour types are rejected: a
// Bayer mosaic is not a colour image.
inline std::vector<uint16_t> read_gray(const std::string& path, int* w_out, int* h_out, int* depth_out) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("failed to load image: " + path);
  std::vector<uint8_t> file;
  if (std::fseek(f, 0, SEEK_END) == 0) {  // the whole file in one read (state images are up to 85 MB)
    const long sz = std::ftell(f);
    std::rewind(f);
    if (sz > 0) {
      file.resize((size_t)sz);
      file.resize(std::fread(file.data(), 1, (size_t)sz, f));
    }
  }
  uint8_t buf[1 << 16];
  size_t n;
  while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + n);  // (a source that cannot seek, or grew)
  std::fclose(f);
  static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
  if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0) throw std::runtime_error("not a PNG file: " + path);
  size_t pos = 8;
  int w = 0, h = 0, depth = 0, ctype = -1;
  std::vector<uint8_t> idat;
  while (pos + 12 <= file.size()) {
    const uint32_t len = be32(&file[pos]);
    const char* type = (const char*)&file[pos + 4];
    const uint8_t* data = &file[pos + 8];
    if (len > file.size() || pos + 12 + len > file.size()) break;
    if (!std::memcmp(type, "IHDR", 4)) {
      if (len < 13) throw std::runtime_error("corrupt PNG header: " + path);
      const uint32_t uw = be32(data), uh = be32(data + 4);
      if (uw == 0 || uh == 0 || uw > 65535u || uh > 65535u) throw std::runtime_error("unsupported PNG dimensions: " + path);
      w = (int)uw; h = (int)uh; depth = data[8]; ctype = data[9];
      if (data[10] != 0 || data[11] != 0 || data[12] != 0) throw std::runtime_error("unsupported PNG coding (interlaced raw image?): " + path);
    } else if (!std::memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), data, data + len);
    } else if (!std::memcmp(type, "IEND", 4)) {
      break;
    }
    pos += 12 + (size_t)len;
  }
  if (ctype != 0 || (depth != 8 && depth != 16)) throw std::runtime_error("raw input must be an 8- or 16-bit greyscale PNG: " + path);
  const int bps = depth / 8;
  const size_t stride = (size_t)w * bps;
  std::vector<uint8_t> raw((stride + 1) * h);
  uLongf rawlen = (uLongf)raw.size();
  if (idat.empty() || uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size())
    throw std::runtime_error("corrupt PNG data: " + path);
  std::vector<uint16_t> out((size_t)w * h);
  std::vector<uint8_t> prev(stride, 0), cur(stride);
  const uint8_t* in = raw.data();
  for (int y = 0; y < h; ++y) {
    const int ft = *in++;
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= (size_t)bps ? cur[i - bps] : 0, b = prev[i], c = i >= (size_t)bps ? prev[i - bps] : 0;
      int v = in[i];
      switch (ft) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) >> 1; break;
        case 4: v += paeth(a, b, c); break;
        default: throw std::runtime_error("corrupt PNG filter: " + path);
      }
      cur[i] = (uint8_t)v;
    }
    in += stride;
    uint16_t* o = &out[(size_t)y * w];
    for (int x = 0; x < w; ++x) o[x] = bps == 1 ? cur[x] : (uint16_t)(cur[2 * x] << 8 | cur[2 * x + 1]);
    prev.swap(cur);
  }
  *w_out = w; *h_out = h; *depth_out = depth;
  return out;
}

}  // namespace pngio
