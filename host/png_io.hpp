// png_io.hpp — minimal PNG codec on zlib for the host binary (the reference uses cv::imread / cv::imwrite,
// RigDescription.cpp:87-105, TRSP:961). Reads 8-bit grey / grey+alpha / RGB / RGBA / palette, non-interlaced;
// writes 8-bit RGB / RGBA. Pixel order at this interface is OpenCV's: B,G,R(,A).
#pragma once
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <thread>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace pngio {

struct Image {
  int w = 0, h = 0, c = 0;  // c = 3 (BGR) or 4 (BGRA)
  std::vector<uint8_t> px;
};

inline uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }
inline void put32(uint8_t* p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; }
inline int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// keep_alpha == false mirrors CV_LOAD_IMAGE_COLOR (3 channels); true mirrors flag -1 (unchanged: 3 or 4 channels).
inline Image read(const std::string& path, bool keep_alpha) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("failed to load image: " + path);
  std::vector<uint8_t> file;
  uint8_t buf[1 << 16];
  size_t n;
  while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + n);
  std::fclose(f);
  static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
  if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0) throw std::runtime_error("not a PNG file: " + path);
  size_t pos = 8;
  int w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
  std::vector<uint8_t> idat, plte, trns;
  while (pos + 12 <= file.size()) {
    const uint32_t len = be32(&file[pos]);
    const char* type = (const char*)&file[pos + 4];
    const uint8_t* data = &file[pos + 8];
    if (pos + 12 + len > file.size()) break;
    if (!std::memcmp(type, "IHDR", 4)) {
      w = (int)be32(data); h = (int)be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
    } else if (!std::memcmp(type, "PLTE", 4)) plte.assign(data, data + len);
    else if (!std::memcmp(type, "tRNS", 4)) trns.assign(data, data + len);
    else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
    else if (!std::memcmp(type, "IEND", 4)) break;
    pos += 12 + len;
  }
  if (w <= 0 || h <= 0 || depth != 8 || interlace != 0)
    throw std::runtime_error("unsupported PNG (need 8-bit, non-interlaced): " + path);
  const int sc = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
  if (!sc) throw std::runtime_error("unsupported PNG colour type: " + path);
  const size_t stride = (size_t)w * sc;
  std::vector<uint8_t> raw((stride + 1) * h);
  uLongf rawlen = raw.size();
  if (uncompress(raw.data(), &rawlen, idat.data(), idat.size()) != Z_OK || rawlen != raw.size())
    throw std::runtime_error("corrupt PNG data: " + path);
  std::vector<uint8_t> prev(stride, 0), cur(stride);
  const bool has_alpha = ctype == 4 || ctype == 6 || (ctype == 3 && !trns.empty());
  Image im;
  im.w = w; im.h = h; im.c = (keep_alpha && has_alpha) ? 4 : 3;
  im.px.resize((size_t)w * h * im.c);
  for (int y = 0; y < h; ++y) {
    const uint8_t* in = &raw[(stride + 1) * y];
    const int ft = in[0];
    ++in;
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= (size_t)sc ? cur[i - sc] : 0, b = prev[i], c = i >= (size_t)sc ? prev[i - sc] : 0;
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
    uint8_t* o = &im.px[(size_t)y * w * im.c];
    for (int x = 0; x < w; ++x) {
      uint8_t r, g, b, a = 255;
      const uint8_t* s = &cur[(size_t)x * sc];
      if (ctype == 0) { r = g = b = s[0]; }
      else if (ctype == 4) { r = g = b = s[0]; a = s[1]; }
      else if (ctype == 2) { r = s[0]; g = s[1]; b = s[2]; }
      else if (ctype == 6) { r = s[0]; g = s[1]; b = s[2]; a = s[3]; }
      else {
        const size_t k = s[0];
        if (3 * k + 2 >= plte.size()) throw std::runtime_error("corrupt PNG palette: " + path);
        r = plte[3 * k]; g = plte[3 * k + 1]; b = plte[3 * k + 2];
        if (k < trns.size()) a = trns[k];
      }
      o[0] = b; o[1] = g; o[2] = r;
      if (im.c == 4) o[3] = a;
      o += im.c;
    }
    prev.swap(cur);
  }
  return im;
}

inline void chunk(FILE* f, const char* type, const uint8_t* data, size_t len) {
  uint8_t hdr[8];
  put32(hdr, (uint32_t)len);
  std::memcpy(hdr + 4, type, 4);
  std::fwrite(hdr, 1, 8, f);
  if (len) std::fwrite(data, 1, len, f);
  uLong crc = crc32(0L, (const Bytef*)type, 4);
  if (len) crc = crc32(crc, data, (uInt)len);
  uint8_t c[4];
  put32(c, (uint32_t)crc);
  std::fwrite(c, 1, 4, f);
}

// px: B,G,R(,A) rows. Filter "up"/"sub" is skipped (type 0) — encode speed matters more than size here: the 8192x8192
// equirect is 201 MB of scanlines and a single zlib stream at level 1 takes ~6 s on one core, 35x the GPU time of the
// frame. The scanlines are therefore deflated in parallel (pigz-style): bands of rows become independent raw-deflate
// streams that end with a sync flush (byte-aligned, no final block) — concatenated they are ONE valid deflate stream;
// the zlib header and the Adler-32 of the whole image (adler32_combine of the bands) are added around them.
inline void write(const std::string& path, const uint8_t* px, int w, int h, int c, int level = 1, int max_threads = 0) {
  if (c != 3 && c != 4) throw std::runtime_error("png write: 3 or 4 channels only");
  const size_t stride = (size_t)w * c + 1;
  // bands of ~2 MB of scanlines
  const int rows_per_band = (int)std::max<size_t>(1, std::min<size_t>((size_t)h, ((size_t)2 << 20) / stride + 1));
  const int nbands = (h + rows_per_band - 1) / rows_per_band;
  struct Band { std::vector<uint8_t> z; uLong adler = 1, crc = 0; size_t raw = 0; bool ok = true; };
  std::vector<Band> bands(nbands);
  auto compress_band = [&](int bi) {
    Band& B = bands[bi];
    const int y0 = bi * rows_per_band, y1 = std::min(h, y0 + rows_per_band);
    std::vector<uint8_t> raw((size_t)(y1 - y0) * stride);
    uint8_t* o = raw.data();
    for (int y = y0; y < y1; ++y) {
      *o++ = 0;  // filter type None
      const uint8_t* s = px + (size_t)y * w * c;
      for (int x = 0; x < w; ++x) {
        o[0] = s[2]; o[1] = s[1]; o[2] = s[0];
        if (c == 4) o[3] = s[3];
        o += c; s += c;
      }
    }
    B.raw = raw.size();
    B.adler = adler32(1L, raw.data(), (uInt)raw.size());
    z_stream zs;
    std::memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) { B.ok = false; return; }
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
  int nthreads = max_threads > 0 ? max_threads : (int)std::thread::hardware_concurrency();
  nthreads = std::max(1, std::min(nthreads, nbands));
  if (nthreads == 1) {
    for (int bi = 0; bi < nbands; ++bi) compress_band(bi);
  } else {
    std::atomic<int> next(0);
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t)
      th.emplace_back([&] {
        for (int bi = next.fetch_add(1); bi < nbands; bi = next.fetch_add(1)) compress_band(bi);
      });
    for (auto& t : th) t.join();
  }
  for (const Band& B : bands)
    if (!B.ok) throw std::runtime_error("png write: deflate failed");
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) throw std::runtime_error("failed to write image: " + path);
  static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
  std::fwrite(sig, 1, 8, f);
  uint8_t ihdr[13];
  put32(ihdr, (uint32_t)w); put32(ihdr + 4, (uint32_t)h);
  ihdr[8] = 8; ihdr[9] = c == 3 ? 2 : 6; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
  chunk(f, "IHDR", ihdr, 13);
  // zlib stream = header, the bands' deflate data, Adler-32; one IDAT chunk per piece
  static const uint8_t zhdr[2] = {0x78, 0x01};
  chunk(f, "IDAT", zhdr, 2);
  uLong adler = 1;
  for (const Band& B : bands) {
    uint8_t hdr[8], crc[4];
    put32(hdr, (uint32_t)B.z.size());
    std::memcpy(hdr + 4, "IDAT", 4);
    put32(crc, (uint32_t)B.crc);
    std::fwrite(hdr, 1, 8, f);
    std::fwrite(B.z.data(), 1, B.z.size(), f);
    std::fwrite(crc, 1, 4, f);
    adler = adler32_combine(adler, B.adler, (z_off_t)B.raw);
  }
  uint8_t tail[4];
  put32(tail, (uint32_t)adler);
  chunk(f, "IDAT", tail, 4);
  chunk(f, "IEND", nullptr, 0);
  std::fclose(f);
}

}  // namespace pngio
