// png_io.hpp — minimal PNG codec on zlib for the host binary (the reference uses cv::imread / cv::imwrite,
// RigDescription.cpp:87-105, TRSP:961). Reads 8-bit grey / grey+alpha / RGB / RGBA / palette, non-interlaced;
// writes 8-bit RGB / RGBA. Pixel order at this interface is OpenCV's: B,G,R(,A).
#pragma once
#include <zlib.h>

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

// px: B,G,R(,A) rows. Filter "up"/"sub" is skipped (type 0) — encode speed matters more than size here.
inline void write(const std::string& path, const uint8_t* px, int w, int h, int c, int level = 1) {
  if (c != 3 && c != 4) throw std::runtime_error("png write: 3 or 4 channels only");
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) throw std::runtime_error("failed to write image: " + path);
  static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
  std::fwrite(sig, 1, 8, f);
  uint8_t ihdr[13];
  put32(ihdr, (uint32_t)w); put32(ihdr + 4, (uint32_t)h);
  ihdr[8] = 8; ihdr[9] = c == 3 ? 2 : 6; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
  chunk(f, "IHDR", ihdr, 13);
  z_stream zs;
  std::memset(&zs, 0, sizeof zs);
  if (deflateInit(&zs, level) != Z_OK) { std::fclose(f); throw std::runtime_error("deflateInit failed"); }
  std::vector<uint8_t> row((size_t)w * c + 1), out(1 << 20);
  auto drain = [&](int flush) {
    int rc;
    do {
      zs.next_out = out.data();
      zs.avail_out = (uInt)out.size();
      rc = deflate(&zs, flush);
      const size_t have = out.size() - zs.avail_out;
      if (have) chunk(f, "IDAT", out.data(), have);
    } while (zs.avail_out == 0 || (flush == Z_FINISH && rc != Z_STREAM_END));
  };
  for (int y = 0; y < h; ++y) {
    row[0] = 0;
    const uint8_t* s = px + (size_t)y * w * c;
    uint8_t* o = row.data() + 1;
    for (int x = 0; x < w; ++x) {
      o[0] = s[2]; o[1] = s[1]; o[2] = s[0];
      if (c == 4) o[3] = s[3];
      o += c; s += c;
    }
    zs.next_in = row.data();
    zs.avail_in = (uInt)row.size();
    drain(Z_NO_FLUSH);
  }
  drain(Z_FINISH);
  deflateEnd(&zs);
  chunk(f, "IEND", nullptr, 0);
  std::fclose(f);
}

}  // namespace pngio
