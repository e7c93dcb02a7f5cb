// jpeg_io.hpp — baseline JPEG reader for the host programs: what cv::imread(path, IMREAD_COLOR) makes of a camera image
// when the rig's images are .jpg (the reference sniffs the extension of the first file of a camera directory and lets
// imread decode it, SystemUtil.h:96-105, RigDescription.cpp:87-105). OpenCV decodes JPEG with libjpeg at its defaults; this
// is that pipeline restated so that the pixels are the same:
//   * sequential Huffman JPEG (SOF0 / SOF1, 8-bit), 1 or 3 components, restart intervals;
//   * the "slow" integer inverse DCT (jidctint.c: 13-bit constants, 2 extra bits after the column pass);
//   * "fancy" chroma upsampling for 2x1 and 2x2 subsampling (jdsample.c: triangle filter, the edge rows / columns
//     repeated), plain replication is not used by the defaults;
//   * YCbCr -> RGB with libjpeg's 16-bit fixed-point tables (jdcolor.c); greyscale is replicated into B, G, R.
// tests/test_cpu_host.py compares it bit for bit with PIL (libjpeg-turbo, whose SIMD paths are bit-exact with libjpeg's C
// code) over subsamplings, qualities, odd sizes and restart markers. Progressive, arithmetic-coded, 12-bit and CMYK files and
// files with an EXIF orientation other than "top-left" (OpenCV >= 3.1 rotates those) are rejected with a message, and so are
// the 4:4:0 (h1v2) and 4:1:1 samplings (cv::imread decodes them; cameras do not write them).
// Assumption, stated: the OpenCV being replaced links libjpeg-turbo or libjpeg 6b-8 (merged h2v1 / h2v2 fancy upsampling as
// above). An OpenCV built with its bundled IJG libjpeg 9 upsamples chroma by DCT scaling and yields different pixels for
// subsampled files; nothing here reproduces that.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "png_io.hpp"

namespace jpegio {

struct Huff {
  uint8_t bits[17] = {0};
  uint8_t vals[256] = {0};
  int mincode[17], maxcode[18], valptr[17];
  bool defined = false;
  void build() {
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
      valptr[l] = k;
      mincode[l] = code;
      code += bits[l];
      k += bits[l];
      maxcode[l] = bits[l] ? code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
    defined = true;
  }
};

struct Component {
  int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
  int wblocks = 0, hblocks = 0;     // allocated size in blocks (whole MCUs)
  int dw = 0, dh = 0;               // downsampled size in samples (what the upsampler sees)
  std::vector<uint8_t> plane;       // wblocks*8 x hblocks*8 samples
  int pred = 0;
};

class Reader {
 public:
  explicit Reader(const std::vector<uint8_t>& d, const std::string& p) : d_(d), path_(p) {}
  pngio::Image decode() {
    if (d_.size() < 4 || d_[0] != 0xFF || d_[1] != 0xD8) fail("not a JPEG file");
    pos_ = 2;
    bool sof = false;
    for (;;) {
      const int m = marker();
      if (m == 0xD9) fail("no image data");
      if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
      const size_t len = be16(pos_);
      if (len < 2 || pos_ + len > d_.size()) fail("truncated segment");
      const size_t seg = pos_ + 2, end = pos_ + len;
      if (m == 0xC0 || m == 0xC1) {
        if (sof) fail("second frame header in one JPEG");  // (hierarchical files only; the geometry below is per frame)
        read_sof(seg, end);
        sof = true;
      }
      else if (m == 0xC2) fail("progressive JPEG is not supported");
      else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) fail("this JPEG process (lossless / arithmetic / hierarchical) is not supported");
      else if (m == 0xCC) fail("arithmetic-coded JPEG is not supported");
      else if (m == 0xC4) read_dht(seg, end);
      else if (m == 0xDB) read_dqt(seg, end);
      else if (m == 0xDD) restart_ = be16(seg);
      else if (m == 0xE1) check_exif(seg, end);
      else if (m == 0xEE) read_adobe(seg, end);
      else if (m == 0xDA) {
        if (!sof) fail("scan before frame header");
        read_sos(seg, end);
        pos_ = end;
        decode_scan();
        break;
      }
      pos_ = end;
    }
    return finish();
  }

 private:
  const std::vector<uint8_t>& d_;
  std::string path_;
  size_t pos_ = 0;
  int w_ = 0, h_ = 0, ncomp_ = 0, hmax_ = 1, vmax_ = 1, restart_ = 0, adobe_transform_ = -1;
  Component comp_[3];
  uint16_t q_[4][64];
  bool qdef_[4] = {false, false, false, false};
  Huff dc_[4], ac_[4];
  // bit reader
  uint32_t bitbuf_ = 0;
  int bitcnt_ = 0;
  bool hit_marker_ = false;

  [[noreturn]] void fail(const std::string& m) const { throw std::runtime_error("failed to load image: " + path_ + " (" + m + ")"); }
  size_t be16(size_t p) const {
    if (p + 1 >= d_.size()) fail("truncated file");
    return (size_t)d_[p] << 8 | d_[p + 1];
  }
  int marker() {
    while (pos_ < d_.size() && d_[pos_] != 0xFF) ++pos_;
    while (pos_ < d_.size() && d_[pos_] == 0xFF) ++pos_;
    if (pos_ >= d_.size()) fail("truncated file");
    return d_[pos_++];
  }
  void read_sof(size_t p, size_t end) {
    if (end - p < 6) fail("bad frame header");
    hmax_ = vmax_ = 1;
    if (d_[p] != 8) fail("only 8-bit JPEG is supported");
    h_ = (int)be16(p + 1);
    w_ = (int)be16(p + 3);
    ncomp_ = d_[p + 5];
    if (w_ <= 0 || h_ <= 0 || (long long)w_ * h_ > (1ll << 28)) fail("bad dimensions");
    if (ncomp_ != 1 && ncomp_ != 3) fail("only greyscale and YCbCr JPEG are supported");
    if (end - p < 6 + (size_t)ncomp_ * 3) fail("bad frame header");
    for (int i = 0; i < ncomp_; ++i) {
      Component& c = comp_[i];
      c.id = d_[p + 6 + i * 3];
      c.h = d_[p + 7 + i * 3] >> 4;
      c.v = d_[p + 7 + i * 3] & 15;
      c.tq = d_[p + 8 + i * 3];
      if (c.h < 1 || c.h > 2 || c.v < 1 || c.v > 2 || c.tq > 3) fail("unsupported sampling factors");
      hmax_ = std::max(hmax_, c.h);
      vmax_ = std::max(vmax_, c.v);
    }
    if (ncomp_ == 1) { comp_[0].h = comp_[0].v = 1; hmax_ = vmax_ = 1; }
    for (int i = 1; i < ncomp_; ++i)
      if (comp_[i].h != 1 || comp_[i].v != 1 || (comp_[0].v == 2 && comp_[0].h == 1)) fail("unsupported chroma subsampling");
  }
  void read_dqt(size_t p, size_t end) {
    static const int zz[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                               35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    while (p < end) {
      const int pq = d_[p] >> 4, tq = d_[p] & 15;
      ++p;
      if (tq > 3 || pq > 1 || p + (size_t)64 * (pq + 1) > end) fail("bad quantisation table");
      for (int i = 0; i < 64; ++i) {
        q_[tq][zz[i]] = pq ? (uint16_t)be16(p) : d_[p];
        p += pq + 1;
      }
      qdef_[tq] = true;
    }
  }
  void read_dht(size_t p, size_t end) {
    while (p < end) {
      const int tc = d_[p] >> 4, th = d_[p] & 15;
      ++p;
      if (tc > 1 || th > 3 || p + 16 > end) fail("bad Huffman table");
      Huff& t = tc ? ac_[th] : dc_[th];
      int n = 0;
      for (int l = 1; l <= 16; ++l) { t.bits[l] = d_[p + l - 1]; n += t.bits[l]; }
      p += 16;
      if (n > 256 || p + (size_t)n > end) fail("bad Huffman table");
      std::memcpy(t.vals, &d_[p], (size_t)n);
      p += (size_t)n;
      t.build();
    }
  }
  void read_adobe(size_t p, size_t end) {
    if (end - p >= 12 && std::memcmp(&d_[p], "Adobe", 5) == 0) adobe_transform_ = d_[p + 11];
  }
  void check_exif(size_t p, size_t end) {  // orientation tag 0x0112 of IFD0; anything but 1 is rejected
    if (end - p < 14 || std::memcmp(&d_[p], "Exif\0\0", 6) != 0) return;
    const size_t t = p + 6;
    const bool le = d_[t] == 'I';
    auto u16 = [&](size_t o) -> unsigned { return o + 1 < end ? (le ? d_[o] | d_[o + 1] << 8 : d_[o] << 8 | d_[o + 1]) : 0u; };
    auto u32 = [&](size_t o) -> unsigned { return o + 3 < end ? (le ? u16(o) | u16(o + 2) << 16 : u16(o) << 16 | u16(o + 2)) : 0u; };
    size_t ifd = t + u32(t + 4);
    const unsigned n = u16(ifd);
    for (unsigned i = 0; i < n; ++i) {
      const size_t e = ifd + 2 + (size_t)i * 12;
      if (e + 12 > end) return;
      if (u16(e) == 0x0112) {
        const unsigned o = u16(e + 8);
        if (o > 1) fail("EXIF orientation " + std::to_string(o) + " is not supported (OpenCV rotates such images)");
        return;
      }
    }
  }
  void read_sos(size_t p, size_t end) {
    if (end <= p) fail("empty scan header");
    const int n = d_[p];
    if (n != ncomp_ || end - p < 1 + (size_t)n * 2 + 3) fail("only single-scan (non-interleaved-free) baseline JPEG is supported");
    for (int i = 0; i < n; ++i) {
      const int id = d_[p + 1 + i * 2], t = d_[p + 2 + i * 2];
      if (id != comp_[i].id) fail("unexpected component order in the scan");
      comp_[i].td = t >> 4;
      comp_[i].ta = t & 15;
      if (comp_[i].td > 3 || comp_[i].ta > 3 || !dc_[comp_[i].td].defined || !ac_[comp_[i].ta].defined || !qdef_[comp_[i].tq])
        fail("scan uses an undefined table");
    }
  }

  // ---- entropy-coded segment ----
  void fill() {
    while (bitcnt_ <= 24) {
      int b = 0;
      if (!hit_marker_ && pos_ < d_.size()) {
        b = d_[pos_];
        if (b == 0xFF) {
          const int b2 = pos_ + 1 < d_.size() ? d_[pos_ + 1] : 0xD9;
          if (b2 == 0) pos_ += 2;
          else { hit_marker_ = true; b = 0; }
        } else ++pos_;
      }
      bitbuf_ |= (uint32_t)b << (24 - bitcnt_);
      bitcnt_ += 8;
    }
  }
  int getbits(int n) {
    if (n == 0) return 0;
    if (bitcnt_ < n) fill();
    const int v = (int)(bitbuf_ >> (32 - n));
    bitbuf_ <<= n;
    bitcnt_ -= n;
    return v;
  }
  int decode_sym(const Huff& t) {
    int code = 0;
    for (int l = 1; l <= 16; ++l) {
      code = code << 1 | getbits(1);
      if (t.maxcode[l] >= 0 && code <= t.maxcode[l] && code >= t.mincode[l]) return t.vals[t.valptr[l] + code - t.mincode[l]];
    }
    fail("bad Huffman code");
  }
  static int extend(int v, int s) { return s && v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

  void decode_block(Component& c, int bx, int by) {
    static const int zz[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                               35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
    int coef[64] = {0};
    const int s = decode_sym(dc_[c.td]);
    if (s > 11) fail("bad DC coefficient");
    c.pred += extend(getbits(s), s);
    coef[0] = c.pred;
    const Huff& ac = ac_[c.ta];
    for (int k = 1; k < 64;) {
      const int rs = decode_sym(ac), r = rs >> 4, sz = rs & 15;
      if (sz == 0) {
        if (r != 15) break;
        k += 16;
        continue;
      }
      k += r;
      if (k > 63) fail("bad AC coefficient");
      coef[zz[k]] = extend(getbits(sz), sz);
      ++k;
    }
    idct(coef, q_[c.tq], &c.plane[((size_t)by * 8) * (size_t)c.wblocks * 8 + (size_t)bx * 8], c.wblocks * 8);
  }

  // jidctint.c (jpeg_idct_islow): dequantisation, column pass into a workspace, row pass, +128 and clamp
  static void idct(const int* in, const uint16_t* q, uint8_t* out, int stride) {
    const int CB = 13, P1 = 2;
    const int F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137, F1961 = 16069,
              F2053 = 16819, F2562 = 20995, F3072 = 25172;
    auto ds = [](long x, int n) -> long { return (x + (1l << (n - 1))) >> n; };
    long ws[64];
    for (int c = 0; c < 8; ++c) {
      long v[8];
      for (int r = 0; r < 8; ++r) v[r] = (long)in[r * 8 + c] * q[r * 8 + c];
      long z2 = v[2], z3 = v[6];
      long z1 = (z2 + z3) * F0541;
      long tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
      z2 = v[0]; z3 = v[4];
      long tmp0 = (z2 + z3) * (1l << CB), tmp1 = (z2 - z3) * (1l << CB);  // (a left shift of a negative value is undefined)
      const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
      tmp0 = v[7]; tmp1 = v[5]; tmp2 = v[3]; tmp3 = v[1];
      z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
      long z4 = tmp1 + tmp3;
      const long z5 = (z3 + z4) * F1175;
      tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
      z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
      z3 += z5; z4 += z5;
      tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
      ws[0 * 8 + c] = ds(tmp10 + tmp3, CB - P1); ws[7 * 8 + c] = ds(tmp10 - tmp3, CB - P1);
      ws[1 * 8 + c] = ds(tmp11 + tmp2, CB - P1); ws[6 * 8 + c] = ds(tmp11 - tmp2, CB - P1);
      ws[2 * 8 + c] = ds(tmp12 + tmp1, CB - P1); ws[5 * 8 + c] = ds(tmp12 - tmp1, CB - P1);
      ws[3 * 8 + c] = ds(tmp13 + tmp0, CB - P1); ws[4 * 8 + c] = ds(tmp13 - tmp0, CB - P1);
    }
    auto put = [](long x) -> uint8_t {  // range_limit[(x) & RANGE_MASK]: the sample + 128, clamped
      x += 128;
      return (uint8_t)(x < 0 ? 0 : x > 255 ? 255 : x);
    };
    for (int r = 0; r < 8; ++r) {
      const long* w = ws + r * 8;
      long z2 = w[2], z3 = w[6];
      long z1 = (z2 + z3) * F0541;
      long tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
      long tmp0 = (w[0] + w[4]) * (1l << CB), tmp1 = (w[0] - w[4]) * (1l << CB);
      const long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
      tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
      z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
      long z4 = tmp1 + tmp3;
      const long z5 = (z3 + z4) * F1175;
      tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
      z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
      z3 += z5; z4 += z5;
      tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
      uint8_t* o = out + (size_t)r * stride;
      const int S = CB + P1 + 3;
      o[0] = put(ds(tmp10 + tmp3, S)); o[7] = put(ds(tmp10 - tmp3, S));
      o[1] = put(ds(tmp11 + tmp2, S)); o[6] = put(ds(tmp11 - tmp2, S));
      o[2] = put(ds(tmp12 + tmp1, S)); o[5] = put(ds(tmp12 - tmp1, S));
      o[3] = put(ds(tmp13 + tmp0, S)); o[4] = put(ds(tmp13 - tmp0, S));
    }
  }

  void decode_scan() {
    const int mcuw = 8 * hmax_, mcuh = 8 * vmax_;
    const int mx = (w_ + mcuw - 1) / mcuw, my = (h_ + mcuh - 1) / mcuh;
    for (int i = 0; i < ncomp_; ++i) {
      Component& c = comp_[i];
      c.wblocks = mx * c.h;
      c.hblocks = my * c.v;
      c.dw = (w_ * c.h + hmax_ - 1) / hmax_;
      c.dh = (h_ * c.v + vmax_ - 1) / vmax_;
      c.plane.assign((size_t)c.wblocks * 8 * c.hblocks * 8, 0);
      c.pred = 0;
    }
    int until_restart = restart_, next_rst = 0;
    for (int y = 0; y < my; ++y)
      for (int x = 0; x < mx; ++x) {
        if (restart_ && until_restart == 0) {
          // byte-align, expect RSTn
          bitbuf_ = 0; bitcnt_ = 0; hit_marker_ = false;
          while (pos_ < d_.size() && d_[pos_] != 0xFF) ++pos_;
          while (pos_ < d_.size() && d_[pos_] == 0xFF) ++pos_;
          if (pos_ >= d_.size() || d_[pos_] != 0xD0 + next_rst) fail("missing restart marker");
          ++pos_;
          next_rst = (next_rst + 1) & 7;
          for (int i = 0; i < ncomp_; ++i) comp_[i].pred = 0;
          until_restart = restart_;
        }
        for (int i = 0; i < ncomp_; ++i)
          for (int by = 0; by < comp_[i].v; ++by)
            for (int bx = 0; bx < comp_[i].h; ++bx) decode_block(comp_[i], x * comp_[i].h + bx, y * comp_[i].v + by);
        --until_restart;
      }
  }

  // jdsample.c: full-size plane of a chroma component (fancy upsampling), w_ x h_
  std::vector<uint8_t> upsample(const Component& c) const {
    const int stride = c.wblocks * 8;
    std::vector<uint8_t> out((size_t)w_ * h_);
    const int hs = hmax_ / c.h, vs = vmax_ / c.v;
    auto row = [&](int r) { return &c.plane[(size_t)std::min(std::max(r, 0), c.dh - 1) * stride]; };
    if (hs == 1 && vs == 1) {
      for (int y = 0; y < h_; ++y) std::memcpy(&out[(size_t)y * w_], row(y), (size_t)w_);
    } else if (hs == 2 && vs == 1) {  // h2v1_fancy_upsample
      std::vector<uint8_t> line((size_t)c.dw * 2);
      for (int y = 0; y < h_; ++y) {
        const uint8_t* in = row(y);
        const int n = c.dw;
        if (n == 1) { line[0] = line[1] = in[0]; }
        else {
          line[0] = in[0];
          line[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
          for (int i = 1; i < n - 1; ++i) {
            const int v = in[i] * 3;
            line[2 * i] = (uint8_t)((v + in[i - 1] + 1) >> 2);
            line[2 * i + 1] = (uint8_t)((v + in[i + 1] + 2) >> 2);
          }
          line[2 * n - 2] = (uint8_t)((in[n - 1] * 3 + in[n - 2] + 1) >> 2);
          line[2 * n - 1] = in[n - 1];
        }
        std::memcpy(&out[(size_t)y * w_], line.data(), (size_t)w_);
      }
    } else if (hs == 2 && vs == 2) {  // h2v2_fancy_upsample: 3/4 of the nearer row + 1/4 of the further, then the same across
      std::vector<uint8_t> line((size_t)c.dw * 2);
      std::vector<int> cs((size_t)c.dw);
      for (int y = 0; y < h_; ++y) {
        const int r = y >> 1;
        const uint8_t* in0 = row(r);
        const uint8_t* in1 = row((y & 1) ? r + 1 : r - 1);
        const int n = c.dw;
        for (int i = 0; i < n; ++i) cs[i] = in0[i] * 3 + in1[i];
        if (n == 1) { line[0] = (uint8_t)((cs[0] * 4 + 8) >> 4); line[1] = (uint8_t)((cs[0] * 4 + 7) >> 4); }
        else {
          line[0] = (uint8_t)((cs[0] * 4 + 8) >> 4);
          line[1] = (uint8_t)((cs[0] * 3 + cs[1] + 7) >> 4);
          for (int i = 1; i < n - 1; ++i) {
            line[2 * i] = (uint8_t)((cs[i] * 3 + cs[i - 1] + 8) >> 4);
            line[2 * i + 1] = (uint8_t)((cs[i] * 3 + cs[i + 1] + 7) >> 4);
          }
          line[2 * n - 2] = (uint8_t)((cs[n - 1] * 3 + cs[n - 2] + 8) >> 4);
          line[2 * n - 1] = (uint8_t)((cs[n - 1] * 4 + 7) >> 4);
        }
        std::memcpy(&out[(size_t)y * w_], line.data(), (size_t)w_);
      }
    } else {
      fail("unsupported chroma subsampling");
    }
    return out;
  }

  pngio::Image finish() {
    pngio::Image im;
    im.w = w_; im.h = h_; im.c = 3;
    im.px.resize((size_t)w_ * h_ * 3);
    if (ncomp_ == 1) {
      const int stride = comp_[0].wblocks * 8;
      for (int y = 0; y < h_; ++y)
        for (int x = 0; x < w_; ++x) {
          const uint8_t g = comp_[0].plane[(size_t)y * stride + x];
          uint8_t* p = &im.px[((size_t)y * w_ + x) * 3];
          p[0] = p[1] = p[2] = g;
        }
      return im;
    }
    if (adobe_transform_ == 0) fail("RGB-coded JPEG (Adobe transform 0) is not supported");
    const std::vector<uint8_t> Y = upsample(comp_[0]), Cb = upsample(comp_[1]), Cr = upsample(comp_[2]);
    // jdcolor.c build_ycc_rgb_table: SCALEBITS 16, FIX(x) = (int)(x * 65536 + 0.5)
    int crr[256], cbb[256];
    long crg[256], cbg[256];
    for (int i = 0; i < 256; ++i) {
      const long x = i - 128;
      crr[i] = (int)((91881 * x + 32768) >> 16);
      cbb[i] = (int)((116130 * x + 32768) >> 16);
      crg[i] = -46802 * x;
      cbg[i] = -22554 * x + 32768;
    }
    auto cl = [](int v) -> uint8_t { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); };
    for (size_t i = 0; i < (size_t)w_ * h_; ++i) {
      const int y = Y[i], cb = Cb[i], cr = Cr[i];
      uint8_t* p = &im.px[i * 3];
      p[2] = cl(y + crr[cr]);
      p[1] = cl(y + (int)((cbg[cb] + crg[cr]) >> 16));
      p[0] = cl(y + cbb[cb]);
    }
    return im;
  }
};

inline pngio::Image read(const std::string& path) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("failed to load image: " + path);
  std::vector<uint8_t> file;
  uint8_t buf[1 << 16];
  size_t n;
  while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + n);
  std::fclose(f);
  return Reader(file, path).decode();
}

// what imread does: the decoder is chosen by the file's signature, not by its name
inline pngio::Image read_any(const std::string& path, bool keep_alpha) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("failed to load image: " + path);
  uint8_t sig[2] = {0, 0};
  const size_t got = std::fread(sig, 1, 2, f);
  std::fclose(f);
  if (got == 2 && sig[0] == 0xFF && sig[1] == 0xD8) return read(path);
  return pngio::read(path, keep_alpha);
}

inline void read_any_into(const std::string& path, bool keep_alpha, pngio::Image& im) {  // ... into a recycled image
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("failed to load image: " + path);
  uint8_t sig[2] = {0, 0};
  const size_t got = std::fread(sig, 1, 2, f);
  std::fclose(f);
  if (got == 2 && sig[0] == 0xFF && sig[1] == 0xD8) im = read(path);
  else pngio::read_into(path, keep_alpha, im);
}

}  // namespace jpegio
