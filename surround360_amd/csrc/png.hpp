// png.hpp — the finished equirect as a PNG file, encoded on the device (png.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "core.hpp"

namespace s360 {

// Geometry of one encode: an 8-bit B,G,R image of w x h pixels (rows contiguous) becomes an 8-bit RGB PNG whose scanlines
// are deflated in bands of `rows_per_band` rows, one workgroup and one IDAT chunk per band.
struct PngPlan {
  int w = 0, h = 0, rows_per_band = 0, nbands = 0;
  size_t line = 0;         // bytes of a filtered scanline: 1 + 3 w
  size_t band_stride = 0;  // bytes reserved per band in the scratch (a band coded as stored blocks always fits)
  size_t file_bound = 0;   // upper bound of the file's size
  static PngPlan make(int w, int h);
};
// What the device leaves per band (device layout; the host reads it back before the bytes).
struct PngBandMeta {
  unsigned bytes;    // deflate bytes of the band's segment (byte aligned: sync-flushed, the last band final)
  unsigned s1, s2;   // Adler-32 pieces of the band's filtered scanlines: sum of bytes, sum of (n - i) * byte[i], both mod 65521
  unsigned stored;   // 1: the band went out as stored blocks (Huffman coding would not have made it smaller)
  unsigned long long file_off;  // where the band's IDAT chunk starts in the file image (filled by the layout kernel)
};
// Enqueues filter + deflate + layout of `bgr` on `st`: afterwards `file` (>= plan.file_bound bytes) holds the IDAT chunks of
// the bands at their final offsets — length and type fields written, the 4 CRC bytes of every chunk left for the host — and
// `meta` (plan.nbands + 1 records; the last one's file_off = the offset behind the last band chunk) the band table. `scratch`
// (nbands x band_stride) is dead once the call's kernels have run: images encoded one after the other on one stream share it.
// Nothing waits.
void png_encode_enqueue(hipStream_t st, const uint8_t* bgr, const PngPlan& plan, DevBuf& scratch, DevBuf& meta, uint8_t* file);
// Host side, after `file[0 .. meta[nbands].file_off)` and the band table have been copied to host memory: writes signature,
// IHDR, sbNd, the zlib header chunk, every band chunk's CRC, the Adler-32 chunk and IEND. Returns the file's size.
size_t png_finish_host(uint8_t* file, size_t cap, const PngPlan& plan, const PngBandMeta* meta, int crc_threads);
// offset of the first band chunk in the file (signature + IHDR + sbNd + zlib-header IDAT)
constexpr size_t kPngPreamble = 8 + 25 + 16 + 14;
uint32_t crc32_update(uint32_t crc, const uint8_t* p, size_t n);  // PNG / zlib CRC-32 (own tables, slicing by 8)

}  // namespace s360
