"""The output frame's PNG, encoded on the device (surround360_amd/csrc/png.hip; replaces imwriteExceptionOnFail's PngEncoder for
the equirect, TestRenderStereoPanorama.cpp:938-961).

The oracle of a lossless codec is a decoder that is not ours: every file must decode — through PIL (libpng + zlib: signature,
chunk CRCs, the zlib stream, its Adler-32) and through the reference's file layout reader of host/png_io.hpp (the parallel band
path) — to exactly the pixels that went in. Beside that, the size against zlib itself run with the reference encoder's settings
on the same filtered scanlines (Sub, Z_BEST_SPEED, Z_RLE — what cv::imwrite sets): the device's token set is Z_RLE's, so the
files must be about as small. Cases: smooth content, noise (bands that leave as stored blocks), flat content (distance-1
matches), one pixel, rows wider than one tile, ragged last band, band heights forced through S360_PNG_BAND_ROWS, symbol
statistics that need the 15-bit length limit, and the frame path (s360_set_png_encode + s360_frame_download_png) against
s360_frame_download_equirect incl. frame pipelining and frame slots. Replayed on the CPU emulation by
tests/test_cpu_library_emulation.py."""
import io
import os
import subprocess
import zlib

import numpy as np
import pytest
from PIL import Image

import rigutil
from surround360_amd import render as R

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CAM, EQR_W, EQR_H = 128, 252, 126


@pytest.fixture(scope="module")
def ctx(tmp_path_factory, rig_json, s360lib):
    d = tmp_path_factory.mktemp("rig_png")
    path = rigutil.scaled_rig_json(rig_json, str(d / "rig_small.json"), CAM / 2048.0)
    c = R.Context(R.RigDescription(path), R.make_params(eqr_width=EQR_W, eqr_height=EQR_H, enable_top=1, enable_bottom=1,
                                                        final_eqr_width=240, final_eqr_height=240, sharpening=0.25))
    c.rig_path = path
    yield c
    c.close()


def filtered_scanlines(bgr):
    """The bytes a PNG encoder deflates for an 8-bit RGB image with the Sub filter on every row."""
    rgb = bgr[:, :, ::-1].astype(np.int16)
    f = rgb.copy()
    f[:, 1:] -= rgb[:, :-1]
    f = (f & 255).astype(np.uint8).reshape(bgr.shape[0], -1)
    return np.concatenate([np.ones((bgr.shape[0], 1), np.uint8), f], axis=1)


def zlib_rle_bytes(bgr):
    """zlib as cv::imwrite's PngEncoder drives it (Z_BEST_SPEED, Z_RLE) over the same filtered scanlines."""
    c = zlib.compressobj(1, zlib.DEFLATED, 15, 8, zlib.Z_RLE)
    return len(c.compress(filtered_scanlines(bgr).tobytes()) + c.flush())


def image_from_filtered(f, h, w):
    """The B,G,R image whose Sub-filtered R,G,B bytes are f (h x 3w): running sums per channel along a row."""
    rgb = np.cumsum(f.reshape(h, w, 3).astype(np.int64), axis=1) & 255
    return np.ascontiguousarray(rgb[:, :, ::-1].astype(np.uint8))


def chunks(png):
    out, pos = [], 8
    while pos < len(png):
        n = int.from_bytes(png[pos:pos + 4], "big")
        out.append((png[pos + 4:pos + 8], png[pos + 8:pos + 8 + n]))
        pos += 12 + n
    return out


def decode_check(png, bgr):
    assert png[:8] == bytes([137, 80, 78, 71, 13, 10, 26, 10])
    Image.MAX_IMAGE_PIXELS = None
    im = Image.open(io.BytesIO(png))
    assert im.mode == "RGB" and im.size == (bgr.shape[1], bgr.shape[0])
    got = np.asarray(im)[:, :, ::-1]
    assert np.array_equal(got, bgr), "%d bytes differ" % int((got != bgr).sum())
    ch = chunks(png)
    assert [t for t, _ in ch[:3]] == [b"IHDR", b"sbNd", b"IDAT"] and ch[-1][0] == b"IEND" and ch[2][1] == b"\x78\x01"
    rows = int.from_bytes(ch[1][1], "big")
    bands = ch[3:-2]
    assert all(t == b"IDAT" for t, _ in bands) and len(bands) == -(-bgr.shape[0] // rows) and len(ch[-2][1]) == 4
    # every band is a raw-deflate segment of its own (what lets host/png_io.hpp's reader take the bands in parallel)
    line = 1 + 3 * bgr.shape[1]
    f = filtered_scanlines(bgr).tobytes()
    for i, (_, data) in enumerate(bands):
        d = zlib.decompressobj(-15)
        assert d.decompress(data) + d.flush() == f[i * rows * line:(i + 1) * rows * line], "band %d" % i
    return rows, len(bands)


def cases():
    rng = np.random.default_rng(11)
    h, w = 97, 333
    yy, xx = np.mgrid[0:h, 0:w]
    smooth = ((np.sin(xx * 0.05)[..., None] * np.cos(yy * 0.03)[..., None] * 90 + 128) + rng.integers(-3, 4, (h, w, 3))).clip(0, 255).astype(np.uint8)
    mixed = smooth.copy()
    mixed[20:40] = 77
    mixed[60:, 100:200] = (0, 0, 255)
    flat = np.zeros((50, 5000, 3), np.uint8)  # rows wider than one 4096-pixel tile
    flat[:] = (10, 200, 30)
    wide = ((np.sin(np.arange(9000) * 0.01)[None, :, None] * 100 + 128) + rng.integers(-2, 3, (5, 9000, 3))).clip(0, 255).astype(np.uint8)
    return {"smooth": smooth, "mixed": mixed, "noise": rng.integers(0, 256, (64, 200, 3), dtype=np.uint8), "flat": flat, "wide": wide,
            "one_pixel": np.full((1, 1, 3), 9, np.uint8), "one_column": rng.integers(0, 256, (300, 1, 3), dtype=np.uint8),
            "one_row": smooth[:1].copy()}


@pytest.mark.parametrize("name", list(cases()))
def test_encode_png_decodes_to_the_input(ctx, name):
    a = cases()[name]
    png = ctx.encode_png(a)
    rows, nb = decode_check(png, a)
    if name == "flat":  # 750 000 bytes of one colour: distance-1 matches (runs are cut at a thread's 24 bytes), not literals
        assert len(png) < a.size // 20
    if name == "noise":  # nothing to gain: stored blocks, a few bytes of framing per band
        assert len(png) <= a.size + a.shape[0] + (12 + 5) * nb + 200


def test_size_is_zlib_rle_size_on_frame_sized_bands(ctx):
    """Bands of the size an 8K frame has (8 rows of 8192 pixels = 196 KB): within 1 % of zlib's Z_RLE output on the same bytes."""
    rng = np.random.default_rng(5)
    h, w = 64, 8192
    yy, xx = np.mgrid[0:h, 0:w]
    a = ((np.sin(xx * 0.01)[..., None] * np.cos(yy * 0.13)[..., None] * 90 + 128) + rng.normal(0, 2.0, (h, w, 3))).clip(0, 255).astype(np.uint8)
    a[10:14, 1000:3000] = 0
    os.environ["S360_PNG_BAND_ROWS"] = "8"
    try:
        png = ctx.encode_png(a)
    finally:
        del os.environ["S360_PNG_BAND_ROWS"]
    rows, nb = decode_check(png, a)
    assert rows == 8 and nb == 8
    ref = zlib_rle_bytes(a)
    assert len(png) < 1.01 * ref + 2048, (len(png), ref)


@pytest.mark.parametrize("band_rows", [1, 3, 7, 1000])
def test_band_heights(ctx, band_rows):
    rng = np.random.default_rng(band_rows)
    a = np.repeat(rng.integers(0, 256, (40, 31, 3), dtype=np.uint8), 3, axis=1)
    os.environ["S360_PNG_BAND_ROWS"] = str(band_rows)
    try:
        png = ctx.encode_png(a)
    finally:
        del os.environ["S360_PNG_BAND_ROWS"]
    rows, nb = decode_check(png, a)
    assert rows == min(band_rows, 40) and nb == -(-40 // rows)


def test_length_limit(ctx):
    """Literal frequencies in Fibonacci proportions make an unrestricted Huffman tree 25 levels deep: the band's code must come
    out limited to deflate's 15 bits and still be complete (zlib refuses over-subscribed and incomplete codes)."""
    fib = [1, 1]
    while len(fib) < 26:
        fib.append(fib[-1] + fib[-2])
    vals = np.concatenate([np.full(c, 3 + 2 * i, np.uint8) for i, c in enumerate(fib)])
    rng = np.random.default_rng(2)
    rng.shuffle(vals)
    w = 2000
    h = len(vals) // (3 * w)
    a = image_from_filtered(vals[:h * 3 * w], h, w)
    os.environ["S360_PNG_BAND_ROWS"] = str(h)  # one band: one code for all of it
    try:
        png = ctx.encode_png(a)
    finally:
        del os.environ["S360_PNG_BAND_ROWS"]
    decode_check(png, a)
    assert len(png) < 0.45 * a.size  # ~2.6 bits of entropy per byte: coded, not stored


def test_our_file_reader_takes_the_bands_in_parallel(ctx, tmp_path):
    """host/png_io.hpp reads the file through its parallel band path ("sbNd") and through the sequential one: same pixels."""
    src = tmp_path / "rd.cpp"
    src.write_text(r'''
#include "png_io.hpp"
int main(int argc, char** argv) {  // argv: in.png out.raw threads
  pngio::g_read_threads = std::atoi(argv[3]);
  pngio::Image im = pngio::read(argv[1], false);
  FILE* f = std::fopen(argv[2], "wb");
  std::fwrite(im.px.data(), 1, im.px.size(), f);
  std::fclose(f);
  return im.c == 3 ? 0 : 1;
}
''')
    exe = str(tmp_path / "rd")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "host"), "-o", exe, str(src), "-lz", "-lpthread"])
    a = cases()["mixed"]
    p = tmp_path / "m.png"
    p.write_bytes(ctx.encode_png(a))
    for threads in ("3", "-1"):  # parallel bands; sequential reader only
        subprocess.check_call([exe, str(p), str(tmp_path / "m.raw"), threads])
        assert np.array_equal(np.fromfile(str(tmp_path / "m.raw"), np.uint8).reshape(a.shape), a), threads


def test_frame_png_equals_downloaded_pixels(ctx):
    """s360_set_png_encode: the rendered frame leaves as a PNG that decodes to s360_frame_download_equirect's bytes — one frame,
    a pipelined pair of frames (age 1 while the next renders), and the frame slots of a batch."""
    side, top, bottom = rigutil.frame_inputs(ctx.rig_path, CAM)
    side2 = [np.ascontiguousarray(s[:, ::-1]) for s in side]
    with pytest.raises(R.S360Error):
        ctx.upload_frame(side, top, bottom)
        ctx.render(False)
        ctx.download_png()  # rendered with the encoder off
    ctx.set_png_encode(True)
    ctx.upload_frame(side, top, bottom)
    ctx.render(False)
    want0 = ctx.download_equirect()
    decode_check(ctx.download_png().tobytes(), want0)
    # pipelined: frame 1 is enqueued, frame 0's PNG is fetched (age 1), then frame 1's (age 0)
    ctx.set_frame_pipelining(True)
    ctx.upload_frame(side, top, bottom)
    ctx.render(False)
    ctx.upload_frame(side2, top, bottom)
    ctx.render(True)
    buf = R.pinned_empty((int(R.lib().s360_frame_png_bound(ctx.h)),))
    decode_check(ctx.download_png(1, buf).tobytes(), want0)
    want1 = ctx.download_equirect()
    assert not np.array_equal(want0, want1)
    decode_check(ctx.download_png(0, buf).tobytes(), want1)
    ctx.set_frame_pipelining(False)
    # frame slots: every slot's own file
    ctx.set_frame_slots(2)
    ctx.select_frame_slot(0)
    ctx.upload_frame(side2, top, bottom)
    ctx.select_frame_slot(1)
    ctx.upload_frame(side, top, bottom)
    ctx.render_batch(False)
    outs = []
    for k in range(2):
        ctx.select_frame_slot(k)
        eq = ctx.download_equirect()
        decode_check(ctx.download_png().tobytes(), eq)
        outs.append(eq)
    assert np.array_equal(outs[1], want0) and not np.array_equal(outs[0], outs[1])
    ctx.select_frame_slot(0)
    ctx.set_frame_slots(1)
    ctx.set_png_encode(False)
    # too small a buffer is refused, not overrun
    ctx.set_png_encode(True)
    ctx.upload_frame(side, top, bottom)
    ctx.render(False)
    with pytest.raises(R.S360Error):
        ctx.download_png(0, np.empty(1000, np.uint8))
    decode_check(ctx.download_png().tobytes(), want0)
    ctx.set_png_encode(False)
