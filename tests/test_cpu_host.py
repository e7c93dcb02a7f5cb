"""Host-side pieces that need no GPU: the PNG codec of the host binary against PIL, and its command-line checks."""
import os
import subprocess

import numpy as np
import pytest
from PIL import Image

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SNIPPET = r'''
#include "png_io.hpp"
int main(int argc, char** argv) {  // argv: in.png keep_alpha out.png
  pngio::Image im = pngio::read(argv[1], argv[2][0] == '1');
  pngio::write(argv[3], im.px.data(), im.w, im.h, im.c);
  std::printf("%d %d %d\n", im.w, im.h, im.c);
  return 0;
}
'''


def test_png_codec_round_trip(tmp_path):
    src = tmp_path / "rt.cpp"
    src.write_text(SNIPPET)
    exe = str(tmp_path / "rt")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "host"), "-o", exe, str(src), "-lz",
                           "-lpthread"])
    rng = np.random.default_rng(5)
    for mode, ch in (("RGB", 3), ("RGBA", 4), ("L", 1), ("P", 1)):
        a = rng.integers(0, 256, (37, 53, ch), dtype=np.uint8)
        img = Image.fromarray(a if ch > 1 else a[:, :, 0], "L" if ch == 1 else mode)
        if mode == "P":
            img = Image.fromarray(rng.integers(0, 256, (37, 53, 3), dtype=np.uint8), "RGB").quantize(64)
        p_in, p_out = str(tmp_path / ("in_%s.png" % mode)), str(tmp_path / ("out_%s.png" % mode))
        img.save(p_in)
        for keep in ("0", "1"):
            out = subprocess.check_output([exe, p_in, keep, p_out], text=True).split()
            want_c = 4 if (keep == "1" and mode == "RGBA") else 3
            assert [int(v) for v in out] == [53, 37, want_c]
            got = np.asarray(Image.open(p_out))
            want = np.asarray(img.convert("RGBA" if want_c == 4 else "RGB"))
            assert np.array_equal(got, want), (mode, keep)


def test_png_reader_bit_depths_and_interlace(tmp_path):
    """Inputs the reference's imread accepts beyond 8-bit non-interlaced: 16-bit (high byte kept), 1/2/4-bit grey and
    palette, Adam7 interlace. Files are written by PIL (libpng) / by hand-built chunks and read back by our codec."""
    import struct
    import zlib
    src = tmp_path / "rt.cpp"
    src.write_text(SNIPPET)
    exe = str(tmp_path / "rt")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "host"), "-o", exe, str(src), "-lz",
                           "-lpthread"])
    rng = np.random.default_rng(6)

    def png(w, h, depth, ctype, rows, interlace=0, extra=b""):
        def ch(t, d):
            return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
        raw = b"".join(b"\0" + r for r in rows)
        return (b"\x89PNG\r\n\x1a\n" + ch(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, interlace)) + extra +
                ch(b"IDAT", zlib.compress(raw)) + ch(b"IEND", b""))

    def run(path):
        out = str(path) + ".out.png"
        dims = subprocess.check_output([exe, str(path), "0", out], text=True).split()
        return [int(v) for v in dims], np.asarray(Image.open(out))

    # 16-bit RGB: high byte is kept
    a16 = rng.integers(0, 65536, (9, 13, 3), dtype=np.uint16)
    p = tmp_path / "rgb16.png"
    p.write_bytes(png(13, 9, 16, 2, [a16[y].astype(">u2").tobytes() for y in range(9)]))
    dims, got = run(p)
    assert dims == [13, 9, 3] and np.array_equal(got, (a16 >> 8).astype(np.uint8))
    # 16-bit grey
    g16 = rng.integers(0, 65536, (7, 10), dtype=np.uint16)
    p = tmp_path / "g16.png"
    p.write_bytes(png(10, 7, 16, 0, [g16[y].astype(">u2").tobytes() for y in range(7)]))
    dims, got = run(p)
    assert np.array_equal(got, np.repeat((g16 >> 8).astype(np.uint8)[..., None], 3, 2))
    # 1/2/4-bit grey (scaled to 0..255) through PIL for the 1-bit case and by hand for 2/4
    for depth in (1, 2, 4):
        v = rng.integers(0, 1 << depth, (6, 11))
        rows = []
        for y in range(6):
            bits = "".join(format(int(x), "0%db" % depth) for x in v[y])
            bits += "0" * (-len(bits) % 8)
            rows.append(int(bits, 2).to_bytes(len(bits) // 8, "big"))
        p = tmp_path / ("g%d.png" % depth)
        p.write_bytes(png(11, 6, depth, 0, rows))
        dims, got = run(p)
        want = (v * 255 // ((1 << depth) - 1)).astype(np.uint8)
        assert np.array_equal(got, np.repeat(want[..., None], 3, 2)), depth
        assert np.array_equal(np.asarray(Image.open(p).convert("RGB")), got), depth  # libpng agrees
    # Adam7-interlaced RGBA, written by hand: pass p holds the pixels (x0 + i*dx, y0 + j*dy)
    a = rng.integers(0, 256, (11, 13, 4), dtype=np.uint8)
    passes = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]
    rows = []
    for (x0, y0, dx, dy) in passes:
        sub = a[y0::dy, x0::dx]
        if sub.size:
            rows += [sub[j].tobytes() for j in range(sub.shape[0])]
    p = tmp_path / "adam7.png"
    p.write_bytes(png(13, 11, 8, 6, rows, interlace=1))
    assert np.array_equal(np.asarray(Image.open(p)), a)  # the hand-built file is a valid interlaced PNG
    dims, got = run(p)
    assert dims == [13, 11, 3] and np.array_equal(got, a[..., :3])
    # truncated / lying headers are rejected, not trusted
    bad = tmp_path / "bad.png"
    bad.write_bytes(b"\x89PNG\r\n\x1a\n" + struct.pack(">I", 4) + b"IHDR" + b"\0\0\0\1" + b"\0\0\0\0")
    assert subprocess.run([exe, str(bad), "0", str(tmp_path / "x.png")], capture_output=True).returncode != 0


def test_png_reader_every_filter_type_across_bands(tmp_path):
    """Scanlines filtered with all five PNG filter types in random order (PIL picks filters by heuristic, the hand-built
    files above use none): pixel sizes 1, 3, 4, 6 and 8 bytes — 3 and 4 take the reader's SSE2 Paeth path — in images
    taller than one of the reader's 256 KB inflate bands, so that a band's first row is predicted from the previous
    band's last. Also: scanline data beyond the image, and a stream that ends early, are corrupt files."""
    import struct
    import zlib
    src = tmp_path / "rt.cpp"
    src.write_text(SNIPPET)
    exe = str(tmp_path / "rt")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "host"), "-o", exe, str(src), "-lz", "-lpthread"])
    rng = np.random.default_rng(11)

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))

    def filtered(rows, bpp):  # rows: list of bytes; returns the filtered scanlines incl. their filter-type bytes
        out, prev = [], np.zeros(len(rows[0]), np.int32)
        for r in rows:
            cur = np.frombuffer(r, np.uint8).astype(np.int32)
            a = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
            c = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
            ft = int(rng.integers(0, 5))
            if ft == 0:
                pred = 0
            elif ft == 1:
                pred = a
            elif ft == 2:
                pred = prev
            elif ft == 3:
                pred = (a + prev) >> 1
            else:
                pp = a + prev - c
                pa, pb, pc = np.abs(pp - a), np.abs(pp - prev), np.abs(pp - c)
                pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
            out.append(bytes([ft]) + ((cur - pred) & 255).astype(np.uint8).tobytes())
            prev = cur
        return out

    def png(w, h, depth, ctype, lines, tail=b"", cut=0):
        raw = b"".join(lines) + tail
        z = zlib.compress(raw[:len(raw) - cut] if cut else raw, 1)
        return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) +
                chunk(b"IDAT", z[:len(z) // 2]) + chunk(b"IDAT", z[len(z) // 2:]) + chunk(b"IEND", b""))

    for name, w, h, depth, ctype, ch in (("rgb", 601, 300, 8, 2, 3), ("rgba", 403, 330, 8, 6, 4), ("grey", 1111, 260, 8, 0, 1),
                                         ("rgb16", 207, 240, 16, 2, 3), ("rgba16", 101, 350, 16, 6, 4), ("one_px_wide", 1, 700, 8, 2, 3)):
        a = rng.integers(0, 1 << depth, (h, w, ch), dtype=np.uint16)
        # smooth rows so that Paeth / Average see every branch, noise columns so that nothing is constant
        a[:, :, 0] = (a[:, :, 0] // 7 + np.arange(w)[None, :] * 3) % (1 << depth)
        rows = [(a[y].astype(">u2") if depth == 16 else a[y].astype(np.uint8)).tobytes() for y in range(h)]
        bpp = ch * depth // 8
        p = tmp_path / (name + ".png")
        lines = filtered(rows, bpp)
        p.write_bytes(png(w, h, depth, ctype, lines))
        if h <= 300:  # the tests' own hand parser of our 16-bit outputs (refprog.png_unfilter) undoes the same five filters
            import refprog
            packed = np.frombuffer(b"".join(lines), np.uint8).reshape(h, 1 + w * bpp)
            assert refprog.png_unfilter(packed, bpp).tobytes() == b"".join(rows), name
        want8 = (a >> 8 if depth == 16 else a).astype(np.uint8)
        pil = np.asarray(Image.open(p))
        if depth == 8:  # the hand-filtered file is a valid PNG: libpng reads back the pixels it was made from
            assert np.array_equal(pil.reshape(h, w, ch), want8), name
        for keep in ("0", "1"):
            out = str(p) + ".out.png"
            dims = [int(v) for v in subprocess.check_output([exe, str(p), keep, out], text=True).split()]
            wc = 4 if (keep == "1" and ch == 4) else 3
            assert dims == [w, h, wc], name
            got = np.asarray(Image.open(out))
            want = np.repeat(want8, 3, 2) if ch == 1 else want8[..., :wc]
            assert np.array_equal(got, want), (name, keep)
    # more scanline data than the header's height, and less
    a = rng.integers(0, 256, (40, 50, 3), dtype=np.uint8)
    lines = filtered([a[y].tobytes() for y in range(40)], 3)
    for label, kw in (("extra", dict(tail=b"\0" * 151)), ("short", dict(cut=200))):
        p = tmp_path / (label + ".png")
        p.write_bytes(png(50, 40, 8, 2, lines, **kw))
        r = subprocess.run([exe, str(p), "0", str(tmp_path / "x.png")], capture_output=True, text=True)
        assert r.returncode != 0, label


BANDS_SNIPPET = r'''
#include "png_io.hpp"
int main(int argc, char** argv) {  // argv: in.png out.png threads
  pngio::Image im = pngio::read(argv[1], true);
  pngio::write(argv[2], im.px.data(), im.w, im.h, im.c, 1, std::atoi(argv[3]));
  return 0;
}
'''


def test_png_writer_parallel_bands(tmp_path):
    """The writer deflates bands of ~2 MB of scanlines independently (sync-flushed raw deflate streams concatenated
    into one zlib stream, Adler-32 combined): images of several bands, a ragged last band, 1 / 3 / many threads must
    all decode (PIL = libpng/zlib, and our own reader) to the source pixels."""
    src = tmp_path / "bands.cpp"
    src.write_text(BANDS_SNIPPET)
    exe = str(tmp_path / "bands")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "host"), "-o", exe, str(src), "-lz",
                           "-lpthread"])
    rng = np.random.default_rng(9)
    for (h, w, ch) in ((1500, 1601, 3), (1111, 1203, 4), (3, 5, 3)):
        yy, xx = np.mgrid[0:h, 0:w]
        a = ((np.sin(xx * 0.013)[..., None] * np.cos(yy * 0.007)[..., None] * 90 + 128) +
             rng.integers(-9, 10, (h, w, ch))).clip(0, 255).astype(np.uint8)
        p_in = str(tmp_path / ("big_%d.png" % ch))
        Image.fromarray(a, "RGB" if ch == 3 else "RGBA").save(p_in)
        outs = []
        for threads in ("1", "3", "0"):
            p_out = str(tmp_path / ("big_%d_%s.png" % (ch, threads)))
            subprocess.check_call([exe, p_in, p_out, threads])
            got = np.asarray(Image.open(p_out))
            assert np.array_equal(got, a), (h, w, ch, threads)
            outs.append(open(p_out, "rb").read())
            subprocess.check_call([exe, p_out, str(tmp_path / "again.png"), "1"])  # our reader accepts its own output
            assert np.array_equal(np.asarray(Image.open(str(tmp_path / "again.png"))), a)
        assert outs[0] == outs[1] == outs[2]  # the file does not depend on the number of threads
        if h > 1000:
            # The banded files are inflated band by band as RAW deflate on the parallel path: the stream's Adler-32 (the trailing
            # 4-byte IDAT) must still be checked there. A file whose trailer is wrong — chunk CRC valid — is refused, like zlib
            # refuses it on the sequential path (a damaged state image must not feed temporal regularisation).
            import struct
            import zlib
            data = outs[0]
            assert b"sbNd" in data
            iend = data.rindex(b"IEND") - 4
            tail = data[iend - 16:iend]  # length(4) "IDAT" adler(4) crc(4)
            assert tail[:8] == struct.pack(">I", 4) + b"IDAT"
            wrong = bytes([tail[8] ^ 0x40]) + tail[9:12]
            bad = data[:iend - 16] + tail[:8] + wrong + struct.pack(">I", zlib.crc32(b"IDAT" + wrong)) + data[iend:]
            p_bad = str(tmp_path / "bad_adler.png")
            open(p_bad, "wb").write(bad)
            for threads in ("1", "3"):
                r = subprocess.run([exe, p_bad, str(tmp_path / "never.png"), threads], capture_output=True, text=True)
                assert r.returncode != 0 and "corrupt PNG" in r.stderr, (threads, r.stderr[-300:])


def test_required_flags_and_unknown_flags():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "surround360_amd", "csrc"), "-j8", "-s"])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])
    exe = os.path.join(ROOT, "host", "TestRenderStereoPanorama")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode != 0 and "rig_json_file" in r.stderr  # requireArg order of TRSP:717
    r = subprocess.run([exe, "--bogus"], capture_output=True, text=True)
    assert r.returncode == 1 and "unknown command line flag" in r.stderr
    r = subprocess.run([exe, "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "--eqr_width" in r.stdout and "--prev_frame_data_dir" in r.stdout
    assert "--bin_list" in r.stdout and "--isp_dir" in r.stdout
    # --bin_list replaces imgs_dir and needs the ISP configurations (the renderer fed from the capture's containers)
    rig = os.path.join(ROOT, "tests", "golden", "rig_17cam.json")
    r = subprocess.run([exe, "--rig_json_file", rig, "--frame_number", "000000"], capture_output=True, text=True)
    assert r.returncode != 0 and "imgs_dir" in r.stderr
    r = subprocess.run([exe, "--rig_json_file", rig, "--bin_list", "a.bin", "--frame_number", "000000"], capture_output=True, text=True)
    assert r.returncode != 0 and "isp_dir" in r.stderr and "imgs_dir" not in r.stderr


def test_optical_flow_harness_flags(tmp_path):
    """host/TestOpticalFlow: requireArg order and messages of TestOpticalFlow.cpp:226-244."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "surround360_amd", "csrc"), "-j8", "-s"])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])
    exe = os.path.join(ROOT, "host", "TestOpticalFlow")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode != 0 and "missing required command line argument: mode" in r.stderr
    r = subprocess.run([exe, "--mode", "video"], capture_output=True, text=True)
    assert r.returncode != 0 and "unrecongized mode" in r.stderr
    r = subprocess.run([exe, "--mode", "test", "--test_dir", str(tmp_path)], capture_output=True, text=True)
    assert r.returncode != 0 and "left_img" in r.stderr
    r = subprocess.run([exe, "--mode", "test", "--test_dir", str(tmp_path), "--left_img", "l.png", "--right_img", "r.png",
                        "--flow_alg", "pixflow_low"], capture_output=True, text=True)
    assert r.returncode != 0 and "failed to load image" in r.stderr
    r = subprocess.run([exe, "--nope", "1"], capture_output=True, text=True)
    assert r.returncode == 1 and "unknown command line flag" in r.stderr


SNIPPET16 = r'''
#include "png_io.hpp"
int main(int argc, char** argv) {  // argv: gray_in.png rgb16_out.png : grey samples g -> 16-bit pixel (B, G, R) = (g, g ^ 0x5a5a, ~g)
  int w, h, depth;
  std::vector<uint16_t> g = pngio::read_gray(argv[1], &w, &h, &depth);
  std::vector<uint16_t> px((size_t)w * h * 3);
  for (size_t i = 0; i < g.size(); ++i) { px[3 * i] = g[i]; px[3 * i + 1] = g[i] ^ 0x5a5a; px[3 * i + 2] = (uint16_t)~g[i]; }
  pngio::write16(argv[2], px.data(), w, h);
  std::printf("%d %d %d\n", w, h, depth);
  return 0;
}
'''


def test_png_16bit_gray_read_and_rgb16_write(tmp_path):
    """The ISP host binary's I/O: greyscale raw images keep their depth (imread GRAYSCALE | ANYDEPTH), the 16-bit BGR
    result is written as a 16-bit RGB PNG. Checked against files written / read by PIL and by a hand-built decoder."""
    import struct
    import zlib
    src = tmp_path / "rt16.cpp"
    src.write_text(SNIPPET16)
    exe = str(tmp_path / "rt16")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "host"), "-o", exe, str(src), "-lz",
                           "-lpthread"])
    rng = np.random.default_rng(16)
    for depth, dt in ((16, np.uint16), (8, np.uint8)):
        g = rng.integers(0, 1 << depth, (41, 67), dtype=np.uint32).astype(dt)
        p_in, p_out = str(tmp_path / ("g%d.png" % depth)), str(tmp_path / ("o%d.png" % depth))
        Image.fromarray(g, "I;16" if depth == 16 else "L").save(p_in)
        out = subprocess.check_output([exe, p_in, p_out], text=True).split()
        assert [int(v) for v in out] == [67, 41, depth]
        data = open(p_out, "rb").read()  # decode the 16-bit RGB PNG by hand (PIL reduces 16-bit RGB to 8 bits)
        assert data[:8] == b"\x89PNG\r\n\x1a\n"
        pos, idat, ihdr = 8, b"", None
        while pos < len(data):
            n, t = struct.unpack(">I4s", data[pos:pos + 8])
            body = data[pos + 8:pos + 8 + n]
            assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(t + body)
            if t == b"IHDR":
                ihdr = struct.unpack(">IIBBBBB", body)
            elif t == b"IDAT":
                idat += body
            pos += 12 + n
        assert ihdr == (67, 41, 16, 2, 0, 0, 0)
        raw = zlib.decompress(idat)
        rows = np.frombuffer(raw, np.uint8).reshape(41, 1 + 67 * 6)
        # the writer filters its scanlines: Sub on every row, like cv::imwrite's PngEncoder (undone here by hand)
        assert (rows[:, 0] == 1).all()
        data = rows[:, 1:].astype(np.uint8).copy()
        for i in range(6, data.shape[1]):
            data[:, i] = (data[:, i].astype(np.int32) + data[:, i - 6]) & 255
        rgb = data.reshape(41, 67, 3, 2).astype(np.uint16)
        rgb = (rgb[..., 0] << 8) | rgb[..., 1]
        g16 = g.astype(np.uint16)
        assert np.array_equal(rgb[..., 2], g16) and np.array_equal(rgb[..., 1], g16 ^ 0x5a5a)
        assert np.array_equal(rgb[..., 0], ~g16)


def test_raw2rgb_flags(tmp_path):
    """host/Raw2Rgb: requireArg order and messages of Raw2Rgb.cpp:377-381, input checks; without a GPU the run ends at
    s360_isp_create with the library's "no HIP device" error (there is no CPU path)."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "surround360_amd", "csrc"), "-j8", "-s"])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])
    exe = os.path.join(ROOT, "host", "Raw2Rgb")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode != 0 and "missing required command line argument: input_image_path" in r.stderr
    r = subprocess.run([exe, "--input_image_path", "x.png", "--output_image_path", "y.png"], capture_output=True, text=True)
    assert r.returncode != 0 and "isp_config_path" in r.stderr
    cfg = tmp_path / "isp.json"
    cfg.write_text('{"CameraIsp": {"bayerPattern": "GBRG", "width": 16, "height": 12}}')
    r = subprocess.run([exe, "--input_image_path", str(tmp_path / "none.png"), "--output_image_path", str(tmp_path / "o.png"),
                        "--isp_config_path", str(cfg)], capture_output=True, text=True)
    assert r.returncode != 0 and "failed to load image" in r.stderr
    Image.fromarray(np.zeros((12, 16, 3), np.uint8), "RGB").save(str(tmp_path / "rgb.png"))
    r = subprocess.run([exe, "--input_image_path", str(tmp_path / "rgb.png"), "--output_image_path", str(tmp_path / "o.png"),
                        "--isp_config_path", str(cfg)], capture_output=True, text=True)
    assert r.returncode != 0 and "greyscale" in r.stderr
    bad = tmp_path / "bad.json"
    bad.write_text('{"CameraIsp": {"bayerPattern": "QQQQ"}}')
    Image.fromarray(np.zeros((12, 16), np.uint8), "L").save(str(tmp_path / "g.png"))
    r = subprocess.run([exe, "--input_image_path", str(tmp_path / "g.png"), "--output_image_path", str(tmp_path / "o.png"),
                        "--isp_config_path", str(bad)], capture_output=True, text=True)
    assert r.returncode != 0 and "bayerPattern" in r.stderr


@pytest.mark.parametrize("bits", [8, 12])
def test_unpacker_raw_path(tmp_path, bits):
    """host/Unpacker without ISP configs (Unpacker.cpp:156-161 "we can still unpack raws"): container parsing
    (BinaryFootageFile.cpp), the frame range flags, per-serial directories, the widened 16-bit TIFFs against the
    oracle's RawConverter, and the serial -> camN renaming of the output directory. Needs no GPU: no ISP object is made."""
    import isputil
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "surround360_amd", "csrc"), "-j8", "-s"])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])
    exe = os.path.join(ROOT, "host", "Unpacker")
    w, h, nf = 32, 12, 4
    serials = [17430921, 16241093, 17431022]  # not in sorted order: cam0 is the smallest serial
    rng = np.random.default_rng(bits)
    frames = [[rng.integers(0, 65536, (h, w), dtype=np.uint16) for _ in serials] for _ in range(nf)]
    binp = tmp_path / "0.bin"
    written = isputil.footage_file(str(binp), frames, bits, serials)
    out, raw, isp = tmp_path / "rgb", tmp_path / "raw", tmp_path / "isp"
    for d in (out, raw, isp):
        d.mkdir()
    r = subprocess.run([exe, "--isp_dir", str(isp), "--output_dir", str(out), "--output_raw_dir", str(raw),
                        "--bin_list", str(binp), "--start_frame", "1", "--frame_count", "2"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "numberOfCameras = 3" in r.stdout and "bpp = %d" % bits in r.stdout
    assert "Cannot convert to RGB, file not found" in r.stderr
    assert sorted(os.listdir(out)) == ["cam0", "cam1", "cam2"]  # renamed (and empty: no ISP config for any serial)
    assert sorted(os.listdir(raw)) == sorted(str(s) for s in serials)  # the raw directory keeps the serial names
    for cam, s in enumerate(serials):
        assert sorted(os.listdir(raw / str(s))) == ["000001.tiff", "000002.tiff"]
        for f in (1, 2):
            got = np.array(Image.open(str(raw / str(s) / ("%06d.tiff" % f))))
            assert got.dtype == np.uint16 and got.shape == (h, w)
            assert np.array_equal(got, O.isp_unpack_frame(written[f][cam], bits, w, h))
    # frame range errors and clamps (Unpacker.cpp:113-129)
    r = subprocess.run([exe, "--isp_dir", str(isp), "--output_dir", str(out), "--bin_list", str(binp), "--start_frame", "9"],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "Start frame (9) larger than total number of frames (3)" in r.stderr
    r = subprocess.run([exe, "--isp_dir", str(isp), "--output_dir", str(out), "--bin_list", str(binp), "--start_frame", "3",
                        "--frame_count", "5"], capture_output=True, text=True)
    assert r.returncode == 0 and "End frame (7) larger than total number of frames (3)" in r.stderr
    r = subprocess.run([exe, "--output_dir", str(out), "--bin_list", str(binp)], capture_output=True, text=True)
    assert r.returncode != 0 and "missing required command line argument: isp_dir" in r.stderr
    r = subprocess.run([exe, "--isp_dir", str(isp), "--output_dir", str(out), "--bin_list", str(tmp_path / "none.bin")],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "Error opening file" in r.stderr


def test_unpacker_rejects_hostile_headers(tmp_path):
    """An untrusted .bin container: header fields that would wrap the frame-size arithmetic around (the reference's
    BinaryFootageFile.cpp trusts them), an odd width with 12-bit packing, negative frame ranges — each an error message,
    never a read outside the mapping."""
    import struct
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "surround360_amd", "csrc"), "-j8", "-s"])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])
    exe = os.path.join(ROOT, "host", "Unpacker")
    out, raw, isp = tmp_path / "rgb", tmp_path / "raw", tmp_path / "isp"
    for d in (out, raw, isp):
        d.mkdir()

    def run(width, height, bpp, ncam, payload=4096, extra=()):
        binp = tmp_path / "h.bin"
        page = struct.pack("<8I", 0xfaceb00c, 0, 0, 1, width, height, bpp, ncam).ljust(4096, b"\0")
        binp.write_bytes(page + bytes(payload))
        return subprocess.run([exe, "--isp_dir", str(isp), "--output_dir", str(out), "--output_raw_dir", str(raw),
                               "--bin_list", str(binp)] + list(extra), capture_output=True, text=True)
    # width * height * bpp / 8 wraps to a small number in 64 bits: 2^31 * 2^31 * 8 / 8 = 2^62, times 12 wraps
    for wd, ht, bpp in ((0x80000000, 0x80000000, 12), (0xffffffff, 0xffffffff, 8), (0, 16, 8), (16, 0, 8), (70000, 16, 8)):
        r = run(wd, ht, bpp, 1)
        assert r.returncode != 0 and "implausible metadata" in r.stderr, (wd, ht, bpp, r.stderr)
    r = run(16, 16, 8, 100000)
    assert r.returncode != 0 and "implausible metadata" in r.stderr
    r = run(15, 4, 12, 1)
    assert r.returncode != 0 and "even width" in r.stderr
    r = run(16, 4, 7, 1)
    assert r.returncode != 0 and "unsupported bits per pixel" in r.stderr
    r = run(16, 4, 8, 1, extra=("--start_frame", "-1"))
    assert r.returncode != 0 and "must not be negative" in r.stderr
    r = run(16, 4, 8, 1, extra=("--frame_count", "-3"))
    assert r.returncode != 0 and "must not be negative" in r.stderr
    r = run(16, 4, 8, 0)  # "No cameras found...": accepted, nothing to do
    assert r.returncode == 0 and "No cameras found" in r.stderr
    r = run(64, 64, 8, 2, payload=64 * 64)  # not even one frame per camera
    assert r.returncode != 0 and "larger than total number of frames" in r.stderr
    r = run(16, 4, 8, 2, payload=16 * 4 * 2 * 3 + 17)  # three whole frames per camera and a ragged tail: the tail is ignored
    assert r.returncode == 0 and len(os.listdir(raw)) == 1, r.stderr


JPEG_SNIPPET = r'''
#include "jpeg_io.hpp"
int main(int argc, char** argv) {  // argv: in.jpg out.raw
  try {
    pngio::Image im = jpegio::read_any(argv[1], false);
    FILE* f = std::fopen(argv[2], "wb");
    std::fwrite(im.px.data(), 1, im.px.size(), f);
    std::fclose(f);
    std::printf("%d %d %d\n", im.w, im.h, im.c);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  return 0;
}
'''


def test_jpeg_reader_equals_libjpeg(tmp_path):
    """host/jpeg_io.hpp against PIL (libjpeg-turbo at libjpeg's defaults, as cv::imread uses it): 4:4:4 / 4:2:2 / 4:2:0,
    qualities, sizes that are not multiples of the MCU, restart markers, greyscale — bit for bit; progressive files and an
    EXIF orientation other than 1 are rejected with a message."""
    src = tmp_path / "jpg.cpp"
    src.write_text(JPEG_SNIPPET)
    exe = str(tmp_path / "jpg")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "host"), "-o", exe, str(src), "-lz"])
    rng = np.random.default_rng(1)

    def scene(h, w):
        yy, xx = np.mgrid[0:h, 0:w]
        a = np.stack([(xx * 3 + yy) % 256, (xx + yy * 2) % 256, (xx * yy // 7) % 256], -1).astype(np.uint8)
        a[h // 4:h // 2, w // 3:w // 2] = [250, 10, 30]
        return np.clip(a.astype(int) + rng.integers(0, 40, (h, w, 3)) - 20, 0, 255).astype(np.uint8)

    def decode(p, shape):
        r = subprocess.run([exe, p, str(tmp_path / "o.raw")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        return np.fromfile(str(tmp_path / "o.raw"), np.uint8).reshape(shape)

    p = str(tmp_path / "t.jpg")
    for (h, w) in ((64, 64), (37, 53), (128, 200), (17, 16), (8, 8), (1, 1), (33, 17)):
        for sub in (0, 1, 2):
            for q, extra in ((30, {}), (75, {"restart_marker_blocks": 3}), (95, {})):
                Image.fromarray(scene(h, w)).save(p, quality=q, subsampling=sub, **extra)
                want = np.asarray(Image.open(p).convert("RGB"))[:, :, ::-1]
                assert np.array_equal(decode(p, (h, w, 3)), want), (h, w, sub, q)
    Image.fromarray(scene(40, 61)[:, :, 0]).save(p, quality=80)  # greyscale -> B = G = R
    assert np.array_equal(decode(p, (40, 61, 3)), np.asarray(Image.open(p).convert("RGB"))[:, :, ::-1])
    Image.fromarray(scene(32, 32)).save(p, progressive=True)
    r = subprocess.run([exe, p, str(tmp_path / "o.raw")], capture_output=True, text=True)
    assert r.returncode != 0 and "progressive" in r.stderr
    ex = Image.Exif()
    ex[0x0112] = 6
    Image.fromarray(scene(32, 32)).save(p, exif=ex)
    r = subprocess.run([exe, p, str(tmp_path / "o.raw")], capture_output=True, text=True)
    assert r.returncode != 0 and "orientation" in r.stderr
    ex[0x0112] = 1
    Image.fromarray(scene(32, 32)).save(p, exif=ex)
    assert np.array_equal(decode(p, (32, 32, 3)), np.asarray(Image.open(p).convert("RGB"))[:, :, ::-1])


def test_median_networks_are_the_generators_output(tmp_path):
    """surround360_amd/csrc/median_tile.inc is what tools/gen_median_network.py writes today: the generator checks every stage
    program (min / max / min3 / max3 / med3) on all 0/1 inputs of its precondition and the composition against numpy's median
    before it writes anything, so an edited or stale file cannot pass for a verified one."""
    out = tmp_path / "median_tile.inc"
    env = dict(os.environ, S360_MEDIAN_INC_OUT=str(out))
    import sys
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_median_network.py")], check=True, env=env, capture_output=True)
    assert out.read_bytes() == open(os.path.join(ROOT, "surround360_amd", "csrc", "median_tile.inc"), "rb").read()
