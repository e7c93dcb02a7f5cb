"""Shared by tests/golden/make_refprogram_golden.py, tests/test_cpu_refprogram.py and tests/test_gpu_refprogram.py: inputs
and digests for runs of a TestRenderStereoPanorama program — the reference's own (oracle/_ref/TestRenderStereoPanorama,
compiled from /root/reference over ref_shim where the reference exists) or the product's (host/TestRenderStereoPanorama).

The inputs are made with integer numpy operations only (PCG64 integers, repeats, integer box filters), so that every
machine generates the same bytes — the golden digests are of the reference program's outputs for exactly these inputs.
They are not a consistent 3-D scene: side camera k sees a window of one long wrapped texture shifted by a third of the
image per camera (adjacent cameras overlap like the rig's), the pole cameras see textures of their own, frame f moves
everything by 2 f pixels."""
import hashlib
import json
import os
import subprocess

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CAM, EQR_W, EQR_H, FINAL = 256, 504, 252, 480
GOLDEN = os.path.join(ROOT, "tests", "golden", "refprogram_golden.json")
REF_EXE = os.path.join(ROOT, "oracle", "_ref", "TestRenderStereoPanorama")
HOST_EXE = os.path.join(ROOT, "host", "TestRenderStereoPanorama")


def _box(a, r):
    """Integer box filter of radius r along both axes (wrapping), exact in int64."""
    for ax in (0, 1):
        acc = np.zeros_like(a)
        for d in range(-r, r + 1):
            acc += np.roll(a, d, axis=ax)
        a = acc // (2 * r + 1)
    return a


def _texture(h, w, seed):
    rng = np.random.default_rng(seed)
    coarse = rng.integers(0, 256, ((h + 15) // 16, (w + 15) // 16, 3), dtype=np.int64)
    fine = rng.integers(0, 256, ((h + 3) // 4, (w + 3) // 4, 3), dtype=np.int64)
    c = np.repeat(np.repeat(coarse, 16, axis=0), 16, axis=1)[:h, :w]
    f = np.repeat(np.repeat(fine, 4, axis=0), 4, axis=1)[:h, :w]
    t = (_box(c, 6) * 5 + _box(f, 1) * 3) // 8
    return np.clip((t - 128) * 2 + 128, 0, 255).astype(np.uint8)  # B,G,R


def rig_ids(rig_path):
    cams = json.load(open(rig_path))["cameras"]
    side = [c["id"] for c in cams if "side" in c.get("group", "")]
    other = [c for c in cams if "side" not in c.get("group", "")]
    top = max(other, key=lambda c: c["forward"][2])["id"]
    bottoms = [c["id"] for c in other if c["id"] != top]
    return side, top, bottoms


def frame_images(rig_path, frame_index, size=CAM):
    """{camera id: size x size x 3 uint8 B,G,R} for every camera of the rig."""
    side, top, bottoms = rig_ids(rig_path)
    step = size // 3
    pano = _texture(size, step * len(side), 360)
    pano = np.roll(pano, -2 * frame_index, axis=1)
    wide = np.concatenate([pano, pano[:, :size]], axis=1)
    out = {cid: np.ascontiguousarray(wide[:, k * step:k * step + size]) for k, cid in enumerate(side)}
    for j, cid in enumerate([top] + bottoms):
        out[cid] = np.ascontiguousarray(np.roll(_texture(size, size, 1000 + j), 2 * frame_index, axis=0))
    return out


def pole_mask(size, cx_num, cx_den):
    """Red (B,G,R = 0,0,255) wedge on white, like res/pole_masks/*.png; integer geometry."""
    yy, xx = np.mgrid[0:size, 0:size]
    m = np.full((size, size, 3), 255, np.uint8)
    red = (np.abs(xx * cx_den - size * cx_num) * 25 < cx_den * (size + yy)) & (yy * 20 > size * 7)
    m[red] = (0, 0, 255)
    return m


def write_inputs(work, rig_path, frames, masks=False):
    """imgs_dir/<camera id>/<frame>.png and the directories the caller of the program makes (batch_process_video.py:133-135)."""
    imgs, out = os.path.join(work, "rgb"), os.path.join(work, "out")
    for k, f in enumerate(frames):
        for cid, img in frame_images(rig_path, k).items():
            d = os.path.join(imgs, cid)
            os.makedirs(d, exist_ok=True)
            Image.fromarray(np.ascontiguousarray(img[:, :, ::-1])).save(os.path.join(d, f + ".png"))
        os.makedirs(os.path.join(out, "debug", f, "flow_images"), exist_ok=True)
        os.makedirs(os.path.join(out, "flow", f), exist_ok=True)
    os.makedirs(os.path.join(out, "logs"), exist_ok=True)
    mdir = None
    if masks:
        mdir = os.path.join(work, "masks")
        os.makedirs(mdir, exist_ok=True)
        _, _, bottoms = rig_ids(rig_path)
        for j, cid in enumerate(bottoms):
            Image.fromarray(np.ascontiguousarray(pole_mask(CAM, 10 - j, 20)[:, :, ::-1])).save(os.path.join(mdir, cid + ".png"))
    return imgs, out, mdir


# name -> (frames, extra flags of the program)
CASES = {
    # (with every flag scripts/batch_process_video.py:29-56 passes, glog's included)
    "two_frames": (["000000", "000001"], ["--enable_top", "--enable_bottom", "--sharpening", "0.0", "--logbuflevel", "-1",
                                          "--stderrthreshold", "0", "--v", "1", "--cubemap_format", "video", "--side_flow_alg",
                                          "pixflow_low", "--polar_flow_alg", "pixflow_low", "--poleremoval_flow_alg", "pixflow_low",
                                          "--cubemap_width", "96", "--cubemap_height", "96", "--interpupilary_dist", "6.4",
                                          "--zero_parallax_dist", "10000"]),
    "sharpen_cubemap_search": (["000000"], ["--enable_top", "--enable_bottom", "--sharpening", "0.25", "--side_flow_alg",
                                            "pixflow_search_20", "--cubemap_width", "96", "--cubemap_height", "96",
                                            "--cubemap_format", "video"]),
    "three_frames_sharpened": (["000007", "000008", "000009"], ["--enable_top", "--enable_bottom", "--sharpening", "0.25"]),
    "pole_removal": (["000000", "000001"], ["--enable_bottom", "--enable_pole_removal", "--sharpening", "0.0", "--cubemap_width",
                                            "64", "--cubemap_height", "64", "--cubemap_format", "photo"]),
}


def run_case(exe, work, rig_path, name, timeout=900, more_args=(), env=None):
    """Runs the frames of a case through the program (chained with --prev_frame_data_dir); returns the output directory.
    more_args / env: opt-in flags of OUR program only (--num_gpus ...) and variables for its process."""
    frames, extra = CASES[name]
    imgs, out, mdir = write_inputs(work, rig_path, frames, masks="--enable_pole_removal" in extra)
    prev = "NONE"
    for f in frames:
        cmd = [exe, "--rig_json_file", rig_path, "--imgs_dir", imgs, "--frame_number", f, "--output_data_dir", out,
               "--prev_frame_data_dir", prev, "--output_equirect_path", os.path.join(out, "eqr_%s.png" % f),
               "--eqr_width", str(EQR_W), "--eqr_height", str(EQR_H), "--final_eqr_width", str(FINAL),
               "--final_eqr_height", str(FINAL)] + extra
        if "--cubemap_width" in extra:
            cmd += ["--output_cubemap_path", os.path.join(out, "cube_%s.png" % f)]
        if "--logbuflevel" in extra:
            cmd += ["--log_dir", os.path.join(out, "logs")]
        if mdir:
            cmd += ["--bottom_pole_masks_dir", mdir]
        cmd += list(more_args)
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **(env or {})))
        assert r.returncode == 0, "%s frame %s: rc %d\n%s" % (name, f, r.returncode, r.stderr[-2000:])
        prev = f
    return out


def run_stream(exe, work, rig_path, name, timeout=900, first=0, count=None, more_args=(), env=None):
    """The frames of a case as ONE stream in one process (host/TestRenderStereoPanorama --num_frames N: temporal state kept
    on the device, frame pipelining, overlapped I/O); returns the output directory. Equirects eqr_<frame>.png as run_case.
    `first` / `count`: only that part of the case's frames (all inputs are written); `more_args`: e.g. --num_streams."""
    frames, extra = CASES[name]
    imgs, out, mdir = write_inputs(work, rig_path, frames)
    count = len(frames) - first if count is None else count
    cmd = [exe, "--rig_json_file", rig_path, "--imgs_dir", imgs, "--frame_number", frames[first], "--num_frames", str(count),
           "--output_data_dir", out, "--output_equirect_path", os.path.join(out, "eqr_%s.png"),
           "--eqr_width", str(EQR_W), "--eqr_height", str(EQR_H), "--final_eqr_width", str(FINAL),
           "--final_eqr_height", str(FINAL)] + extra + list(more_args)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, "%s as a stream: rc %d\n%s" % (name, r.returncode, r.stderr[-2000:])
    return out


def _digest_png(path):
    a = np.asarray(Image.open(path))
    return hashlib.sha256(repr(a.shape).encode() + a.tobytes()).hexdigest()


def digests(out, name):
    """SHA-256 of every result and state file of a case: PNGs by decoded pixels (+ shape), flow .bin files by bytes."""
    frames, extra = CASES[name]
    d = {}
    for f in frames:
        d["eqr_%s" % f] = _digest_png(os.path.join(out, "eqr_%s.png" % f))
        cube = os.path.join(out, "cube_%s.png" % f)
        if os.path.exists(cube):
            d["cube_%s" % f] = _digest_png(cube)
        for sub, kind in (("flow", ".bin"), (os.path.join("debug", f, "flow_images"), ".png")):
            folder = os.path.join(out, "flow", f) if sub == "flow" else os.path.join(out, sub)
            for fn in sorted(os.listdir(folder)):
                p = os.path.join(folder, fn)
                if fn.endswith(".bin"):
                    d["%s/%s" % (f, fn)] = hashlib.sha256(open(p, "rb").read()).hexdigest()
                elif fn.endswith(".png"):
                    d["%s/%s" % (f, fn)] = _digest_png(p)
    return d


# ---- the ISP program (camera_isp/Raw2Rgb.cpp, soft path) ------------------------------------------------------------------
REF_RAW2RGB = os.path.join(ROOT, "oracle", "_ref", "Raw2Rgb")
HOST_RAW2RGB = os.path.join(ROOT, "host", "Raw2Rgb")
RAW_W, RAW_H = 160, 96
# name -> (input depth, flags of the program)
RAW_CASES = {
    "png16_bpp16_edge_aware": (16, ["--output_bpp", "16", "--demosaic_filter", "2"]),
    "png16_bpp8_bilinear_resize2_black20": (16, ["--output_bpp", "8", "--demosaic_filter", "0", "--resize", "2", "--black_level_offset", "20"]),
    "png8_bpp8_edge_aware_no_tone_curve": (8, ["--output_bpp", "8", "--demosaic_filter", "2", "--disable_tone_curve"]),
}


# Raw2Rgb --accelerate: the reference's CameraIspPipe (its Halide pipeline; not buildable here, so no golden digests — these cases
# are compared between the reference's program on the library, the host program and the oracle's restatement)
PIPE_RAW_CASES = {
    "accelerate_bpp16": (16, ["--output_bpp", "16", "--accelerate"]),
    "accelerate_fast_bpp8_black20": (16, ["--output_bpp", "8", "--accelerate", "--fast", "--black_level_offset", "20"]),
    "accelerate_png8_bpp8_no_tone_curve": (8, ["--output_bpp", "8", "--accelerate", "--disable_tone_curve"]),
}


def bayer_frame_int(w, h, seed, pattern="GBRG"):
    """H x W uint16 Bayer mosaic of an integer-generated texture (full 16-bit range: v * 257)."""
    tex = _texture(h, w, seed).astype(np.uint16) * 257
    idx = {"R": 2, "G": 1, "B": 0}
    raw = np.zeros((h, w), np.uint16)
    for i in range(2):
        for j in range(2):
            raw[i::2, j::2] = tex[i::2, j::2, idx[pattern[i * 2 + j]]]
    return raw


def run_raw_case(exe, work, config_json, name):
    """Writes the input PNG (16-bit, or the high bytes as an 8-bit PNG) and runs the program; returns (raw16 the ISP sees, output path)."""
    depth, flags = RAW_CASES[name] if name in RAW_CASES else PIPE_RAW_CASES[name]
    os.makedirs(work, exist_ok=True)
    raw = bayer_frame_int(RAW_W, RAW_H, 77)
    cfg = os.path.join(work, "isp.json")
    open(cfg, "w").write(config_json)
    inp, outp = os.path.join(work, "in.png"), os.path.join(work, "out.png")
    if depth == 16:
        Image.fromarray(raw).save(inp)  # (uint16 -> 16-bit greyscale PNG)
        seen = raw
    else:
        raw8 = (raw >> 8).astype(np.uint8)
        Image.fromarray(raw8).save(inp)
        seen = raw8.astype(np.uint16) * 257  # convert8bitTo16bit
    r = subprocess.run([exe, "--input_image_path", inp, "--output_image_path", outp, "--isp_config_path", cfg] + flags,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, "%s: rc %d\n%s" % (name, r.returncode, r.stderr[-2000:])
    return seen, outp


def png_unfilter(rows, bpp):
    """rows: H x (1 + stride) uint8 scanlines with their filter-type bytes, as inflated from a PNG's IDAT; returns the
    H x stride unfiltered bytes (all five filter types; bpp = bytes per pixel)."""
    h, stride = rows.shape[0], rows.shape[1] - 1
    out = np.zeros((h, stride), np.int32)
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        ft, line = int(rows[y, 0]), rows[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        else:  # Sub, Average, Paeth: serial along the row, one pixel's bytes at a time
            cur = np.zeros(stride, np.int32)
            for i in range(0, stride, bpp):
                a = cur[i - bpp:i] if i else np.zeros(bpp, np.int32)
                b = prev[i:i + bpp]
                c = prev[i - bpp:i] if i else np.zeros(bpp, np.int32)
                if ft == 1:
                    pred = a
                elif ft == 3:
                    pred = (a + b) >> 1
                elif ft == 4:
                    pp = a + b - c
                    pa, pb, pc = np.abs(pp - a), np.abs(pp - b), np.abs(pp - c)
                    pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))
                else:
                    raise AssertionError("bad PNG filter type %d" % ft)
                cur[i:i + bpp] = (line[i:i + bpp] + pred) & 255
        out[y] = cur
        prev = cur
    return out.astype(np.uint8)


def png_pixels_bgr(path):
    """8- or 16-bit RGB PNG -> H x W x 3 B,G,R array of the file's depth (PIL does not decode 16-bit RGB)."""
    import struct
    import zlib
    data = open(path, "rb").read()
    pos, idat, ihdr = 8, b"", None
    while pos < len(data):
        n, t = struct.unpack(">I4s", data[pos:pos + 8])
        if t == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", data[pos + 8:pos + 8 + n])
        elif t == b"IDAT":
            idat += data[pos + 8:pos + 8 + n]
        pos += 12 + n
    w, h, depth, ctype = ihdr[:4]
    if depth == 8:
        return np.asarray(Image.open(path))[:, :, ::-1]
    assert depth == 16 and ctype == 2
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * 6)
    px = png_unfilter(rows, 6).reshape(h, w, 3, 2).astype(np.uint16)
    return ((px[..., 0] << 8) | px[..., 1])[..., ::-1]


def check_two_streams(exe, tmp_path, env=None, more_args=(), streams=2):
    """`exe --num_frames 3 --num_streams 2` against two single invocations with the two segments (frames 7-8, frame 9): every
    equirect and the state files behind each stream's last frame, file for file; the first stream's frames against the
    reference program's chain. Used on the CPU emulation (two emulated devices) and on the GPU box (both streams on its GPU:
    there, and with --stream_gpus 1 anywhere, the streams are the frame slots of ONE context). streams=3: three streams of one
    frame each."""
    import rigutil
    rig = rigutil.scaled_rig_json(os.path.join(ROOT, "tests", "golden", "rig_17cam.json"), str(tmp_path / "rig_small.json"),
                                  CAM / 2048.0)
    name = "three_frames_sharpened"
    frames = CASES[name][0]
    both = run_stream(exe, str(tmp_path / "both"), rig, name, more_args=["--num_streams", str(streams), "--v", "1"] + list(more_args), env=env)
    segments = [(0, 2), (2, 1)] if streams == 2 else [(0, 1), (1, 1), (2, 1)]
    singles = {}
    for first, count in segments:
        out = run_stream(exe, str(tmp_path / ("seg%d" % first)), rig, name, first=first, count=count, env=env)
        for k in range(first, first + count):
            singles[frames[k]] = out

    def files(root, frame):
        d = {"eqr": _digest_png(os.path.join(root, "eqr_%s.png" % frame))}
        for folder in (os.path.join(root, "flow", frame), os.path.join(root, "debug", frame, "flow_images")):
            if os.path.isdir(folder):
                for fn in sorted(os.listdir(folder)):
                    p = os.path.join(folder, fn)
                    d[fn] = _digest_png(p) if fn.endswith(".png") else hashlib.sha256(open(p, "rb").read()).hexdigest()
        return d
    for f in frames:
        a, b = files(both, f), files(singles[f], f)
        assert a == b, "frame %s: %s" % (f, sorted(k for k in set(a) | set(b) if a.get(k) != b.get(k))[:8])
    for first, count in segments:  # the state behind each stream's last frame
        assert len(files(both, frames[first + count - 1])) > 30
    # and the first stream's frames are the reference program's chain (frame 9 of the chain has a predecessor, the segment's has not)
    golden = json.load(open(GOLDEN))[name]
    for f in frames[:segments[0][1]]:
        assert files(both, f)["eqr"] == golden["eqr_%s" % f], f
