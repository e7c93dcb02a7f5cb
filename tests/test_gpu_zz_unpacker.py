"""host/Unpacker end to end on the GPU: a synthetic two-camera capture container -> per-serial ISP configurations ->
16-bit PNGs, bit-exact against the oracle (RawConverter + the accelerated pipeline's restatement by default like the
reference's Unpacker, the pinned soft ISP with --soft_isp), raw TIFFs beside them, camN renaming.
Sorted last on purpose: written after round 2's GPU minutes were spent, its first hardware run is the round-end suite
(everything it calls — s360_isp_process_packed, the 16-bit PNG writer — is covered by earlier tests)."""
import json
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import refprog
from PIL import Image

import isputil

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _png16_bgr(path):
    data = open(path, "rb").read()
    pos, idat, ihdr = 8, b"", None
    while pos < len(data):
        n, t = struct.unpack(">I4s", data[pos:pos + 8])
        if t == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", data[pos + 8:pos + 8 + n])
        elif t == b"IDAT":
            idat += data[pos + 8:pos + 8 + n]
        pos += 12 + n
    w, h = ihdr[:2]
    assert ihdr[2:4] == (16, 2)  # 16-bit RGB, as imwriteExceptionOnFail of a CV_16UC3 Mat
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * 6)
    px = refprog.png_unfilter(rows, 6).reshape(h, w, 3, 2).astype(np.uint16)  # (the writer filters its scanlines)
    return ((px[..., 0] << 8) | px[..., 1])[..., ::-1]


@pytest.mark.parametrize("soft", [False, True], ids=["pipe", "soft_isp"])
@pytest.mark.parametrize("bits", [12, 8])
def test_unpacker_binary(tmp_path, oracle, s360lib, bits, soft):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])
    check_unpacker(os.path.join(ROOT, "host", "Unpacker"), tmp_path, oracle, bits, soft)


def check_unpacker(exe, tmp_path, oracle, bits, soft=True):
    """(also run by tests/test_cpu_library_emulation.py on the program linked against the emulated library)"""
    w, h, nf = 128, 96, 3
    serials = [17430921, 16241093]
    configs = [isputil.CONFIG_FULL, isputil.CONFIG_GRBG_NOSHARP]
    frames = [[isputil.bayer_frame(w, h, seed=10 * f + c + bits) for c in range(2)] for f in range(nf)]
    binp = tmp_path / "0.bin"
    written = isputil.footage_file(str(binp), frames, bits, serials)
    out, raw, ispd = tmp_path / "rgb", tmp_path / "raw", tmp_path / "isp"
    for d in (out, raw, ispd):
        d.mkdir()
    for s, js in zip(serials, configs):
        (ispd / ("%d.json" % s)).write_text(js)
    r = subprocess.run([exe, "--isp_dir", str(ispd), "--output_dir", str(out), "--output_raw_dir", str(raw),
                        "--bin_list", str(binp)] + (["--soft_isp"] if soft else []), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    order = sorted(range(2), key=lambda c: serials[c])  # cam0 = the smallest serial number
    assert sorted(os.listdir(out)) == ["cam0", "cam1"]
    for n, cam in enumerate(order):
        ocfg = oracle.isp_config_from_json(configs[cam], 16)
        assert sorted(os.listdir(out / ("cam%d" % n))) == ["%06d.png" % f for f in range(nf)]
        for f in range(nf):
            raw16 = oracle.isp_unpack_frame(written[f][cam], bits, w, h)
            got = _png16_bgr(str(out / ("cam%d" % n) / ("%06d.png" % f)))
            want = oracle.isp_run(ocfg, raw16) if soft else oracle.isp_pipe_run(ocfg, raw16)  # (pipe: CameraIspPipe, not pinned)
            assert np.array_equal(got, want), (bits, cam, f)
            tiff = np.array(Image.open(str(raw / str(serials[cam]) / ("%06d.tiff" % f))))
            assert np.array_equal(tiff, raw16)


def check_bin_list(unpacker_exe, trsp_exe, tmp_path, bits=12, soft=False, chain=True, two_devices_env=None):
    """host/TestRenderStereoPanorama --bin_list: the capture's containers -> ISP -> stereo frame on the device, against the
    chain through files (host/Unpacker writes 16-bit PNGs, the renderer reads them back): the same equirects, for two chained
    frames and for the two frames as one stream. (Also run by tests/test_cpu_library_emulation.py on the emulated programs.)"""
    import rigutil
    cam = refprog.CAM
    rig = rigutil.scaled_rig_json(os.path.join(ROOT, "tests", "golden", "rig_17cam.json"), str(tmp_path / "rig_small.json"), cam / 2048.0)
    ids = [c["id"] for c in json.load(open(rig))["cameras"]]
    n = len(ids)
    assert sorted(ids) == sorted("cam%d" % k for k in range(n))
    serials = [40000 + 7 * k for k in range(n)]  # ascending: Unpacker's cam<k> is the k-th smallest serial number
    configs = [isputil.CONFIG_GRBG_NOSHARP if k % 3 else isputil.CONFIG_FULL for k in range(n)]
    pats = ["GRBG" if k % 3 else "RGGB" for k in range(n)]
    frames = [[isputil.bayer_frame(cam, cam, seed=100 * f + k, pattern=pats[k]) for k in range(n)] for f in range(2)]
    binp = tmp_path / "0.bin"
    isputil.footage_file(str(binp), frames, bits, serials)
    ispd, imgs = tmp_path / "isp", tmp_path / "imgs"
    ispd.mkdir()
    imgs.mkdir()
    for s, js in zip(serials, configs):
        (ispd / ("%d.json" % s)).write_text(js)
    soft_flag = ["--soft_isp"] if soft else []
    r = subprocess.run([unpacker_exe, "--isp_dir", str(ispd), "--output_dir", str(imgs), "--bin_list", str(binp)] + soft_flag,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    common = ["--rig_json_file", rig, "--eqr_width", str(refprog.EQR_W), "--eqr_height", str(refprog.EQR_H), "--final_eqr_width",
              str(refprog.FINAL), "--final_eqr_height", str(refprog.FINAL), "--enable_top", "--enable_bottom", "--sharpening", "0.25"]

    def run(tag, src, frame, prev, extra=(), env=None):
        out = tmp_path / tag
        for d in (out, out / "flow", out / "debug", out / "flow" / frame, out / "debug" / frame, out / "debug" / frame / "flow_images"):
            d.mkdir(exist_ok=True)
        cmd = [trsp_exe] + common + src + ["--frame_number", frame, "--output_data_dir", str(out), "--prev_frame_data_dir", prev,
                                           "--output_equirect_path", str(out / "eqr_{frame}.png")] + list(extra)
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, "%s: rc %d\n%s" % (tag, r.returncode, r.stderr[-2000:])
        return out

    files = ["--imgs_dir", str(imgs)]
    bins = ["--bin_list", str(binp), "--isp_dir", str(ispd)] + soft_flag
    a = run("files", files, "000000", "NONE", ["--output_equirect_path", str(tmp_path / "files" / "eqr_000000.png")])
    b = run("bins", bins, "000000", "NONE", ["--output_equirect_path", str(tmp_path / "bins" / "eqr_000000.png")])
    want = refprog.png_pixels_bgr(str(a / "eqr_000000.png"))
    assert want.std() > 5
    assert np.array_equal(refprog.png_pixels_bgr(str(b / "eqr_000000.png")), want), "bins"
    if not chain:
        return
    run("files", files, "000001", "000000", ["--output_equirect_path", str(tmp_path / "files" / "eqr_000001.png")])
    run("bins", bins, "000001", "000000", ["--output_equirect_path", str(tmp_path / "bins" / "eqr_000001.png")])
    c = run("stream", bins, "000000", "NONE", ["--num_frames", "2"])
    for f in ("000000", "000001"):
        want = refprog.png_pixels_bgr(str(a / ("eqr_%s.png" % f)))
        assert np.array_equal(refprog.png_pixels_bgr(str(b / ("eqr_%s.png" % f))), want), ("bins", f)
        assert np.array_equal(refprog.png_pixels_bgr(str(c / ("eqr_%s.png" % f))), want), ("stream", f)
    # --num_streams 2 --stream_gpus 1: the two frames as two streams in the frame slots of ONE context, both fed from the containers
    # (frame 1 starts a stream of its own: no previous frame) — and, here, written as PNGs the device encoded
    alone = run("bins1", bins, "000001", "NONE", ["--output_equirect_path", str(tmp_path / "bins1" / "eqr_000001.png")])
    e = run("slots2", bins, "000000", "NONE", ["--num_frames", "2", "--num_streams", "2", "--stream_gpus", "1"])
    assert np.array_equal(refprog.png_pixels_bgr(str(e / "eqr_000000.png")), refprog.png_pixels_bgr(str(a / "eqr_000000.png")))
    assert np.array_equal(refprog.png_pixels_bgr(str(e / "eqr_000001.png")), refprog.png_pixels_bgr(str(alone / "eqr_000001.png")))
    if two_devices_env:  # --num_streams 2: frame 0 on one device, frame 1 as a stream of its own on the next (each opens the containers)
        d = run("streams2", bins, "000000", "NONE", ["--num_frames", "2", "--num_streams", "2"], env=two_devices_env)
        assert np.array_equal(refprog.png_pixels_bgr(str(d / "eqr_000000.png")), refprog.png_pixels_bgr(str(a / "eqr_000000.png")))
        assert np.array_equal(refprog.png_pixels_bgr(str(d / "eqr_000001.png")), refprog.png_pixels_bgr(str(alone / "eqr_000001.png")))


def test_renderer_fed_from_the_capture_containers(tmp_path, oracle, s360lib):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])
    check_bin_list(os.path.join(ROOT, "host", "Unpacker"), os.path.join(ROOT, "host", "TestRenderStereoPanorama"), tmp_path)
