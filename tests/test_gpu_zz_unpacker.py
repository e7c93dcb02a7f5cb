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
