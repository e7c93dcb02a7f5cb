"""Synthetic Bayer frames and ISP configurations for the soft-ISP tests (test plumbing).

The configurations are this repo's own (they exercise every key the reference's CameraIsp constructor reads,
CameraIsp.h:425-607); the reference's shipped res/config/isp/*.json are only read in place when /root/reference exists."""
import json

import numpy as np

CONFIG_FULL = json.dumps({"CameraIsp": {
    "serial": 7, "name": "synthetic", "bitsPerPixel": 12,
    "compandingLut": [[0.0, 0.0, 0.0], [0.5, 0.6, 0.0], [1.0, 1.0, 0.0]],
    "blackLevel": [1210.0, 1302.5, 1188.0],
    "clampMin": [0.01, 0.0, 0.02], "clampMax": [0.97, 0.99, 0.95],
    "vignetteRollOffH": [[1.35, 1.3, 1.4], [1.12, 1.1, 1.15], [1.0, 1.0, 1.0], [1.08, 1.1, 1.11], [1.3, 1.27, 1.33]],
    "vignetteRollOffV": [[1.2, 1.25, 1.22], [1.05, 1.04, 1.06], [1.0, 1.0, 1.0], [1.06, 1.05, 1.07], [1.25, 1.2, 1.3],
                         [1.3, 1.31, 1.29]],
    "whiteBalanceGain": [1.37, 1.0, 1.81],
    "stuckPixelThreshold": 5, "stuckPixelDarknessThreshold": 0.11, "stuckPixelRadius": 0,
    "denoise": 0.8, "denoiseRadius": 4,
    "ccm": [[1.11, -0.07, 0.02], [0.13, 1.21, -0.28], [-0.12, -0.09, 1.3]],
    "sharpening": [0.5, 0.45, 0.6], "sharpeningSupport": 0.006, "noiseCore": 850.0,
    "saturation": 1.25, "contrast": 1.1,
    "lowKeyBoost": [-0.2, -0.15, -0.1], "highKeyBoost": [0.2, 0.1, 0.15],
    "gamma": [0.4545, 0.5, 0.42],
    "bayerPattern": "RGGB"}})
CONFIG_MINIMAL = json.dumps({"CameraIsp": {"bayerPattern": "BGGR", "whiteBalanceGain": [1.2, 1.0, 1.4]}})
CONFIG_EMPTY = json.dumps({"NotAnIsp": {}})  # every default (GBRG)
CONFIG_GRBG_NOSHARP = json.dumps({"CameraIsp": {
    "bayerPattern": "GRBG", "blackLevel": [600.0, 600.0, 600.0], "gamma": [0.45, 0.45, 0.45], "saturation": 0.9,
    "ccm": [[1.3, -0.2, -0.1], [-0.15, 1.4, -0.25], [0.02, -0.3, 1.28]],
    "vignetteRollOffH": [[1.2, 1.2, 1.2], [1.0, 1.0, 1.0], [1.2, 1.2, 1.2]]}})
CONFIGS = {"full": CONFIG_FULL, "minimal": CONFIG_MINIMAL, "empty": CONFIG_EMPTY, "grbg": CONFIG_GRBG_NOSHARP}


def stuck_pixel_config(radius, threshold, darkness, base=CONFIG_FULL):
    """CONFIG_FULL with removeStuckPixels switched on (CameraIsp.h:1024-1104): stuckPixelRadius > 0."""
    j = json.loads(base)
    j["CameraIsp"].update(stuckPixelRadius=radius, stuckPixelThreshold=threshold, stuckPixelDarknessThreshold=darkness)
    return json.dumps(j)


def bayer_frame(w, h, seed=0, pattern="GBRG", bits=16):
    """A smooth colour scene with edges and noise, mosaiced: H x W uint16 (full 16-bit range used, like the 12-bit
    sensor data Unpacker scales up)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    r = 0.35 + 0.25 * np.sin(xx / 13.0 + 0.3) * np.cos(yy / 17.0)
    g = 0.40 + 0.30 * np.cos(xx / 19.0) * np.sin(yy / 11.0 + 0.7)
    b = 0.30 + 0.20 * np.sin((xx + yy) / 23.0)
    box = ((xx > w * 0.3) & (xx < w * 0.55) & (yy > h * 0.25) & (yy < h * 0.7))
    r[box] += 0.3; g[box] -= 0.2; b[box] += 0.25
    disc = (xx - w * 0.75) ** 2 + (yy - h * 0.4) ** 2 < (0.12 * min(w, h)) ** 2
    r[disc] = 0.95; g[disc] = 0.93; b[disc] = 0.97  # near saturation
    rgb = np.stack([r, g, b], -1) + 0.01 * rng.normal(size=(h, w, 3))
    idx = {"R": 0, "G": 1, "B": 2}
    raw = np.zeros((h, w))
    for i in range(2):
        for j in range(2):
            raw[i::2, j::2] = rgb[i::2, j::2, idx[pattern[i * 2 + j]]]
    return np.clip(raw * 65535.0 + 0.06 * 65535, 0, 65535).astype(np.uint16)


def pack_frame(raw16, bits):
    """The sensor-side packing RawConverter undoes: raw16 (H x W uint16) -> bytes of an 8- or 12-bit frame (even W)."""
    h, w = raw16.shape
    if bits == 8:
        return (raw16 >> 8).astype(np.uint8).ravel()
    v = (raw16 >> 4).astype(np.uint32)  # 12-bit samples
    a, b = v[:, 0::2], v[:, 1::2]
    out = np.zeros((h, w // 2, 3), np.uint8)
    out[..., 0] = a >> 4
    out[..., 1] = (a & 0xF) | ((b & 0xF) << 4)
    out[..., 2] = b >> 4
    return out.ravel()


def footage_file(path, frames, bits, serials, timestamp=1234, file_index=0, file_count=1):
    """A capture container as BinaryFootageFile.cpp reads it: a 4096-byte metadata page, then the packed frames
    interleaved by camera (frames[f][cam] = H x W uint16). The camera stamps its serial number over bytes 4..7 of every
    frame; returns the frame bytes as written (frames_bytes[f][cam]) so a test can unpack exactly those."""
    h, w = frames[0][0].shape
    ncam = len(frames[0])
    page = np.zeros(4096, np.uint8)
    page[:32] = np.array([0xfaceb00c, timestamp, file_index, file_count, w, h, bits, ncam], np.uint32).view(np.uint8)
    written = []
    with open(path, "wb") as f:
        f.write(page.tobytes())
        for per_cam in frames:
            row = []
            for cam, img in enumerate(per_cam):
                fr = pack_frame(img, bits).copy()
                fr[4:8] = np.array([serials[cam]], np.uint32).view(np.uint8)
                f.write(fr.tobytes())
                row.append(fr)
            written.append(row)
    return written
