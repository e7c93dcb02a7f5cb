"""Host-side mirror of the reference's ISP interface over the C ABI of libs360 (include/s360.h, s360_isp_*):

  CameraIsp(json, output_bpp) + the Raw2Rgb flags     SR/camera_isp/CameraIsp.h:425-607, Raw2Rgb.cpp:25-39, 441-456
  load_image + get_image                              CameraIsp.h:831-854, 1275-1299
  CameraIspPipe(json, fast, output_bpp)               SR/camera_isp/CameraIspPipe.h (pipe=PIPE / PIPE_FAST: the Halide
                                                      pipeline's arithmetic restated from CameraIspGen.cpp, not pinned)

Everything computes on the GPU; numpy arrays stand in for cv::Mat (H x W uint16 raw, H x W x 3 BGR out)."""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import IspConfig, check, lib

BILINEAR_DM_FILTER, FREQUENCY_DM_FILTER, EDGE_AWARE_DM_FILTER = 0, 1, 2
SOFT, PIPE, PIPE_FAST = 0, 1, 2  # s360_isp_config.pipe: CameraIsp / CameraIspPipe / CameraIspPipe(fast = true)


def config_from_json(json_text, output_bpp=8, demosaic_filter=EDGE_AWARE_DM_FILTER, resize=1, disable_tone_curve=False,
                     black_level_offset=0, pipe=SOFT):
    """The CameraIsp constructor's reading of an ISP configuration text, plus the Raw2Rgb flags (pipe: --accelerate / --fast)."""
    c = IspConfig()
    lib().s360_isp_config_defaults(C.byref(c))
    c.output_bpp, c.demosaic_filter, c.resize, c.pipe = output_bpp, demosaic_filter, resize, pipe
    c.disable_tone_curve, c.black_level_offset = int(bool(disable_tone_curve)), black_level_offset
    check(lib().s360_isp_config_from_json(json_text.encode(), C.byref(c)))
    return c


def config_tables(cfg, w=0, h=0):
    """Host-side tables of a configuration: composite CCM x 4095 (3x3), tone curve (4096x3) and, for a w x h frame, the
    vignette gains per column / row (w x 3, h x 3). No device needed."""
    ccm, lut = np.zeros(9, np.float32), np.zeros((4096, 3), np.float32)
    ch = np.zeros((max(w, 1), 3), np.float32)
    cv = np.zeros((max(h, 1), 3), np.float32)
    check(lib().s360_isp_config_tables(C.byref(cfg), ccm.ctypes.data_as(C.c_void_p), lut.ctypes.data_as(C.c_void_p), w, h,
                                       ch.ctypes.data_as(C.c_void_p) if w and h else None,
                                       cv.ctypes.data_as(C.c_void_p) if w and h else None))
    return ccm.reshape(3, 3), lut, ch, cv


class CameraIsp:
    """One configuration on one GPU; frames of any size can follow."""

    def __init__(self, config, device=0):
        self.config = config
        h = C.c_void_p()
        check(lib().s360_isp_create(C.byref(h), device, C.byref(config)))
        self.h = h

    def get_image(self, raw16):
        """raw16: H x W uint16 Bayer -> (H / resize) x (W / resize) x 3 BGR, uint8 or uint16 by output_bpp."""
        raw = np.ascontiguousarray(raw16, np.uint16)
        hh, ww = raw.shape
        r = self.config.resize
        out = np.empty((hh // r, ww // r, 3), np.uint8 if self.config.output_bpp == 8 else np.uint16)
        check(lib().s360_isp_process(self.h, raw.ctypes.data_as(C.c_void_p), ww, hh, out.ctypes.data_as(C.c_void_p)))
        return out

    def get_image_packed(self, frame, bits, w, h):
        """frame: the sensor's packed bytes of one w x h image (8 or 12 bits per pixel, as in a .bin container)."""
        fr = np.ascontiguousarray(frame, np.uint8)
        r = self.config.resize
        out = np.empty((h // r, w // r, 3), np.uint8 if self.config.output_bpp == 8 else np.uint16)
        check(lib().s360_isp_process_packed(self.h, fr.ctypes.data_as(C.c_void_p), bits, w, h,
                                            out.ctypes.data_as(C.c_void_p)))
        return out

    def close(self):
        if self.h:
            lib().s360_isp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass
