"""ctypes bindings for the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg — never by the product package.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

ORACLE_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
_LIB = None


class CameraC(C.Structure):
    _fields_ = [
        ("type", C.c_int), ("is_side", C.c_int), ("has_fov", C.c_int), ("pad_", C.c_int),
        ("origin", C.c_double * 3), ("forward", C.c_double * 3), ("up", C.c_double * 3), ("right", C.c_double * 3),
        ("resolution", C.c_double * 2), ("principal", C.c_double * 2), ("distortion", C.c_double * 2),
        ("focal", C.c_double * 2), ("fov", C.c_double),
    ]


class ParamsC(C.Structure):
    _fields_ = [
        ("interpupilary_dist", C.c_double), ("zero_parallax_dist", C.c_double), ("sharpening", C.c_double),
        ("side_alpha_feather_size", C.c_int), ("std_alpha_feather_size", C.c_int),
        ("enable_top", C.c_int), ("enable_bottom", C.c_int),
        ("eqr_width", C.c_int), ("eqr_height", C.c_int), ("final_eqr_width", C.c_int), ("final_eqr_height", C.c_int),
        ("side_flow_search20", C.c_int), ("polar_flow_search20", C.c_int),
        ("enable_pole_removal", C.c_int), ("poleremoval_flow_search20", C.c_int),
    ]


def build(force=False):
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".h", ".cpp", "Makefile"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_frame_create.restype = C.c_void_p
        _LIB.orc_frame_render.restype = C.c_double
        _LIB.orc_camera_get_fov.restype = C.c_double
        _LIB.orc_camera_undistort_distort.restype = C.c_double
        _LIB.orc_approximate_fov.restype = C.c_float
        _LIB.orc_frame_usable_pixels_radius.restype = C.c_float
    return _LIB


def _p(a, t=None):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def camera_from_json(j):
    """Camera(const dynamic& json), Camera.cpp:44-83."""
    c = CameraC()
    c.type = 0 if j["type"] == "FTHETA" else 1
    c.is_side = 1 if "side" in j.get("group", "") else 0
    for k, n in (("origin", 3), ("forward", 3), ("up", 3), ("right", 3), ("resolution", 2), ("focal", 2)):
        for i in range(n):
            getattr(c, k)[i] = float(j[k][i])
    pr = j.get("principal", [j["resolution"][0] / 2, j["resolution"][1] / 2])
    di = j.get("distortion", [0, 0])
    for i in range(2):
        c.principal[i] = float(pr[i])
        c.distortion[i] = float(di[i])
    c.has_fov = 1 if "fov" in j else 0
    c.fov = float(j.get("fov", 0.0))
    return c


def load_rig(path):
    with open(path) as f:
        cams = json.load(f)["cameras"]
    arr = (CameraC * len(cams))(*[camera_from_json(c) for c in cams])
    return arr, [c["id"] for c in cams]


def make_params(**kw):
    p = ParamsC()
    p.interpupilary_dist = 6.4
    p.zero_parallax_dist = 10000.0
    p.sharpening = 0.0
    p.side_alpha_feather_size = 100
    p.std_alpha_feather_size = 31
    p.enable_top = 0
    p.enable_bottom = 0
    p.eqr_width = 256
    p.eqr_height = 128
    p.final_eqr_width = 3480
    p.final_eqr_height = 960
    names = {f[0] for f in ParamsC._fields_}
    for k, v in kw.items():
        if k not in names:  # (setattr on a ctypes Structure would silently make a Python attribute)
            raise AttributeError("no oracle parameter '%s'" % k)
        setattr(p, k, v)
    return p


# ---- primitives ---------------------------------------------------------------
def resize_cubic_u8(src, dw, dh):
    h, w, c = src.shape
    d = np.empty((dh, dw, c), np.uint8)
    lib().orc_resize_cubic_u8(_p(np.ascontiguousarray(src)), w, h, c, _p(d), dw, dh)
    return d


def _f3(a):
    a = np.ascontiguousarray(a, np.float32)
    return a[:, :, None] if a.ndim == 2 else a


def resize_cubic_f32(src, dw, dh):
    s = _f3(src)
    h, w, c = s.shape
    d = np.empty((dh, dw, c), np.float32)
    lib().orc_resize_cubic_f32(_p(s), w, h, c, _p(d), dw, dh)
    return d if src.ndim == 3 else d[:, :, 0]


def resize_linear_f32(src, dw, dh):
    s = _f3(src)
    h, w, c = s.shape
    d = np.empty((dh, dw, c), np.float32)
    lib().orc_resize_linear_f32(_p(s), w, h, c, _p(d), dw, dh)
    return d if src.ndim == 3 else d[:, :, 0]


def remap_cubic_u8(src, mp):
    src = np.ascontiguousarray(src)
    mp = np.ascontiguousarray(mp, np.float32)
    h, w, c = src.shape
    dh, dw, _ = mp.shape
    d = np.empty((dh, dw, c), np.uint8)
    lib().orc_remap_cubic_u8(_p(src), w, h, c, _p(mp), dw, dh, _p(d))
    return d


def remap_cubic_f32(src, mp):
    s = _f3(src)
    mp = np.ascontiguousarray(mp, np.float32)
    h, w, c = s.shape
    dh, dw, _ = mp.shape
    d = np.empty((dh, dw, c), np.float32)
    lib().orc_remap_cubic_f32(_p(s), w, h, c, _p(mp), dw, dh, _p(d))
    return d


def gaussian_blur_f32(src, ksize, sigma):
    s = _f3(src)
    h, w, c = s.shape
    d = np.empty_like(s)
    lib().orc_gaussian_blur_f32(_p(s), w, h, c, ksize, C.c_double(sigma), _p(d))
    return d if src.ndim == 3 else d[:, :, 0]


def gaussian_kernel(n, sigma):
    k = np.empty(n, np.float32)
    lib().orc_gaussian_kernel(n, C.c_double(sigma), _p(k))
    return k


def sobel(src, dir_y):
    s = np.ascontiguousarray(src, np.float32)
    d = np.empty_like(s)
    lib().orc_sobel(_p(s), s.shape[1], s.shape[0], int(dir_y), _p(d))
    return d


def median5(src):
    s = _f3(src)
    h, w, c = s.shape
    d = np.empty_like(s)
    lib().orc_median5_f32(_p(s), w, h, c, _p(d))
    return d if src.ndim == 3 else d[:, :, 0]


def bicubic_tab():
    tf = np.empty((1024, 16), np.float32)
    ti = np.empty((1024, 16), np.int16)
    lib().orc_bicubic_tab(_p(tf), _p(ti))
    return tf, ti


def feather_alpha_channel(src, erode_size):
    s = np.ascontiguousarray(src)
    d = np.empty_like(s)
    lib().orc_feather_alpha_channel(_p(s), s.shape[1], s.shape[0], erode_size, _p(d))
    return d


def offset_horizontal_wrap(src, offset):
    s = np.ascontiguousarray(src)
    d = np.empty_like(s)
    lib().orc_offset_horizontal_wrap(_p(s), s.shape[1], s.shape[0], s.shape[2], C.c_float(offset), _p(d))
    return d


def flatten_layers(base, top):
    b = np.ascontiguousarray(base)
    t = np.ascontiguousarray(top)
    d = np.empty_like(b)
    lib().orc_flatten_layers_deghost_prefer_base(_p(b), _p(t), b.shape[1], b.shape[0], _p(d))
    return d


def sharpen(bgr, amount):
    b = np.ascontiguousarray(bgr).copy()
    lib().orc_sharpen(_p(b), b.shape[1], b.shape[0], C.c_float(amount))
    return b


# ---- PixFlow --------------------------------------------------------------------
HINT = {"UNKNOWN": 0, "RIGHT": 1, "DOWN": 2, "LEFT": 3, "UP": 4}


def pixflow_levels(w, h):
    lw = (C.c_int * 64)()
    lh = (C.c_int * 64)()
    n = lib().orc_pixflow_levels(w, h, lw, lh)
    return [(lw[i], lh[i]) for i in range(n)]


def compute_optical_flow(i0, i1, alg="pixflow_low", hint="UNKNOWN", prev_flow=None, prev_i0=None, prev_i1=None,
                         want_levels=False):
    i0 = np.ascontiguousarray(i0)
    i1 = np.ascontiguousarray(i1)
    h, w, _ = i0.shape
    flow = np.empty((h, w, 2), np.float32)
    lv = None
    if want_levels:
        sizes = pixflow_levels(w, h)
        lv = np.empty(sum(a * b * 2 for a, b in sizes), np.float32)
    pf = np.ascontiguousarray(prev_flow, np.float32) if prev_flow is not None else None
    p0 = np.ascontiguousarray(prev_i0) if prev_i0 is not None else None
    p1 = np.ascontiguousarray(prev_i1) if prev_i1 is not None else None
    rc = lib().orc_compute_optical_flow(alg.encode(), _p(i0), _p(i1), w, h, _p(pf), _p(p0), _p(p1), HINT[hint],
                                        _p(flow), _p(lv))
    if rc != 0:
        raise ValueError("unrecognized flow algorithm name: " + alg)
    if want_levels:
        out, off = [], 0
        for (a, b) in reversed(sizes):
            out.append(lv[off:off + a * b * 2].reshape(b, a, 2))
            off += a * b * 2
        return flow, out
    return flow


def pixflow_entry(i0):
    i0 = np.ascontiguousarray(i0)
    h, w, _ = i0.shape
    dw, dh = int(w * 0.5), int(h * 0.5)
    down = np.empty((dh, dw, 4), np.uint8)
    I = np.empty((dh, dw), np.float32)
    A = np.empty((dh, dw), np.float32)
    lib().orc_pixflow_entry(_p(i0), w, h, _p(down), _p(I), _p(A))
    return down, I, A


def pixflow_level(I0, I1, a0, a1, flow=None, hint="UNKNOWN", search20=False):
    h, w = I0.shape
    f = np.zeros((h, w, 2), np.float32) if flow is None else np.ascontiguousarray(flow, np.float32).copy()
    lib().orc_pixflow_level(_p(np.ascontiguousarray(I0, np.float32)), _p(np.ascontiguousarray(I1, np.float32)),
                            _p(np.ascontiguousarray(a0, np.float32)), _p(np.ascontiguousarray(a1, np.float32)), w, h,
                            _p(f), 0 if flow is None else 1, HINT[hint], int(search20))
    return f


# ---- geometry ---------------------------------------------------------------------
def spherical_warp_map(cam, dw, dh, l, r, t, b):
    m = np.empty((dh, dw, 2), np.float32)
    lib().orc_spherical_warp_map(C.byref(cam), dw, dh, C.c_float(l), C.c_float(r), C.c_float(t), C.c_float(b), _p(m))
    return m


def bicubic_remap_to_spherical(cam, src, dw, dh, dc, l, r, t, b):
    src = np.ascontiguousarray(src)
    d = np.empty((dh, dw, dc), np.uint8)
    lib().orc_bicubic_remap_to_spherical(C.byref(cam), _p(src), src.shape[1], src.shape[0], src.shape[2], dw, dh, dc,
                                         C.c_float(l), C.c_float(r), C.c_float(t), C.c_float(b), _p(d))
    return d


class Frame:
    """renderStereoPanorama (TestRenderStereoPanorama.cpp:716-972) on the oracle."""

    def __init__(self, cams, params):
        self.cams, self.params = cams, params
        self.h = C.c_void_p(lib().orc_frame_create(cams, len(cams), C.byref(params)))
        ints = (C.c_int * 6)()
        fl = (C.c_float * 5)()
        lib().orc_frame_geometry(self.h, ints, fl)
        (self.cam_image_width, self.cam_image_height, self.overlap_image_width, self.num_novel_views,
         self.top_rows, self.bottom_rows) = list(ints)
        (self.h_radians, self.v_radians, self.fov_horizontal_radians, self.verge_disp,
         self.zero_parallax_shift) = list(fl)
        self.n_side = sum(1 for c in cams if c.is_side)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_frame_destroy(self.h)
            self.h = None

    def pole_ramp(self):
        o = (C.c_float * 4)()
        lib().orc_frame_pole_ramp(self.h, o)
        return list(o)

    def render(self, side, top=None, bottom=None, use_prev=False, threaded=False):
        side = [np.ascontiguousarray(s) for s in side]
        h, w, ch = side[0].shape
        ptrs = (C.c_void_p * len(side))(*[s.ctypes.data for s in side])
        top = np.ascontiguousarray(top) if top is not None else None
        bottom = np.ascontiguousarray(bottom) if bottom is not None else None
        pole = top if top is not None else bottom
        ph, pw = (pole.shape[0], pole.shape[1]) if pole is not None else (0, 0)
        sec = lib().orc_frame_render(self.h, ptrs, w, h, ch, _p(top), _p(bottom), pw, ph, int(use_prev), int(threaded))
        return self.get_u8("out"), sec

    def stage_seconds(self):
        o = (C.c_double * 5)()
        lib().orc_frame_stage_seconds(self.h, o)
        return dict(zip(["projection", "side_flow", "novel_view", "poles", "total"], list(o)))

    def get_u8(self, name, idx=0):
        whc = (C.c_int * 3)()
        if lib().orc_frame_get_u8(self.h, name.encode(), idx, whc, None) != 0:
            raise KeyError(name)
        d = np.empty((whc[1], whc[0], whc[2]), np.uint8)
        lib().orc_frame_get_u8(self.h, name.encode(), idx, whc, _p(d))
        return d

    def set_pole_removal(self, bottom2, mask, mask2):
        """Secondary bottom camera image + red pole masks (BGR) for enable_pole_removal."""
        b2, m1, m2 = (np.ascontiguousarray(a, np.uint8) for a in (bottom2, mask, mask2))
        assert b2.shape == m1.shape == m2.shape and b2.shape[2] == 3
        lib().orc_frame_set_pole_removal(self.h, _p(b2), _p(m1), _p(m2), b2.shape[1], b2.shape[0])

    def bottom2_index(self):
        return lib().orc_frame_bottom2_index(self.h)

    def usable_pixels_radius(self, cam_idx):
        return float(lib().orc_frame_usable_pixels_radius(self.h, cam_idx))

    def cubemap(self, face_w, face_h, fmt="video"):
        whc = (C.c_int * 3)()
        if lib().orc_frame_cubemap(self.h, face_w, face_h, fmt.encode(), whc, None) != 0:
            raise ValueError("no frame rendered or unexpected cubemap format")
        d = np.empty((whc[1], whc[0], 3), np.uint8)
        lib().orc_frame_cubemap(self.h, face_w, face_h, fmt.encode(), whc, _p(d))
        return d

    def get_f32(self, name, idx=0):
        whc = (C.c_int * 3)()
        if lib().orc_frame_get_f32(self.h, name.encode(), idx, whc, None) != 0:
            raise KeyError(name)
        d = np.empty((whc[1], whc[0], whc[2]), np.float32)
        lib().orc_frame_get_f32(self.h, name.encode(), idx, whc, _p(d))
        return d

    def combine_lazy_novel_views(self, img_l, img_r, flow_l_to_r, flow_r_to_l):
        w = self.params.eqr_width // self.n_side
        cl = np.empty((self.cam_image_height, w, 4), np.uint8)
        cr = np.empty_like(cl)
        lib().orc_frame_combine_lazy_novel_views(
            self.h, _p(np.ascontiguousarray(img_l)), _p(np.ascontiguousarray(img_r)),
            _p(np.ascontiguousarray(flow_l_to_r, np.float32)), _p(np.ascontiguousarray(flow_r_to_l, np.float32)),
            _p(cl), _p(cr))
        return cl, cr

    def pole_to_side_flow(self, side, pole, want_flow=False):
        side = np.ascontiguousarray(side)
        pole = np.ascontiguousarray(pole)
        sh, sw, _ = side.shape
        ph, pw, _ = pole.shape
        out = np.empty_like(side)
        fl = np.empty((ph, int(np.float32(pw) * np.float32(1.2)), 2), np.float32) if want_flow else None
        lib().orc_frame_pole_to_side_flow(self.h, _p(side), sw, sh, _p(pole), pw, ph, _p(out), _p(fl))
        return (out, fl) if want_flow else out


# ---- soft ISP (oracle/isp.h; reference CameraIsp.h through Raw2Rgb's non-accelerated path) --------------------------
ISP_MAX_CURVE_POINTS = 16
BAYER_PATTERNS = ("RGGB", "GRBG", "GBRG", "BGGR")


class IspConfigC(C.Structure):
    """orc::IspConfig (oracle/isp.h) field by field."""
    _fields_ = [
        ("blackLevel", C.c_float * 3), ("clampMin", C.c_float * 3), ("clampMax", C.c_float * 3),
        ("whiteBalanceGain", C.c_float * 3), ("ccm", C.c_float * 9), ("saturation", C.c_float), ("contrast", C.c_float),
        ("gamma", C.c_float * 3), ("lowKeyBoost", C.c_float * 3), ("highKeyBoost", C.c_float * 3),
        ("sharpening", C.c_float * 3), ("sharpeningSupport", C.c_float), ("noiseCore", C.c_float),
        ("nVignetteH", C.c_int), ("nVignetteV", C.c_int),
        ("vignetteRollOffH", (C.c_float * 3) * ISP_MAX_CURVE_POINTS),
        ("vignetteRollOffV", (C.c_float * 3) * ISP_MAX_CURVE_POINTS),
        ("stuckPixelRadius", C.c_int), ("bayerPattern", C.c_int),
        ("outputBpp", C.c_int), ("demosaicFilter", C.c_int), ("resize", C.c_int), ("disableToneCurve", C.c_int),
        ("blackLevelOffset", C.c_int),
        ("stuckPixelThreshold", C.c_int), ("stuckPixelDarknessThreshold", C.c_float),
    ]


def isp_config_from_json(json_text, output_bpp=8, demosaic_filter=2, resize=1, disable_tone_curve=0,
                         black_level_offset=0):
    """The CameraIsp constructor's reading of the "CameraIsp" object (CameraIsp.h:425-607): defaults, then the keys
    present; JSON doubles are narrowed to float exactly as `v.x = vec[0].ToDouble()` does."""
    c = IspConfigC()
    for k in range(3):
        c.clampMax[k] = c.whiteBalanceGain[k] = c.gamma[k] = 1.0
    for k in (0, 4, 8):
        c.ccm[k] = 1.0
    c.saturation = c.contrast = 1.0
    c.sharpeningSupport = np.float32(10.0) / np.float32(2048.0)
    c.noiseCore = 1000.0
    c.nVignetteH = c.nVignetteV = 1
    for k in range(3):
        c.vignetteRollOffH[0][k] = c.vignetteRollOffV[0][k] = 1.0
    c.bayerPattern = 2  # "GBRG"
    j = json.loads(json_text).get("CameraIsp", {})

    def vec(name, dst):
        if name in j:
            for k in range(3):
                dst[k] = j[name][k]
    vec("blackLevel", c.blackLevel); vec("clampMin", c.clampMin); vec("clampMax", c.clampMax)
    vec("whiteBalanceGain", c.whiteBalanceGain); vec("gamma", c.gamma); vec("lowKeyBoost", c.lowKeyBoost)
    vec("highKeyBoost", c.highKeyBoost); vec("sharpening", c.sharpening)
    if "ccm" in j:
        for a in range(3):
            for b in range(3):
                c.ccm[a * 3 + b] = j["ccm"][a][b]
    for name in ("saturation", "contrast", "sharpeningSupport", "noiseCore"):
        if name in j:
            setattr(c, name, j[name])
    for name, dst, cnt in (("vignetteRollOffH", c.vignetteRollOffH, "nVignetteH"),
                           ("vignetteRollOffV", c.vignetteRollOffV, "nVignetteV")):
        if name in j:
            pts = j[name]
            assert 1 <= len(pts) <= ISP_MAX_CURVE_POINTS
            setattr(c, cnt, len(pts))
            for i, p in enumerate(pts):
                for k in range(3):
                    dst[i][k] = p[k]
    if "stuckPixelRadius" in j:
        c.stuckPixelRadius = 2 * int(j["stuckPixelRadius"])
    c.stuckPixelThreshold = int(j.get("stuckPixelThreshold", 0))
    c.stuckPixelDarknessThreshold = float(j.get("stuckPixelDarknessThreshold", 0.0))
    if "bayerPattern" in j:  # setup() uses find(): the first pattern name contained in the string, in this order
        c.bayerPattern = next(i for i, n in enumerate(BAYER_PATTERNS) if n in j["bayerPattern"])
    c.outputBpp, c.demosaicFilter, c.resize = output_bpp, demosaic_filter, resize
    c.disableToneCurve, c.blackLevelOffset = disable_tone_curve, black_level_offset
    return c


def _isp_out(raw, cfg):
    h, w = raw.shape
    return np.zeros((h // cfg.resize, w // cfg.resize, 3), np.uint8 if cfg.outputBpp == 8 else np.uint16)


def isp_run(cfg, raw):
    """oracle restatement: raw (H x W uint16 Bayer) -> (H/resize) x (W/resize) x 3 BGR, uint8 / uint16."""
    raw = np.ascontiguousarray(raw, np.uint16)
    assert lib().orc_isp_config_size() == C.sizeof(IspConfigC)
    out = _isp_out(raw, cfg)
    err = C.create_string_buffer(256)
    if lib().orc_isp_run(C.byref(cfg), _p(raw), raw.shape[1], raw.shape[0], _p(out), err, 256) != 0:
        raise RuntimeError(err.value.decode())
    return out


def isp_pipe_run(cfg, raw, fast=False):
    """oracle restatement of the accelerated ISP (oracle/isp_pipe.h: CameraIspGen.cpp; pinned to the generator executed — ref_isp_pipe_run below): raw -> H x W x 3 BGR.
    cfg.resize / cfg.demosaicFilter play no part."""
    raw = np.ascontiguousarray(raw, np.uint16)
    assert lib().orc_isp_config_size() == C.sizeof(IspConfigC)
    out = np.zeros(raw.shape + (3,), np.uint8 if cfg.outputBpp == 8 else np.uint16)
    err = C.create_string_buffer(256)
    if lib().orc_isp_pipe_run(C.byref(cfg), int(bool(fast)), _p(raw), raw.shape[1], raw.shape[0], _p(out), err, 256) != 0:
        raise RuntimeError(err.value.decode())
    return out


def isp_tables(cfg):
    ccm, lut = np.zeros(9, np.float32), np.zeros((4096, 3), np.float32)
    lib().orc_isp_tables(C.byref(cfg), _p(ccm), _p(lut))
    return ccm.reshape(3, 3), lut


def ref_isp_lib():
    """oracle/_ref/libref_isp.so — the reference's own CameraIsp.h compiled over the container stand-in (see ref_lib)."""
    return ref_lib("isp")


def ref_isp_run(json_text, raw, output_bpp=8, demosaic_filter=2, resize=1, disable_tone_curve=0, black_level_offset=0):
    raw = np.ascontiguousarray(raw, np.uint16)
    h, w = raw.shape
    out = np.zeros((h // resize, w // resize, 3), np.uint8 if output_bpp == 8 else np.uint16)
    err = C.create_string_buffer(256)
    if ref_isp_lib().ref_isp_run(json_text.encode(), _p(raw), w, h, output_bpp, demosaic_filter, resize,
                                 disable_tone_curve, black_level_offset, _p(out), err, 256) != 0:
        raise RuntimeError(err.value.decode())
    return out


def ref_isp_pipe_lib():
    """oracle/_ref/libref_isppipe.so — the reference's Halide generator camera_isp/CameraIspGen.cpp EXECUTED over
    oracle/ref_shim/halide_eval (a lazy evaluator of the Halide front end), under its CameraIspPipe.h (oracle/ref_ispgen.cpp,
    ref_isppipe.cpp)."""
    return ref_lib("isppipe")


def ref_isp_pipe_run(json_text, raw, output_bpp=8, fast=False, disable_tone_curve=0, black_level_offset=0, unpacker=False,
                     bits_per_pixel=16):
    """The accelerated ISP as the reference's programs run it: Raw2Rgb --accelerate [--fast] (Raw2Rgb.cpp:427-440), or with
    unpacker=True Unpacker's call sequence (Unpacker.cpp:176-183: 16-bit output, full pipeline, tone map on). raw -> H x W x 3 BGR."""
    raw = np.ascontiguousarray(raw, np.uint16)
    h, w = raw.shape
    if unpacker:
        output_bpp, fast = 16, False
    out = np.zeros((h, w, 3), np.uint8 if output_bpp == 8 else np.uint16)
    err = C.create_string_buffer(512)
    if ref_isp_pipe_lib().ref_isp_pipe_run(int(bool(unpacker)), json_text.encode(), _p(raw), w, h, output_bpp, int(bool(fast)),
                                           disable_tone_curve, black_level_offset, bits_per_pixel, _p(out), err, 512) != 0:
        raise RuntimeError(err.value.decode())
    return out


def isp_packed_bytes(bits, w, h):
    return w * h if bits == 8 else h * (3 * w // 2)


def isp_unpack_frame(frame, bits, w, h):
    """oracle: RawConverter::convert8Frame / convert12Frame."""
    fr = np.ascontiguousarray(frame, np.uint8)
    out = np.zeros((h, w), np.uint16)
    lib().orc_isp_unpack_frame(bits, _p(fr), w, h, _p(out))
    return out


def ref_unpack_frame(frame, bits, w, h):
    fr = np.ascontiguousarray(frame, np.uint8)
    out = np.zeros((h, w), np.uint16)
    ref_isp_lib().ref_convert_frame(bits, _p(fr), w, h, _p(out))
    return out


# ---- the reference's own sources, compiled over oracle/ref_shim (oracle/_ref; built where /root/reference exists) ------
_REF_LIBS = {}


def ref_lib(name):
    """oracle/_ref/libref_<name>.so ("isp", "isppipe", "pixflow", "render") or None."""
    so = os.path.join(ORACLE_DIR, "_ref", "libref_%s.so" % name)
    if name not in _REF_LIBS:
        if os.path.isdir("/root/reference/surround360_render/source/optical_flow"):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "ref"])
        _REF_LIBS[name] = C.CDLL(so) if os.path.exists(so) else None
    return _REF_LIBS[name]


def _ref_call(fn, *args):
    err = C.create_string_buffer(256)
    if fn(*args, err, 256) != 0:
        raise RuntimeError(err.value.decode())


def ref_compute_optical_flow(i0, i1, alg="pixflow_low", hint="UNKNOWN", prev_flow=None, prev_i0=None, prev_i1=None):
    """The reference's PixFlow.h (makeOpticalFlowByName(alg)->computeOpticalFlow) over the oracle's OpenCV primitives."""
    i0, i1 = np.ascontiguousarray(i0), np.ascontiguousarray(i1)
    h, w, _ = i0.shape
    flow = np.zeros((h, w, 2), np.float32)
    pf = np.ascontiguousarray(prev_flow, np.float32) if prev_flow is not None else None
    p0 = np.ascontiguousarray(prev_i0) if prev_i0 is not None else None
    p1 = np.ascontiguousarray(prev_i1) if prev_i1 is not None else None
    _ref_call(ref_lib("pixflow").ref_pixflow, alg.encode(), _p(i0), _p(i1), w, h, _p(pf), _p(p0), _p(p1), HINT[hint], _p(flow))
    return flow


def ref_combine_lazy_novel_views(img_l, img_r, flow_l_to_r, flow_r_to_l, chunk_w, num_novel_views, cam_image_width,
                                 verge_disp):
    img_l, img_r = np.ascontiguousarray(img_l), np.ascontiguousarray(img_r)
    fl, fr = np.ascontiguousarray(flow_l_to_r, np.float32), np.ascontiguousarray(flow_r_to_l, np.float32)
    h, w, _ = img_l.shape
    cl, cr = np.zeros((h, chunk_w, 4), np.uint8), np.zeros((h, chunk_w, 4), np.uint8)
    _ref_call(ref_lib("render").ref_combine_lazy_novel_views, _p(img_l), _p(img_r), _p(fl), _p(fr), w, h, chunk_w,
              num_novel_views, cam_image_width, C.c_float(verge_disp), _p(cl), _p(cr))
    return cl, cr


def ref_flatten_layers(base, top):
    base, top = np.ascontiguousarray(base), np.ascontiguousarray(top)
    out = np.zeros_like(base)
    _ref_call(ref_lib("render").ref_flatten_layers, _p(base), _p(top), base.shape[1], base.shape[0], _p(out))
    return out


def ref_feather_alpha_channel(src, erode_size):
    src = np.ascontiguousarray(src)
    out = np.zeros_like(src)
    _ref_call(ref_lib("render").ref_feather_alpha_channel, _p(src), src.shape[1], src.shape[0], erode_size, _p(out))
    return out


def ref_offset_horizontal_wrap(src, offset):
    src = np.ascontiguousarray(src)
    out = np.zeros_like(src)
    _ref_call(ref_lib("render").ref_offset_horizontal_wrap, _p(src), src.shape[1], src.shape[0], src.shape[2],
              C.c_float(offset), _p(out))
    return out
