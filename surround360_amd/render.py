"""Host-side mirror of the reference's operator interface for the stereo-panorama path, over the
C ABI of libs360 (include/s360.h). Names follow the reference:

  RigDescription                      SR/render/RigDescription.h:27-59
  make_optical_flow_by_name           SR/optical_flow/OpticalFlowFactory.h:23-64
  OpticalFlow.compute_optical_flow    SR/optical_flow/OpticalFlowInterface.h:34-41
  NovelViewGeneratorAsymmetricFlow    SR/optical_flow/NovelView.h:160-190
  bicubic_remap_to_spherical          SR/render/ImageWarper.h:59-66
  flatten_layers_deghost_prefer_base, offset_horizontal_wrap, feather_alpha_channel   SR/util/CvUtil.h
  StereoPanoramaRenderer.render       renderStereoPanorama, SR/test/TestRenderStereoPanorama.cpp:716-972

Everything computes on the GPU through libs360; numpy arrays are the cv::Mat stand-ins
(H x W x C uint8, H x W x 2 float32 for flow).
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import HINT, Camera, Geometry, Params, S360Error, check, lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _u8(a):
    return np.ascontiguousarray(a, np.uint8)


def pinned_empty(shape, dtype=np.uint8):
    """A numpy array in page-locked host memory (s360_host_alloc): images uploaded from such an array are sent in place (and
    must stay untouched until Context.uploads_complete() has returned), a download into one is a single DMA transfer.
    Freed with the array."""
    import weakref
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = lib().s360_host_alloc(max(n, 1))
    if not p:
        raise MemoryError("s360_host_alloc(%d) failed: %s" % (n, lib().s360_last_error(None).decode()))
    raw = (C.c_uint8 * max(n, 1)).from_address(p)
    arr = np.frombuffer(raw, np.uint8, n).view(dtype).reshape(shape)
    weakref.finalize(raw, lib().s360_host_free, p)
    return arr


class VrCamException(S360Error):
    """The reference's exception type for bad arguments / unknown algorithms (VrCamException.h:18-23)."""


class RigDescription:
    """RigDescription(filename): rig JSON -> cameras; rig_side_only = cameras whose group contains 'side'."""

    MAX_CAMS = 64

    def __init__(self, filename):
        arr = (Camera * self.MAX_CAMS)()
        n = check(lib().s360_rig_load_json(str(filename).encode(), arr, self.MAX_CAMS))
        self.rig = (Camera * n)(*arr[:n])
        self.rig_side_only = [c for c in self.rig if c.is_side]
        if not self.rig_side_only:
            raise VrCamException(_capi.ERR_INVALID_ARG, "rig has no side cameras")
        self.top_index = lib().s360_rig_find_top(self.rig, n)
        self.bottom_index = lib().s360_rig_find_bottom(self.rig, n)

    def get_side_camera_count(self):
        return len(self.rig_side_only)

    def get_side_camera_id(self, idx):
        return self.rig_side_only[idx].id.decode()

    def get_top_camera_id(self):
        return self.rig[self.top_index].id.decode()

    def get_bottom_camera2_id(self):
        return self.rig[lib().s360_rig_find_bottom2(self.rig, len(self.rig))].id.decode()

    def get_bottom_camera_id(self):
        return self.rig[self.bottom_index].id.decode()

    def top_camera(self):
        return self.rig[self.top_index]

    def bottom_camera(self):
        return self.rig[self.bottom_index]


def make_params(**kw):
    """The gflags defaults of TestRenderStereoPanorama.cpp:44-70."""
    p = Params()
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
    p.side_flow_alg = b"pixflow_low"
    p.polar_flow_alg = b"pixflow_low"
    p.enable_pole_removal = 0
    p.poleremoval_flow_alg = b"pixflow_low"
    for k, v in kw.items():
        if not hasattr(p, k):
            raise TypeError("unknown flag: " + k)
        setattr(p, k, v.encode() if isinstance(v, str) else v)
    return p


class Context:
    """One s360_ctx: one GPU, its stream and its persistent HBM buffers."""

    def __init__(self, rig, params=None, device=0):
        self.rig = rig
        self.params = params if params is not None else make_params()
        h = C.c_void_p()
        check(lib().s360_create(C.byref(h), int(device), rig.rig, len(rig.rig), C.byref(self.params)))
        self.h = h
        g = Geometry()
        check(lib().s360_get_geometry(self.h, C.byref(g)), self.h)
        self.geometry = g

    def close(self):
        if getattr(self, "h", None):
            lib().s360_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _ck(self, rc):
        return check(rc, self.h)

    def synchronize(self):
        self._ck(lib().s360_synchronize(self.h))

    @property
    def stream(self):
        return lib().s360_stream(self.h)

    # ---- operators --------------------------------------------------------------------------
    def compute_optical_flow(self, i0, i1, alg="pixflow_low", hint="UNKNOWN", prev_flow=None, prev_i0=None,
                             prev_i1=None):
        i0, i1 = _u8(i0), _u8(i1)
        batched = i0.ndim == 4
        b = i0.shape[0] if batched else 1
        h, w = i0.shape[-3], i0.shape[-2]
        flow = np.empty(i0.shape[:-1] + (2,), np.float32)
        pf = np.ascontiguousarray(prev_flow, np.float32) if prev_flow is not None else None
        p0 = _u8(prev_i0) if prev_i0 is not None else None
        p1 = _u8(prev_i1) if prev_i1 is not None else None
        rc = lib().s360_compute_optical_flow_batch(self.h, alg.encode(), b, _p(i0), _p(i1), w, h, _p(pf), _p(p0), _p(p1),
                                                   HINT[hint], _p(flow))
        if rc == _capi.ERR_UNKNOWN_ALG:
            raise VrCamException(rc, "unrecognized flow algorithm name: " + alg)
        self._ck(rc)
        return flow

    def debug_flow_levels(self, i0, i1, alg="pixflow_low", hint="UNKNOWN"):
        i0, i1 = _u8(i0), _u8(i1)
        h, w = i0.shape[:2]
        cap = w * h * 8
        buf = np.empty(cap, np.float32)
        n = C.c_int()
        self._ck(lib().s360_debug_flow_levels(self.h, alg.encode(), _p(i0), _p(i1), w, h, HINT[hint], _p(buf),
                                              C.c_size_t(cap), C.byref(n)))
        return buf, n.value

    def spherical_warp_map(self, cam, dw, dh, l, r, t, b):
        m = np.empty((dh, dw, 2), np.float32)
        self._ck(lib().s360_spherical_warp_map(self.h, _p(m), dw, dh, C.byref(cam), C.c_float(l), C.c_float(r),
                                               C.c_float(t), C.c_float(b)))
        return m

    def bicubic_remap_to_spherical(self, src, cam, dw, dh, dc, l, r, t, b):
        src = _u8(src)
        dst = np.empty((dh, dw, dc), np.uint8)
        self._ck(lib().s360_bicubic_remap_to_spherical(self.h, _p(dst), dw, dh, dc, _p(src), src.shape[1], src.shape[0],
                                                       src.shape[2], C.byref(cam), C.c_float(l), C.c_float(r),
                                                       C.c_float(t), C.c_float(b)))
        return dst

    def combine_lazy_novel_views(self, image_l, image_r, flow_l_to_r, flow_r_to_l):
        g = self.geometry
        w = self.params.eqr_width // self.rig.get_side_camera_count()
        cl = np.empty((g.cam_image_height, w, 4), np.uint8)
        cr = np.empty_like(cl)
        self._ck(lib().s360_combine_lazy_novel_views(
            self.h, _p(_u8(image_l)), _p(_u8(image_r)), _p(np.ascontiguousarray(flow_l_to_r, np.float32)),
            _p(np.ascontiguousarray(flow_r_to_l, np.float32)), _p(cl), _p(cr)))
        return cl, cr

    def flatten_layers_deghost_prefer_base(self, bottom_layer, top_layer):
        b, t = _u8(bottom_layer), _u8(top_layer)
        out = np.empty_like(b)
        self._ck(lib().s360_flatten_layers_deghost_prefer_base(self.h, _p(b), _p(t), b.shape[1], b.shape[0], _p(out)))
        return out

    def offset_horizontal_wrap(self, src, offset):
        s = _u8(src)
        out = np.empty_like(s)
        self._ck(lib().s360_offset_horizontal_wrap(self.h, _p(s), s.shape[1], s.shape[0], s.shape[2], C.c_float(offset),
                                                   _p(out)))
        return out

    def feather_alpha_channel(self, src, erode_size):
        s = _u8(src)
        out = np.empty_like(s)
        self._ck(lib().s360_feather_alpha_channel(self.h, _p(s), s.shape[1], s.shape[0], int(erode_size), _p(out)))
        return out

    def pole_to_side_flow(self, side, pole, want_flow=False):
        side, pole = _u8(side), _u8(pole)
        out = np.empty_like(side)
        rows = pole.shape[0]
        fl = None
        if want_flow:
            fl = np.empty((rows, int(np.float32(side.shape[1]) * np.float32(1.2)), 2), np.float32)
        self._ck(lib().s360_pole_to_side_flow(self.h, _p(side), _p(pole), rows, _p(out), _p(fl)))
        return (out, fl) if want_flow else out

    def sharpen(self, bgr, sharpening):
        b = _u8(bgr).copy()
        self._ck(lib().s360_sharpen(self.h, _p(b), b.shape[1], b.shape[0], C.c_float(sharpening)))
        return b

    # ---- frame level -----------------------------------------------------------------------
    def upload_frame(self, side, top=None, bottom=None):
        for i, s in enumerate(side):
            s = _u8(s)
            self._ck(lib().s360_frame_upload_side(self.h, i, _p(s), s.shape[1], s.shape[0], s.shape[2]))
        if top is not None:
            t = _u8(top)
            self._ck(lib().s360_frame_upload_top(self.h, _p(t), t.shape[1], t.shape[0]))
        if bottom is not None:
            b = _u8(bottom)
            self._ck(lib().s360_frame_upload_bottom(self.h, _p(b), b.shape[1], b.shape[0]))

    def upload_raw(self, isp, camera, raw16):
        """A camera's raw Bayer frame (H x W uint16) through `isp` (surround360_amd.isp.CameraIsp, output_bpp 16) into
        this frame's source slot on the device. camera: side index, -1 top, -2 bottom."""
        r = np.ascontiguousarray(raw16, np.uint16)
        self._ck(lib().s360_frame_upload_raw(self.h, isp.h, int(camera), _p(r), r.shape[1], r.shape[0]))

    def upload_packed(self, isp, camera, frame, bits, w, h):
        """The same from the sensor's packed bytes of one w x h frame (8 or 12 bits per pixel, as in a capture's .bin container)."""
        fr = np.ascontiguousarray(frame, np.uint8)
        self._ck(lib().s360_frame_upload_packed(self.h, isp.h, int(camera), _p(fr), int(bits), int(w), int(h)))

    def upload_pole_removal(self, bottom2, mask, mask2):
        """Secondary bottom camera image + the two red pole masks (BGR) for enable_pole_removal (PoleRemoval.cpp:48-66)."""
        b2, m1, m2 = _u8(bottom2), _u8(mask), _u8(mask2)
        assert b2.shape == m1.shape == m2.shape and b2.shape[2] == 3
        self._ck(lib().s360_frame_upload_pole_removal(self.h, _p(b2), _p(m1), _p(m2), b2.shape[1], b2.shape[0]))

    # ---- frame slots: several independent frames through one launch sequence ----
    def set_frame_slots(self, n):
        self._ck(lib().s360_set_frame_slots(self.h, int(n)))

    def select_frame_slot(self, k):
        self._ck(lib().s360_select_frame_slot(self.h, int(k)))

    def render_batch(self, use_prev=False):
        self._ck(lib().s360_frame_render_batch(self.h, int(use_prev)))

    def render_slots(self, slots, use_prev=False):
        """s360_frame_render_slots: the given frame slots (ascending) as one batch."""
        arr = (C.c_int * len(slots))(*slots)
        self._ck(lib().s360_frame_render_slots(self.h, arr, len(slots), int(use_prev)))

    def render(self, use_prev=False):
        self._ck(lib().s360_frame_render(self.h, int(use_prev)))

    def render_pairs(self, p0, p1, use_prev=False):
        self._ck(lib().s360_frame_render_pairs(self.h, p0, p1, int(use_prev)))

    def finish(self, pole_mask=15, use_prev=False):
        self._ck(lib().s360_frame_finish(self.h, pole_mask, int(use_prev)))

    # ---- multi-GPU: sharded frame + native RCCL strip gather (s360.h, "multi-GPU") ----
    @staticmethod
    def comm_get_unique_id():
        buf = C.create_string_buffer(128)
        check(lib().s360_comm_get_unique_id(buf))
        return buf.raw

    @staticmethod
    def comm_library_path():
        """The file the RCCL entry points were resolved from (None: no librccl could be loaded)."""
        p = lib().s360_comm_library_path()
        return p.decode() if p else None

    def comm_init_rank(self, unique_id, rank, nranks):
        self._ck(lib().s360_comm_init_rank(self.h, C.c_char_p(unique_id), int(rank), int(nranks)))

    def comm_destroy(self):
        self._ck(lib().s360_comm_destroy(self.h))

    def comm_size(self):
        """ncclCommCount of the context's communicator (0: it has none)."""
        n = lib().s360_comm_size(self.h)
        if n < 0:
            self._ck(-1)
        return n

    def comm_rank(self):
        return lib().s360_comm_rank(self.h)

    def comm_stats(self, which):
        """{calls, bytes_sent, bytes_received} of exchange `which` (0 strips, 1 pole layers) since the communicator was made."""
        out = (C.c_ulonglong * 3)()
        self._ck(lib().s360_comm_stats(self.h, int(which), out))
        return {"calls": int(out[0]), "bytes_sent": int(out[1]), "bytes_received": int(out[2])}

    def gather_strips(self, bounds, root=0):
        arr = (C.c_int * len(bounds))(*bounds)
        self._ck(lib().s360_frame_gather_strips(self.h, arr, int(root)))

    def exchange_strips(self, bounds, need_mask):
        b = (C.c_int * len(bounds))(*bounds)
        n = (C.c_int * len(need_mask))(*need_mask)
        self._ck(lib().s360_frame_exchange_strips(self.h, b, n))

    def pole_units(self, pole_mask, use_prev=False):
        self._ck(lib().s360_frame_pole_units(self.h, int(pole_mask), int(bool(use_prev))))

    def gather_pole_layers(self, owner, root=0):
        o = (C.c_int * 4)(*owner)
        self._ck(lib().s360_frame_gather_pole_layers(self.h, o, int(root)))

    def composite(self, pole_mask=15):
        self._ck(lib().s360_frame_composite(self.h, int(pole_mask)))

    def comm_loopback(self, src_pair, dst_pair):
        self._ck(lib().s360_comm_loopback(self.h, int(src_pair), int(dst_pair)))

    def set_partition(self, p0, p1):
        self._ck(lib().s360_frame_set_partition(self.h, int(p0), int(p1)))

    def strip_ptr(self, eye):
        p = C.c_void_p()
        n = C.c_size_t()
        self._ck(lib().s360_frame_strip_ptr(self.h, eye, C.byref(p), C.byref(n)))
        return p.value, n.value

    def equirect_dev(self):
        p = C.c_void_p()
        n = C.c_size_t()
        self._ck(lib().s360_frame_equirect_dev(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def download_equirect(self):
        g = self.geometry
        out = np.empty((g.out_height, g.out_width, 3), np.uint8)
        self._ck(lib().s360_frame_download_equirect(self.h, _p(out)))
        return out

    def download_equirect_of(self, age, out=None):
        """age 0: the frame enqueued last; 1: the one before (fetched while the last one still renders). `out`: a buffer to
        fetch into (pinned_empty: one DMA transfer); the context's lock is released while the call waits for the frame."""
        g = self.geometry
        if out is None:
            out = np.empty((g.out_height, g.out_width, 3), np.uint8)
        assert out.shape == (g.out_height, g.out_width, 3) and out.dtype == np.uint8 and out.flags["C_CONTIGUOUS"]
        self._ck(lib().s360_frame_download_equirect_of(self.h, int(age), _p(out)))
        return out

    # ---- the equirect as a PNG file, encoded on the device (s360.h; replaces imwriteExceptionOnFail, TRSP:938-961) ----
    def set_output_double_buffer(self, on=True):
        """Two output buffers per slot without frame pipelining: a batch host enqueues step k+1, then fetches step k (age 1)."""
        self._ck(lib().s360_set_output_double_buffer(self.h, int(bool(on))))

    def set_png_encode(self, on=True):
        self._ck(lib().s360_set_png_encode(self.h, int(bool(on))))

    def download_png(self, age=0, out=None):
        """The finished frame as the bytes of a complete PNG file (rendered with set_png_encode(True)). `out`: a uint8 buffer of at
        least s360_frame_png_bound bytes (pinned_empty: one DMA transfer); returns a view of the file's bytes in it."""
        cap = int(lib().s360_frame_png_bound(self.h))
        if out is None:
            out = np.empty(cap, np.uint8)
        assert out.dtype == np.uint8 and out.ndim == 1 and out.flags["C_CONTIGUOUS"]
        n = C.c_size_t(0)
        self._ck(lib().s360_frame_download_png(self.h, int(age), _p(out), C.c_size_t(out.size), C.byref(n)))
        return out[:n.value]

    def encode_png(self, bgr):
        """Operator form: an (h, w, 3) uint8 B,G,R image -> the bytes of a PNG file (8-bit RGB), encoded on the device."""
        bgr = np.ascontiguousarray(bgr, np.uint8)
        h, w = bgr.shape[:2]
        assert bgr.shape == (h, w, 3)
        out = np.empty(int(lib().s360_png_bound(w, h)), np.uint8)
        n = C.c_size_t(0)
        self._ck(lib().s360_encode_png(self.h, _p(bgr), w, h, _p(out), C.c_size_t(out.size), C.byref(n)))
        return out[:n.value].tobytes()

    def uploads_complete(self):
        """Blocks until every upload enqueued so far has left its host buffer (needed for buffers from pinned_empty only)."""
        self._ck(lib().s360_frame_uploads_complete(self.h))

    def set_sweep_mode(self, mode):
        """'latency' (default) or 'throughput' — which sweep kernel PixFlow uses (bit-identical results)."""
        self._ck(lib().s360_set_sweep_mode(self.h, mode.encode()))

    def set_frame_pipelining(self, on=True):
        """One video stream: overlap the pole stage of frame k with the side stage of frame k+1 (same results)."""
        self._ck(lib().s360_set_frame_pipelining(self.h, 1 if on else 0))

    def cubemap(self, face_width, face_height, fmt="video"):
        """Stereo cubemap of the last rendered frame (TestRenderStereoPanorama.cpp:917-935), BGR."""
        whc = (C.c_int * 3)()
        self._ck(lib().s360_frame_cubemap(self.h, int(face_width), int(face_height), fmt.encode(), whc, None))
        out = np.empty((whc[1], whc[0], 3), np.uint8)
        self._ck(lib().s360_frame_cubemap(self.h, int(face_width), int(face_height), fmt.encode(), whc, _p(out)))
        return out

    def set_sharpening(self, sharpening):
        """FLAGS_sharpening for the frames rendered from now on (TRSP:56, :901)."""
        self._ck(lib().s360_set_sharpening(self.h, C.c_double(sharpening)))

    def keep_intermediates(self, on=True):
        self._ck(lib().s360_set_keep_intermediates(self.h, int(on)))

    def get_u8(self, name, idx=0):
        whc = (C.c_int * 3)()
        self._ck(lib().s360_frame_get_u8(self.h, name.encode(), idx, whc, None))
        d = np.empty((whc[1], whc[0], whc[2]), np.uint8)
        self._ck(lib().s360_frame_get_u8(self.h, name.encode(), idx, whc, _p(d)))
        return d

    def get_f32(self, name, idx=0):
        whc = (C.c_int * 3)()
        self._ck(lib().s360_frame_get_f32(self.h, name.encode(), idx, whc, None))
        d = np.empty((whc[1], whc[0], whc[2]), np.float32)
        self._ck(lib().s360_frame_get_f32(self.h, name.encode(), idx, whc, _p(d)))
        return d

    # ---- measurement ------------------------------------------------------------------------
    def profile_enable(self, on=True):
        self._ck(lib().s360_profile_enable(self.h, int(on)))

    def profile_get(self):
        names = C.create_string_buffer(4096)
        ms = (C.c_float * 64)()
        cnt = (C.c_int * 64)()
        n = self._ck(lib().s360_profile_get(self.h, names, 4096, ms, cnt, 64))
        ns = names.value.decode().split(";") if n else []
        return {ns[i]: (ms[i], cnt[i]) for i in range(n)}


def save_flow_to_file(flow, filename):
    """saveFlowToFile (CvUtil.cpp:159-177)."""
    f = np.ascontiguousarray(flow, np.float32)
    check(lib().s360_save_flow_to_file(str(filename).encode(), _p(f), f.shape[1], f.shape[0]))


def read_flow_from_file(filename):
    """readFlowFromFile (CvUtil.cpp:179-199) -> H x W x 2 float32."""
    w, h = C.c_int(), C.c_int()
    check(lib().s360_read_flow_from_file(str(filename).encode(), None, C.byref(w), C.byref(h), C.c_size_t(0)))
    out = np.empty((h.value, w.value, 2), np.float32)
    check(lib().s360_read_flow_from_file(str(filename).encode(), _p(out), C.byref(w), C.byref(h), C.c_size_t(out.size)))
    return out


# ---- reference-shaped operator objects ----------------------------------------------------------
class OpticalFlow:
    """OpticalFlowInterface implementation returned by make_optical_flow_by_name."""

    def __init__(self, ctx, name):
        self.ctx, self.name = ctx, name

    def compute_optical_flow(self, i0_bgra, i1_bgra, prev_flow=None, prev_i0_bgra=None, prev_i1_bgra=None,
                             hint="UNKNOWN"):
        return self.ctx.compute_optical_flow(i0_bgra, i1_bgra, self.name, hint, prev_flow, prev_i0_bgra, prev_i1_bgra)


def make_optical_flow_by_name(ctx, flow_alg_name):
    if flow_alg_name not in ("pixflow_low", "pixflow_search_20"):
        raise VrCamException(_capi.ERR_UNKNOWN_ALG, "unrecognized flow algorithm name: " + flow_alg_name)
    return OpticalFlow(ctx, flow_alg_name)


class NovelViewGeneratorAsymmetricFlow:
    """prepare() computes flowLtoR / flowRtoL (NovelView.cpp:270-299); combine_lazy_novel_views renders the
    left/right eye chunks of the pair (NovelView.cpp:226-268)."""

    def __init__(self, ctx, flow_alg_name):
        self.ctx, self.flow_alg_name = ctx, flow_alg_name
        self.image_l = self.image_r = self.flow_l_to_r = self.flow_r_to_l = None

    def prepare(self, color_image_l, color_image_r, prev_flow_l_to_r=None, prev_flow_r_to_l=None,
                prev_color_image_l=None, prev_color_image_r=None):
        alg = make_optical_flow_by_name(self.ctx, self.flow_alg_name)
        self.image_l, self.image_r = _u8(color_image_l).copy(), _u8(color_image_r).copy()
        self.flow_l_to_r = alg.compute_optical_flow(self.image_l, self.image_r, prev_flow_l_to_r, prev_color_image_l,
                                                    prev_color_image_r, "LEFT")
        self.flow_r_to_l = alg.compute_optical_flow(self.image_r, self.image_l, prev_flow_r_to_l, prev_color_image_r,
                                                    prev_color_image_l, "RIGHT")

    def get_flow_l_to_r(self):
        return self.flow_l_to_r

    def get_flow_r_to_l(self):
        return self.flow_r_to_l

    def combine_lazy_novel_views(self):
        return self.ctx.combine_lazy_novel_views(self.image_l, self.image_r, self.flow_l_to_r, self.flow_r_to_l)


class StereoPanoramaRenderer:
    """renderStereoPanorama for one rig + flag set; frames are uploaded, rendered and downloaded explicitly."""

    def __init__(self, rig_json_file, device=0, **flags):
        self.rig = RigDescription(rig_json_file)
        self.params = make_params(**flags)
        if self.params.eqr_width % self.rig.get_side_camera_count() != 0:
            raise VrCamException(_capi.ERR_INVALID_ARG,
                                 "eqr_width must be evenly divisible by the number of cameras")
        self.ctx = Context(self.rig, self.params, device)

    def render(self, side_images, top_image=None, bottom_image=None, use_prev=False):
        self.ctx.upload_frame(side_images, top_image, bottom_image)
        self.ctx.render(use_prev)
        return self.ctx.download_equirect()
