import os as _os; _os.makedirs('/tmp/s360_fuzz', exist_ok=True)
import os, sys, ctypes as C, itertools
ROOT = __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
from surround360_amd import _capi
_capi.LIB_PATH = sys.argv[1]
from surround360_amd import isp as I
import numpy as np, isputil
L = _capi.lib()
raw = np.zeros(64*64*4, np.uint16); out = np.zeros(64*64*16, np.uint16)
PR = raw.ctypes.data_as(C.POINTER(C.c_uint16)); PO = out.ctypes.data_as(C.c_void_p); PB = raw.ctypes.data_as(C.POINTER(C.c_uint8))
n=0
for js,kw in ((isputil.CONFIG_FULL, dict(output_bpp=16)), (isputil.CONFIG_FULL, dict(output_bpp=8, demosaic_filter=0, resize=2)), (isputil.CONFIG_MINIMAL, dict(output_bpp=8, resize=8)), (isputil.CONFIG_MINIMAL, dict(output_bpp=8, resize=0)), (isputil.CONFIG_MINIMAL, dict(output_bpp=7)), (isputil.CONFIG_MINIMAL, dict(output_bpp=8, resize=-2)), (isputil.CONFIG_MINIMAL, dict(output_bpp=8, demosaic_filter=7)),
              (isputil.CONFIG_FULL, dict(output_bpp=16, pipe=1)), (isputil.CONFIG_FULL, dict(output_bpp=8, pipe=2)), (isputil.CONFIG_FULL, dict(output_bpp=8, pipe=1, resize=2)),
              (isputil.CONFIG_FULL, dict(output_bpp=8, pipe=3)), (isputil.stuck_pixel_config(1, 1, 2.0), dict(output_bpp=16)), (isputil.stuck_pixel_config(3, 0, 2.0), dict(output_bpp=8, resize=2))):
    try:
        cfg = I.config_from_json(js, **kw)
        isp = I.CameraIsp(cfg)
    except _capi.S360Error as e:
        print('create rejected', kw, str(e)[:80]); continue
    h = isp.h
    for w,hh in itertools.product([0,-1,1,2,3,4,5,6,7,8,9,16,17,-2147483648], repeat=2):
        f = L.s360_isp_process; f.restype=C.c_int; f.argtypes=None
        r = f(h, PR, C.c_int(w), C.c_int(hh), PO); n+=1
        if r >= 0 and (w <= 0 or hh <= 0): print('ACCEPTED', kw, w, hh)
        for bits in (0, 7, 8, 12, 16, -1):
            f = L.s360_isp_process_packed; f.restype=C.c_int; f.argtypes=None
            r = f(h, PB, C.c_int(bits), C.c_int(w), C.c_int(hh), PO); n+=1
            if r >= 0 and (w <= 0 or hh <= 0 or bits not in (8,12)): print('ACCEPTED packed', kw, bits, w, hh)
    if kw.get('pipe'):  # the generated functions' own signature: sizes, strides, selectors, NULL tables
        class A(C.Structure):
            _fields_ = [("input", C.c_void_p), ("input_stride", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("vignette_h", C.c_void_p),
                        ("vignette_v", C.c_void_p), ("black_level", C.c_float * 3), ("white_balance_gain", C.c_float * 3), ("clamp_min", C.c_float * 3),
                        ("clamp_max", C.c_float * 3), ("sharpening", C.c_float * 3), ("sharpening_support", C.c_float), ("noise_core", C.c_float),
                        ("ccm", C.c_void_p), ("tone_table", C.c_void_p), ("bgr", C.c_int32), ("bayer_pattern", C.c_int32), ("fast", C.c_int32),
                        ("output_bpp", C.c_int32), ("output", C.c_void_p)]
        tab = np.ones(4096 * 3, np.float32); tone = np.zeros(4096 * 3, np.uint16)
        f = L.s360_isp_pipe_generated; f.restype = C.c_int; f.argtypes = None
        for w, hh, stride, bpp, pat, nul in itertools.product([0, -1, 15, 16, 33, 64], [0, 15, 16, 40], [0, 16, 64, 100], [8, 16, 12], [0, 1, 2, -1], [0, 1, 2, 3, 4, 5]):
            a = A()
            a.input = raw.ctypes.data; a.input_stride = stride; a.width = w; a.height = hh
            a.vignette_h = tab.ctypes.data; a.vignette_v = tab.ctypes.data; a.ccm = tab.ctypes.data; a.tone_table = tone.ctypes.data; a.output = out.ctypes.data
            for k in range(3):
                a.white_balance_gain[k] = 1.0; a.clamp_max[k] = 1.0
            a.sharpening_support = 0.01; a.noise_core = 100.0; a.bgr = 1; a.bayer_pattern = pat; a.fast = kw['pipe'] == 2; a.output_bpp = bpp
            if nul: setattr(a, ["input", "vignette_h", "vignette_v", "ccm", "tone_table", "output"][nul - 1 if nul < 6 else 5], None)
            r = f(h, C.byref(a)); n += 1
            bad = w < 16 or hh < 16 or stride < w or bpp not in (8, 16) or pat not in (0, 1) or nul
            if r >= 0 and bad: print('ACCEPTED generated', w, hh, stride, bpp, pat, nul)
            if r < 0 and not bad: print('REFUSED generated', w, hh, stride, bpp, pat, _capi.lib().s360_last_error(None))
        r = f(h, None); n += 1
        assert r < 0
    isp.close() if hasattr(isp,'close') else None
print('calls', n)
