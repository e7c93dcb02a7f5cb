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
for js,kw in ((isputil.CONFIG_FULL, dict(output_bpp=16)), (isputil.CONFIG_FULL, dict(output_bpp=8, demosaic_filter=0, resize=2)), (isputil.CONFIG_MINIMAL, dict(output_bpp=8, resize=8)), (isputil.CONFIG_MINIMAL, dict(output_bpp=8, resize=0)), (isputil.CONFIG_MINIMAL, dict(output_bpp=7)), (isputil.CONFIG_MINIMAL, dict(output_bpp=8, resize=-2)), (isputil.CONFIG_MINIMAL, dict(output_bpp=8, demosaic_filter=7))):
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
    isp.close() if hasattr(isp,'close') else None
print('calls', n)
