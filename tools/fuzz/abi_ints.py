import os as _os; _os.makedirs('/tmp/s360_fuzz', exist_ok=True)
import os, sys, ctypes as C, itertools
ROOT = __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
from surround360_amd import _capi
_capi.LIB_PATH = sys.argv[1]
from surround360_amd import render as R
import numpy as np, rigutil
L = _capi.lib()
path = rigutil.scaled_rig_json(ROOT + '/tests/golden/rig_17cam.json', '/tmp/s360_fuzz/rig_abi.json', 64/2048.0)
rig = R.RigDescription(path)
ctx = R.Context(rig, R.make_params(eqr_width=140, eqr_height=70, enable_top=1, enable_bottom=1))
h = ctx.h
buf = np.zeros(64*64*16, np.uint8); fbuf = np.zeros(64*64*8, np.float32)
P8 = buf.ctypes.data_as(C.POINTER(C.c_uint8)); PF = fbuf.ctypes.data_as(C.POINTER(C.c_float))
def call(name, *a):
    f = getattr(L, name); f.restype = C.c_int; f.argtypes = None
    r = f(*a); return r
vals = [0, -1, 1, 2, 3, 5, -2147483648]
n=0
for w,hh in itertools.product(vals, vals):
    for nm, args in (
      ('s360_flatten_layers_deghost_prefer_base', (h, P8, P8, C.c_int(w), C.c_int(hh), P8)),
      ('s360_offset_horizontal_wrap', (h, P8, C.c_int(w), C.c_int(hh), C.c_int(4), C.c_float(1.5), P8)),
      ('s360_feather_alpha_channel', (h, P8, C.c_int(w), C.c_int(hh), C.c_int(3), P8)),
      ('s360_sharpen', (h, P8, C.c_int(w), C.c_int(hh), C.c_float(0.25))),
      ('s360_compute_optical_flow', (h, b"pixflow_low", P8, P8, C.c_int(w), C.c_int(hh), None, None, None, C.c_int(0), PF)),
      ('s360_frame_upload_side', (h, C.c_int(0), P8, C.c_int(w), C.c_int(hh), C.c_int(3))),
      ('s360_frame_upload_top', (h, P8, C.c_int(w), C.c_int(hh))),
      ('s360_save_flow_to_file', (b"/tmp/s360_fuzz/x.bin", PF, C.c_int(w), C.c_int(hh))),
    ):
        r = call(nm, *args); n+=1
        if r >= 0 and (w <= 0 or hh <= 0): print('ACCEPTED non-positive', nm, w, hh, r, flush=True)
for ch in (0,1,2,5,-1):
    r = call('s360_offset_horizontal_wrap', h, P8, C.c_int(8), C.c_int(8), C.c_int(ch), C.c_float(1.0), P8)
    if r >= 0 and ch not in (1,3,4): print('ACCEPTED channels', ch)
for idx in (-1, 14, 1000, -2147483648):
    r = call('s360_frame_upload_side', h, C.c_int(idx), P8, C.c_int(8), C.c_int(8), C.c_int(3))
    if r >= 0: print('ACCEPTED side idx', idx)
for a,b in ((-1,3),(3,1),(0,15),(14,14),(5,-2)):
    r = call('s360_frame_render_pairs', h, C.c_int(a), C.c_int(b), C.c_int(0))
    if r >= 0: print('ACCEPTED pair range', a, b, r)
for k in (-1, 1, 64, 65, 1<<30):
    r = call('s360_select_frame_slot', h, C.c_int(k))
    if r >= 0: print('ACCEPTED slot', k)
for k in (-1, 0, 65, 1<<30):
    r = call('s360_set_frame_slots', h, C.c_int(k))
    if r >= 0: print('ACCEPTED nslots', k)
for age in (-1, 2, 100):
    r = call('s360_frame_download_equirect_of', h, C.c_int(age), P8)
    if r >= 0: print('ACCEPTED age', age)
for fw,fh in ((0,8),(8,0),(-1,8),(1<<20,1<<20)):
    whc=(C.c_int*3)()
    r = call('s360_frame_cubemap', h, C.c_int(fw), C.c_int(fh), b"video", whc, None)
    if r >= 0: print('ACCEPTED cubemap', fw, fh, list(whc))
for nm in (b"nonsense", b"", b"projection"):
    whc=(C.c_int*3)()
    r = call('s360_frame_get_u8', h, nm, C.c_int(99999), whc, None)
    if r >= 0: print('ACCEPTED get_u8', nm, list(whc))
print('calls', n, 'done')
