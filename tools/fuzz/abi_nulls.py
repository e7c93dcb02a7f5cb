"""Every pointer argument of the C ABI as NULL (one at a time, the others valid): an error code, never a crash.
usage: python tools/fuzz/abi_nulls.py <libs360 build>   (tools/libs360_emu.so or tools/fuzz/libs360_asan.so)"""
import ctypes as C
import itertools
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
if len(sys.argv) > 2:
    from surround360_amd import _capi
    _capi.LIB_PATH = sys.argv[1]
    from surround360_amd import render as R
    import numpy as np, rigutil
    L = _capi.lib()
    os.makedirs('/tmp/s360_fuzz', exist_ok=True)
    path = rigutil.scaled_rig_json(ROOT + '/tests/golden/rig_17cam.json', '/tmp/s360_fuzz/rig_nulls.json', 64 / 2048.0)
    rig = R.RigDescription(path)
    ctx = R.Context(rig, R.make_params(eqr_width=140, eqr_height=70, enable_top=1, enable_bottom=1))
    h = ctx.h
    buf = np.zeros(1 << 20, np.uint8); fbuf = np.zeros(1 << 18, np.float32)
    P8 = buf.ctypes.data_as(C.c_void_p); PF = fbuf.ctypes.data_as(C.c_void_p)
    cam = C.byref(rig.rig_side_only[0])
    i3 = (C.c_int * 3)(); i4 = (C.c_int * 4)(0, 0, 0, 0); bnd = (C.c_int * 2)(0, 14); need = (C.c_int * 1)(3)
    geo = _capi.Geometry() if hasattr(_capi, 'Geometry') else (C.c_int * 32)()
    vp = C.c_void_p(); sz = C.c_size_t()
    w, hh = C.c_int(16), C.c_int(16)
    calls = {
      's360_get_geometry': [h, C.byref(geo)],
      's360_compute_optical_flow': [h, b"pixflow_low", P8, P8, w, hh, None, None, None, C.c_int(0), PF],
      's360_compute_optical_flow_batch': [h, b"pixflow_low", C.c_int(1), P8, P8, w, hh, None, None, None, C.c_int(0), PF],
      's360_bicubic_remap_to_spherical': [h, P8, w, hh, C.c_int(4), P8, w, hh, C.c_int(3), cam, C.c_float(.1), C.c_float(-.1), C.c_float(.1), C.c_float(-.1)],
      's360_spherical_warp_map': [h, PF, w, hh, cam, C.c_float(.1), C.c_float(-.1), C.c_float(.1), C.c_float(-.1)],
      's360_flatten_layers_deghost_prefer_base': [h, P8, P8, w, hh, P8],
      's360_offset_horizontal_wrap': [h, P8, w, hh, C.c_int(4), C.c_float(1.5), P8],
      's360_feather_alpha_channel': [h, P8, w, hh, C.c_int(3), P8],
      's360_sharpen': [h, P8, w, hh, C.c_float(0.25)],
      's360_frame_upload_side': [h, C.c_int(0), P8, w, hh, C.c_int(3)],
      's360_frame_upload_top': [h, P8, w, hh],
      's360_frame_upload_bottom': [h, P8, w, hh],
      's360_frame_strip_ptr': [h, C.c_int(0), C.byref(vp), C.byref(sz)],
      's360_frame_download_equirect': [h, P8],
      's360_frame_equirect_dev': [h, C.byref(vp), C.byref(sz)],
      's360_frame_cubemap': [h, C.c_int(8), C.c_int(8), b"video", i3, P8],
      's360_frame_get_u8': [h, b"projection", C.c_int(0), i3, P8],
      's360_frame_get_f32': [h, b"flow_l_to_r", C.c_int(0), i3, PF],
      's360_frame_set_prev_side': [h, C.c_int(0), PF, PF, P8, P8],
      's360_frame_set_prev_pole': [h, C.c_int(0), PF, P8, P8],
      's360_frame_gather_strips': [h, bnd, C.c_int(0)],
      's360_frame_exchange_strips': [h, bnd, need],
      's360_frame_gather_pole_layers': [h, i4, C.c_int(0)],
      's360_set_sweep_mode': [h, b"latency"],
      's360_profile_get': [h, P8, C.c_size_t(64), PF, i3, C.c_int(1)],
      's360_save_flow_to_file': [b"/tmp/s360_fuzz/n.bin", PF, w, hh],
      's360_read_flow_from_file': [b"/tmp/s360_fuzz/n.bin", PF, C.byref(w), C.byref(hh), C.c_size_t(1 << 18)],
      's360_rig_load_json': [path.encode(), C.cast(P8, C.c_void_p), C.c_int(20)],
      's360_isp_config_from_json': [b"{}", P8],
      's360_comm_get_unique_id': [P8],
      's360_comm_init_rank': [h, P8, C.c_int(0), C.c_int(1)],
    }
    name = sys.argv[2]; k = int(sys.argv[3])
    a = list(calls[name])
    a[k] = None
    f = getattr(L, name); f.restype = C.c_int; f.argtypes = None
    r = f(*a)
    print('RC', r)
    sys.exit(0)
# driver: one subprocess per (function, pointer argument)
import re
sig = {
  's360_get_geometry': [0, 1], 's360_compute_optical_flow': [0, 1, 2, 3, 10], 's360_compute_optical_flow_batch': [0, 1, 3, 4, 11],
  's360_bicubic_remap_to_spherical': [0, 1, 5, 9], 's360_spherical_warp_map': [0, 1, 4], 's360_flatten_layers_deghost_prefer_base': [0, 1, 2, 5],
  's360_offset_horizontal_wrap': [0, 1, 6], 's360_feather_alpha_channel': [0, 1, 5], 's360_sharpen': [0, 1], 's360_frame_upload_side': [0, 2],
  's360_frame_upload_top': [0, 1], 's360_frame_upload_bottom': [0, 1], 's360_frame_strip_ptr': [0, 2, 3], 's360_frame_download_equirect': [0, 1],
  's360_frame_equirect_dev': [0, 1, 2], 's360_frame_cubemap': [0, 3, 4], 's360_frame_get_u8': [0, 1, 3], 's360_frame_get_f32': [0, 1, 3],
  's360_frame_set_prev_side': [0, 2, 3, 4, 5], 's360_frame_set_prev_pole': [0, 2, 3, 4], 's360_frame_gather_strips': [0, 1], 's360_frame_exchange_strips': [0, 1, 2],
  's360_frame_gather_pole_layers': [0, 1], 's360_set_sweep_mode': [0, 1], 's360_profile_get': [0, 1, 3, 4], 's360_save_flow_to_file': [0, 1],
  's360_read_flow_from_file': [0, 2, 3], 's360_rig_load_json': [0, 1], 's360_isp_config_from_json': [0, 1], 's360_comm_get_unique_id': [0],
  's360_comm_init_rank': [0, 1],
}
bad = 0
for name, ks in sig.items():
    for k in ks:
        r = subprocess.run([sys.executable, __file__, sys.argv[1], name, str(k)], capture_output=True, text=True, timeout=300)
        m = re.search(r'RC (-?\d+)', r.stdout)
        if r.returncode != 0 or not m:
            bad += 1
            print('CRASH', name, 'arg', k, 'rc', r.returncode, (r.stderr.strip().splitlines() or [''])[-1][:160], flush=True)
        elif int(m.group(1)) >= 0 and not (name in ('s360_frame_cubemap', 's360_frame_get_u8', 's360_frame_get_f32', 's360_read_flow_from_file', 's360_profile_get') and k >= 2):
            print('accepted NULL:', name, 'arg', k, 'rc', m.group(1), flush=True)
print('done,', bad, 'crashes')
