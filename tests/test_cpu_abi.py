"""CPU checks of the C ABI surface: the library loads, exports every symbol include/s360.h declares, the
host-side rig loader/geometry agree with the oracle, and compute entry points fail loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def header_symbols():
    src = open(os.path.join(ROOT, "include", "s360.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(s360_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(s360lib):
    from surround360_amd import _capi
    names = header_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(s360lib, n), "libs360.so does not export " + n
    assert sorted(_capi.SYMBOLS) == names, "python binding list out of sync with include/s360.h"


def test_version_and_device_count(s360lib):
    assert b"gfx950" in s360lib.s360_version()
    assert s360lib.s360_device_count() >= 0


def test_rig_loader_matches_oracle(s360lib, oracle, rig_json):
    from surround360_amd import render as R
    rig = R.RigDescription(rig_json)
    assert rig.get_side_camera_count() == 14
    assert [rig.get_side_camera_id(i) for i in (0, 13)] == ["cam1", "cam14"]
    assert rig.get_top_camera_id() == "cam0" and rig.get_bottom_camera_id() == "cam15"
    cams, ids = oracle.load_rig(rig_json)
    for i, cam in enumerate(rig.rig):
        r9 = (C.c_double * 9)()
        oracle.lib().orc_camera_rotation(C.byref(cams[i]), r9)
        assert list(cam.rotation) == list(r9), "rotation of %s differs from the oracle" % ids[i]
        # Camera::pixel on a few rig points
        for p in ([1e6, 2e5, -3e5], [-4e5, 9e5, 1e5], [10.0, -20.0, 1e6 if cam.type == 0 else 5.0]):
            pt = (C.c_double * 3)(*p)
            a = (C.c_double * 2)()
            b = (C.c_double * 2)()
            s360lib.s360_camera_pixel(C.byref(cam), pt, a)
            oracle.lib().orc_camera_pixel(C.byref(cams[i]), pt, b)
            assert list(a) == list(b) or (np.isnan(a[0]) and np.isnan(b[0]))
        assert s360lib.s360_camera_get_fov(C.byref(cam)) == oracle.lib().orc_camera_get_fov(C.byref(cams[i]))


@pytest.mark.parametrize("w,h", [(8400, 4096), (2058, 1029), (1008, 504)])
def test_geometry_matches_oracle_and_survey(s360lib, oracle, rig_json, w, h):
    from surround360_amd import render as R
    from surround360_amd._capi import Geometry
    rig = R.RigDescription(rig_json)
    p = R.make_params(eqr_width=w, eqr_height=h, enable_top=1, enable_bottom=1)
    g = Geometry()
    assert s360lib.s360_derive_geometry(rig.rig, len(rig.rig), C.byref(p), C.byref(g)) == 0
    cams, _ = oracle.load_rig(rig_json)
    f = oracle.Frame(cams, oracle.make_params(eqr_width=w, eqr_height=h, enable_top=1, enable_bottom=1))
    assert (g.cam_image_width, g.cam_image_height, g.overlap_image_width, g.num_novel_views, g.top_rows,
            g.bottom_rows) == (f.cam_image_width, f.cam_image_height, f.overlap_image_width, f.num_novel_views,
                               f.top_rows, f.bottom_rows)
    for a, b in ((g.h_radians, f.h_radians), (g.v_radians, f.v_radians),
                 (g.verge_at_infinity_slab_displacement, f.verge_disp),
                 (g.zero_parallax_novel_view_shift_pixels, f.zero_parallax_shift)):
        assert np.float32(a) == np.float32(b)
    ramp = (C.c_float * 4)()
    assert s360lib.s360_pole_ramp(rig.rig, len(rig.rig), ramp) == 0
    assert [np.float32(x) for x in ramp] == [np.float32(x) for x in f.pole_ramp()]
    if w == 8400:  # SURVEY.md §8 table (8K preset)
        assert (g.cam_image_width, g.cam_image_height, g.overlap_image_width, g.num_novel_views) == (1814, 1769, 1214, 600)
        assert g.top_rows == 2104 and abs(g.verge_at_infinity_slab_displacement - 196.9) < 0.05


def test_eqr_width_must_divide_by_camera_count(s360lib, rig_json):
    from surround360_amd import render as R
    with pytest.raises(R.VrCamException):  # TestRenderStereoPanorama.cpp:729-738
        R.StereoPanoramaRenderer(rig_json, eqr_width=8192, eqr_height=4096)


def test_unknown_flow_algorithm_name(s360lib, rig_json):
    from surround360_amd import render as R
    with pytest.raises(R.VrCamException):
        R.make_optical_flow_by_name(None, "pixflow_high")


def test_no_cpu_fallback(s360lib, rig_json):
    """Without a HIP device s360_create must fail with S360_ERR_NO_DEVICE, never compute on the CPU."""
    if s360lib.s360_device_count() > 0:
        pytest.skip("GPU present")
    from surround360_amd import _capi, render as R
    rig = R.RigDescription(rig_json)
    with pytest.raises(_capi.S360Error) as e:
        R.Context(rig, R.make_params(eqr_width=1008, eqr_height=504))
    assert e.value.code == _capi.ERR_NO_DEVICE


def test_product_does_not_link_the_oracle():
    """The shipped library and package must not reference oracle/ (ldd + source scan)."""
    import subprocess
    so = os.path.join(ROOT, "surround360_amd", "libs360.so")
    assert "oracle" not in subprocess.check_output(["ldd", so]).decode()
    for dp, _, files in os.walk(os.path.join(ROOT, "surround360_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                assert "oracle_lib" not in txt and "liboracle" not in txt and 'oracle/' not in txt, fn


def test_flow_file_round_trip(s360lib, tmp_path):
    """saveFlowToFile / readFlowFromFile byte format (CvUtil.cpp:159-199): int32 rows, int32 cols, (fx,fy) pairs."""
    rng = np.random.default_rng(0)
    f = rng.normal(size=(37, 53, 2)).astype(np.float32)
    path = str(tmp_path / "flowLtoR_0.bin").encode()
    assert s360lib.s360_save_flow_to_file(path, f.ctypes.data_as(C.c_void_p), 53, 37) == 0
    raw = open(path, "rb").read()
    assert np.frombuffer(raw[:8], np.int32).tolist() == [37, 53] and len(raw) == 8 + 37 * 53 * 8
    assert np.array_equal(np.frombuffer(raw[8:], np.float32).reshape(37, 53, 2), f)
    w, h = C.c_int(), C.c_int()
    out = np.empty_like(f)
    assert s360lib.s360_read_flow_from_file(path, out.ctypes.data_as(C.c_void_p), C.byref(w), C.byref(h),
                                            C.c_size_t(out.size)) == 0
    assert (w.value, h.value) == (53, 37) and np.array_equal(out, f)
    assert s360lib.s360_read_flow_from_file(b"/nonexistent/x.bin", None, C.byref(w), C.byref(h), C.c_size_t(0)) < 0


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: include/s360.h must compile as C99 and as C++11 on its own (no torch / HIP types)."""
    import subprocess
    hdr = os.path.join(ROOT, "include", "s360.h")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", hdr])
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", hdr])
    code = re.sub(r"/\*.*?\*/", "", open(hdr).read(), flags=re.S)  # comments may mention where pointers come from
    assert "torch" not in code and "hip/" not in code and "hipStream" not in code and "#include <std" in code


def test_page_locked_buffers_without_a_device(s360lib):
    """s360_host_alloc / s360_host_free / s360_frame_uploads_complete (round 4: streaming hosts) where no HIP device is attached:
    the allocation fails loudly (NULL + a message: there is no silent pageable fallback), freeing NULL is a no-op, a NULL
    context is an argument error."""
    import ctypes as C
    if s360lib.s360_device_count() > 0:
        p = s360lib.s360_host_alloc(1 << 20)
        assert p
        s360lib.s360_host_free(p)
    else:
        assert not s360lib.s360_host_alloc(1 << 20)
        assert b"s360_host_alloc" in s360lib.s360_last_error(None)
    assert not s360lib.s360_host_alloc(0)
    s360lib.s360_host_free(None)
    assert s360lib.s360_frame_uploads_complete(None) < 0
    assert s360lib.s360_frame_download_equirect_of(None, 0, C.c_void_p()) < 0


def test_rccl_library_resolution(s360lib):
    """s360_comm_library_path: librccl is loaded on first use and the file the entry points came from is reported (the
    hardware-day checklist, tools/gpu_multi.sh, prints it first); S360_RCCL_LIB names another one, and a wrong name is an error
    with the reason, not a silent fall-back to the default."""
    import subprocess
    import sys
    p = s360lib.s360_comm_library_path()
    assert p and b"rccl" in p, p
    code = ("import ctypes as C; L = C.CDLL(%r); L.s360_comm_library_path.restype = C.c_char_p; L.s360_last_error.restype = C.c_char_p;"
            "p = L.s360_comm_library_path(); print(p, L.s360_last_error(None))" % os.path.join(ROOT, "surround360_amd", "libs360.so"))
    out = subprocess.check_output([sys.executable, "-c", code], env=dict(os.environ, S360_RCCL_LIB="/nonexistent/librccl.so"), text=True)
    assert out.startswith("None") and "S360_RCCL_LIB=/nonexistent/librccl.so" in out, out
