"""ctypes bindings of libs360.so (include/s360.h). The library is built in-tree by
surround360_amd/csrc/Makefile; importing this module never builds anything and never falls
back to a CPU implementation — if the shared object is missing, loading fails loudly."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libs360.so")

OK = 0
ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_UNKNOWN_ALG, ERR_IO, ERR_STATE = -1, -2, -3, -4, -5, -6
HINT = {"UNKNOWN": 0, "RIGHT": 1, "DOWN": 2, "LEFT": 3, "UP": 4}


class Camera(C.Structure):
    _fields_ = [
        ("type", C.c_int32), ("is_side", C.c_int32),
        ("position", C.c_double * 3), ("rotation", C.c_double * 9), ("resolution", C.c_double * 2),
        ("principal", C.c_double * 2), ("distortion", C.c_double * 2), ("focal", C.c_double * 2),
        ("fov_threshold", C.c_double), ("id", C.c_char * 32),
    ]


class Params(C.Structure):
    _fields_ = [
        ("interpupilary_dist", C.c_double), ("zero_parallax_dist", C.c_double), ("sharpening", C.c_double),
        ("side_alpha_feather_size", C.c_int32), ("std_alpha_feather_size", C.c_int32),
        ("enable_top", C.c_int32), ("enable_bottom", C.c_int32),
        ("eqr_width", C.c_int32), ("eqr_height", C.c_int32),
        ("final_eqr_width", C.c_int32), ("final_eqr_height", C.c_int32),
        ("side_flow_alg", C.c_char * 32), ("polar_flow_alg", C.c_char * 32),
        ("enable_pole_removal", C.c_int32), ("poleremoval_flow_alg", C.c_char * 32),
    ]


class Geometry(C.Structure):
    _fields_ = [
        ("cam_image_width", C.c_int32), ("cam_image_height", C.c_int32),
        ("overlap_image_width", C.c_int32), ("num_novel_views", C.c_int32),
        ("top_rows", C.c_int32), ("bottom_rows", C.c_int32), ("out_width", C.c_int32), ("out_height", C.c_int32),
        ("h_radians", C.c_float), ("v_radians", C.c_float), ("fov_horizontal_radians", C.c_float),
        ("verge_at_infinity_slab_displacement", C.c_float), ("zero_parallax_novel_view_shift_pixels", C.c_float),
    ]


ISP_MAX_CURVE_POINTS = 16


class IspConfig(C.Structure):
    """s360_isp_config (include/s360.h)."""
    _fields_ = [
        ("black_level", C.c_float * 3), ("clamp_min", C.c_float * 3), ("clamp_max", C.c_float * 3),
        ("white_balance_gain", C.c_float * 3), ("ccm", C.c_float * 9), ("saturation", C.c_float),
        ("contrast", C.c_float), ("gamma", C.c_float * 3), ("low_key_boost", C.c_float * 3),
        ("high_key_boost", C.c_float * 3), ("sharpening", C.c_float * 3), ("sharpening_support", C.c_float),
        ("noise_core", C.c_float), ("n_vignette_h", C.c_int32), ("n_vignette_v", C.c_int32),
        ("vignette_roll_off_h", (C.c_float * 3) * ISP_MAX_CURVE_POINTS),
        ("vignette_roll_off_v", (C.c_float * 3) * ISP_MAX_CURVE_POINTS),
        ("stuck_pixel_radius", C.c_int32), ("bayer_pattern", C.c_int32), ("output_bpp", C.c_int32),
        ("demosaic_filter", C.c_int32), ("resize", C.c_int32), ("disable_tone_curve", C.c_int32),
        ("black_level_offset", C.c_int32),
        ("stuck_pixel_threshold", C.c_int32), ("stuck_pixel_darkness_threshold", C.c_float),
        ("pipe", C.c_int32),
    ]


# every symbol include/s360.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "s360_version", "s360_device_count", "s360_last_error", "s360_rig_load_json", "s360_camera_init",
    "s360_camera_pixel", "s360_camera_get_fov", "s360_rig_find_top", "s360_rig_find_bottom", "s360_rig_find_bottom2", "s360_camera_usable_pixels_radius", "s360_derive_geometry",
    "s360_pole_ramp", "s360_create",
    "s360_destroy", "s360_get_geometry", "s360_stream", "s360_synchronize", "s360_set_sharpening", "s360_compute_optical_flow",
    "s360_compute_optical_flow_batch", "s360_bicubic_remap_to_spherical", "s360_spherical_warp_map",
    "s360_combine_lazy_novel_views", "s360_flatten_layers_deghost_prefer_base", "s360_offset_horizontal_wrap",
    "s360_feather_alpha_channel", "s360_pole_to_side_flow", "s360_sharpen", "s360_frame_upload_side",
    "s360_frame_upload_top", "s360_frame_upload_bottom", "s360_frame_upload_pole_removal", "s360_frame_set_prev_pole_removal", "s360_frame_render", "s360_frame_render_pairs",
    "s360_frame_set_prev_side", "s360_frame_set_prev_pole", "s360_frame_strip_ptr", "s360_frame_finish", "s360_frame_download_equirect", "s360_frame_equirect_dev",
    "s360_frame_cubemap", "s360_frame_get_u8", "s360_frame_get_f32", "s360_set_keep_intermediates", "s360_set_sweep_mode", "s360_set_frame_pipelining", "s360_debug_flow_levels",
    "s360_profile_enable", "s360_profile_get", "s360_save_flow_to_file", "s360_read_flow_from_file",
    "s360_comm_get_unique_id", "s360_comm_library_path", "s360_comm_init_rank", "s360_comm_init_all", "s360_comm_destroy", "s360_comm_size", "s360_comm_rank", "s360_comm_stats",
    "s360_frame_gather_strips", "s360_frame_exchange_strips", "s360_frame_pole_units", "s360_frame_gather_pole_layers", "s360_frame_composite", "s360_comm_loopback", "s360_frame_set_partition",
    "s360_frame_download_equirect_of", "s360_set_frame_slots", "s360_select_frame_slot", "s360_frame_render_batch", "s360_frame_render_slots",
    "s360_isp_config_defaults", "s360_isp_config_from_json", "s360_isp_create", "s360_isp_destroy", "s360_isp_process",
    "s360_isp_config_tables", "s360_isp_process_packed", "s360_frame_upload_raw", "s360_frame_upload_packed", "s360_isp_pipe_generated",
    "s360_host_alloc", "s360_host_free", "s360_frame_uploads_complete",
    "s360_set_output_double_buffer", "s360_set_png_encode", "s360_frame_png_bound", "s360_frame_download_png", "s360_frame_download_png_slot", "s360_frame_download_equirect_slot", "s360_png_bound", "s360_encode_png",
]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libs360.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(surround360_amd has no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        L.s360_version.restype = C.c_char_p
        L.s360_last_error.restype = C.c_char_p
        L.s360_last_error.argtypes = [C.c_void_p]
        L.s360_camera_get_fov.restype = C.c_double
        L.s360_camera_usable_pixels_radius.restype = C.c_float
        L.s360_comm_library_path.restype = C.c_char_p
        L.s360_stream.restype = C.c_void_p
        L.s360_stream.argtypes = [C.c_void_p]
        L.s360_host_alloc.restype = C.c_void_p
        L.s360_host_alloc.argtypes = [C.c_size_t]
        L.s360_host_free.restype = None
        L.s360_frame_png_bound.restype = C.c_size_t
        L.s360_frame_png_bound.argtypes = [C.c_void_p]
        L.s360_png_bound.restype = C.c_size_t
        L.s360_png_bound.argtypes = [C.c_int, C.c_int]
        L.s360_host_free.argtypes = [C.c_void_p]
        L.s360_isp_config_defaults.restype = None
        L.s360_isp_destroy.restype = None
        L.s360_isp_destroy.argtypes = [C.c_void_p]
        for name in ("s360_destroy", "s360_synchronize", "s360_get_geometry"):
            getattr(L, name).argtypes = None
        _lib = L
    return _lib


class S360Error(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("s360 error %d: %s" % (code, msg))
        self.code = code


def check(rc, ctx=None):
    if rc < 0:
        msg = lib().s360_last_error(ctx)
        raise S360Error(rc, msg.decode() if msg else "")
    return rc
