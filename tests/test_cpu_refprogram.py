"""The reference's WHOLE program against the oracle. oracle/_ref/TestRenderStereoPanorama is the reference's own
TestRenderStereoPanorama.cpp (main(), flags, threads, file layout) with Camera.cpp, RigDescription.cpp, ImageWarper.cpp,
PoleRemoval.cpp, NovelView.cpp, PixFlow.h, CvUtil.cpp ... compiled from /root/reference over stand-ins for OpenCV, Eigen,
folly, gflags and glog (oracle/ref_shim; built by `make -C oracle ref` where the reference exists). Every file it writes
— the stereo equirect, the cubemap, all 28 + 4 flows, the overlap and pole state images, the pole-removal state — must
equal the oracle's result for the same inputs bit for bit, and its digests must be the committed golden ones
(tests/golden/refprogram_golden.json), which is what the GPU tests hold the HIP program to on a box without the reference.
What this pins: everything the reference's authors wrote on the path. What it cannot: the OpenCV / Eigen primitives, which
are the stand-ins' on both sides (oracle/cvlite.h header)."""
import json
import os

import numpy as np
import pytest
from PIL import Image

import oracle_lib as O
import refprog
import rigutil

def _have_program():
    if not os.path.exists(refprog.REF_EXE) and os.path.isdir("/root/reference/surround360_render/source"):
        import subprocess
        subprocess.call(["make", "-C", os.path.join(refprog.ROOT, "oracle"), "-s", "ref"])
    return os.path.exists(refprog.REF_EXE)


pytestmark = pytest.mark.skipif(not _have_program(), reason="oracle/_ref/TestRenderStereoPanorama not there and no "
                                "/root/reference to build it from (make -C oracle ref)")

EYES = ["top_left", "top_right", "bottom_left", "bottom_right"]


def _png_bgr(path):
    a = np.asarray(Image.open(path))
    return a[:, :, [2, 1, 0, 3]] if a.shape[2] == 4 else a[:, :, ::-1]


def _flow_bin(path):
    hdr = np.fromfile(path, dtype=np.int32, count=2)
    return np.fromfile(path, dtype=np.float32, offset=8).reshape(int(hdr[0]), int(hdr[1]), 2)


def _same_bits(a, b):
    return a.shape == b.shape and np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


@pytest.fixture(scope="module")
def rig_small(tmp_path_factory):
    return rigutil.scaled_rig_json(os.path.join(refprog.ROOT, "tests", "golden", "rig_17cam.json"),
                                   str(tmp_path_factory.mktemp("rig") / "rig_small.json"), refprog.CAM / 2048.0)


@pytest.mark.parametrize("name", list(refprog.CASES))
def test_reference_program_equals_oracle_and_golden(tmp_path, rig_small, name):
    out = refprog.run_case(refprog.REF_EXE, str(tmp_path), rig_small, name)
    frames, extra = refprog.CASES[name]
    flag = lambda k, d=None: extra[extra.index(k) + 1] if k in extra else d  # noqa: E731
    removal = "--enable_pole_removal" in extra
    params = O.make_params(eqr_width=refprog.EQR_W, eqr_height=refprog.EQR_H, final_eqr_width=refprog.FINAL,
                           final_eqr_height=refprog.FINAL, enable_top=int("--enable_top" in extra),
                           enable_bottom=int("--enable_bottom" in extra), enable_pole_removal=int(removal),
                           sharpening=float(flag("--sharpening", 0.0)),
                           side_flow_search20=int(flag("--side_flow_alg") == "pixflow_search_20"))
    cams, ids = O.load_rig(rig_small)
    of = O.Frame(cams, params)
    side_ids, top_id, bottoms = refprog.rig_ids(rig_small)
    for k, f in enumerate(frames):
        imgs = refprog.frame_images(rig_small, k)
        bottom_id = bottoms[0]
        if removal:
            b2 = ids[of.bottom2_index()]
            bottom_id = [b for b in bottoms if b != b2][0]
            of.set_pole_removal(imgs[b2], refprog.pole_mask(refprog.CAM, 10 - bottoms.index(bottom_id), 20),
                                refprog.pole_mask(refprog.CAM, 10 - bottoms.index(b2), 20))
        want, _ = of.render([imgs[c] for c in side_ids], imgs[top_id] if "--enable_top" in extra else None, imgs[bottom_id],
                            use_prev=k > 0)
        assert np.array_equal(_png_bgr(os.path.join(out, "eqr_%s.png" % f)), want), "%s frame %s: equirect" % (name, f)
        if flag("--cubemap_width"):
            cw = int(flag("--cubemap_width"))
            assert np.array_equal(_png_bgr(os.path.join(out, "cube_%s.png" % f)), of.cubemap(cw, cw, flag("--cubemap_format")))
        fdir, idir = os.path.join(out, "flow", f), os.path.join(out, "debug", f, "flow_images")
        for i in range(len(side_ids)):  # the temporal state of the 14 pairs (TestRenderStereoPanorama.cpp:201-255)
            assert _same_bits(_flow_bin(os.path.join(fdir, "flowLtoR_%d.bin" % i)), of.get_f32("flow_l_to_r", i)), (f, i)
            assert _same_bits(_flow_bin(os.path.join(fdir, "flowRtoL_%d.bin" % i)), of.get_f32("flow_r_to_l", i)), (f, i)
            assert np.array_equal(_png_bgr(os.path.join(idir, "overlap_%d_L.png" % i)), of.get_u8("overlap_l", i))
            assert np.array_equal(_png_bgr(os.path.join(idir, "overlap_%d_R.png" % i)), of.get_u8("overlap_r", i))
        for e, eye in enumerate(EYES):  # the pole units (TestRenderStereoPanorama.cpp:413-452)
            if os.path.exists(os.path.join(fdir, "flow_%s.bin" % eye)):
                assert _same_bits(_flow_bin(os.path.join(fdir, "flow_%s.bin" % eye)), of.get_f32("flow_pole", e)), (f, eye)
                assert np.array_equal(_png_bgr(os.path.join(idir, "extendedSideSpherical_%s.png" % eye)), of.get_u8("extended_side", e))
                assert np.array_equal(_png_bgr(os.path.join(idir, "extendedFisheyeSpherical_%s.png" % eye)), of.get_u8("extended_fisheye", e))
        if removal:  # PoleRemoval.cpp:95-126
            assert _same_bits(_flow_bin(os.path.join(fdir, "flow_bottom_secondary.bin")), of.get_f32("flow_bottom_secondary"))
            assert np.array_equal(_png_bgr(os.path.join(idir, "bottomImage.png")), of.get_u8("bottom_image"))
            assert np.array_equal(_png_bgr(os.path.join(idir, "bottomImage2.png")), of.get_u8("bottom_image2"))
    golden = json.load(open(refprog.GOLDEN))[name]
    assert refprog.digests(out, name) == golden


@pytest.mark.skipif(os.environ.get("S360_RUN_8K_REFPROGRAM") != "1", reason="minutes of CPU: set S360_RUN_8K_REFPROGRAM=1")
def test_reference_program_equals_oracle_at_8k(tmp_path):
    """BASELINE configs[2] (17 cameras of 2048x2048, eqr 8400x4096 -> 8192x8192, top + bottom): the reference program's
    stereo equirect against the oracle's, all 201 326 592 bytes. Measured in this container (8 cores): inputs 44 s,
    reference program 91 s, oracle 100 s, 0 mismatching bytes."""
    import subprocess
    rig = os.path.join(refprog.ROOT, "tests", "golden", "rig_17cam.json")
    imgs8 = refprog.frame_images(rig, 0, size=2048)
    imgs, out = str(tmp_path / "rgb"), str(tmp_path / "out")
    for cid, img in imgs8.items():
        os.makedirs(os.path.join(imgs, cid))
        Image.fromarray(np.ascontiguousarray(img[:, :, ::-1])).save(os.path.join(imgs, cid, "000000.png"), compress_level=1)
    os.makedirs(os.path.join(out, "debug", "000000", "flow_images"))
    os.makedirs(os.path.join(out, "flow", "000000"))
    eqr = os.path.join(out, "eqr.png")
    subprocess.check_call([refprog.REF_EXE, "--rig_json_file", rig, "--imgs_dir", imgs, "--frame_number", "000000",
                           "--output_data_dir", out, "--prev_frame_data_dir", "NONE", "--output_equirect_path", eqr,
                           "--eqr_width", "8400", "--eqr_height", "4096", "--final_eqr_width", "8192", "--final_eqr_height", "8192",
                           "--enable_top", "--enable_bottom", "--sharpening", "0.0"], timeout=3000)
    cams, _ = O.load_rig(rig)
    of = O.Frame(cams, O.make_params(eqr_width=8400, eqr_height=4096, final_eqr_width=8192, final_eqr_height=8192,
                                     enable_top=1, enable_bottom=1))
    side_ids, top_id, bottoms = refprog.rig_ids(rig)
    want, _ = of.render([imgs8[c] for c in side_ids], imgs8[top_id], imgs8[bottoms[0]], threaded=True)
    Image.MAX_IMAGE_PIXELS = None
    got = np.asarray(Image.open(eqr))[:, :, ::-1]
    assert got.shape == want.shape == (8192, 8192, 3) and np.array_equal(got, want)


@pytest.mark.parametrize("name", list(refprog.RAW_CASES))
def test_reference_raw2rgb_program_equals_oracle_and_golden(tmp_path, name):
    """camera_isp/Raw2Rgb.cpp (its main(), flags, 8 -> 16 bit widening, the soft CameraIsp) compiled from /root/reference
    over the stand-ins: 16-bit and 8-bit greyscale PNG in, 8- / 16-bit RGB PNG out, equal to the oracle's ISP."""
    import hashlib
    import isputil
    if not os.path.exists(refprog.REF_RAW2RGB):
        pytest.skip("oracle/_ref/Raw2Rgb not built")
    depth, flags = refprog.RAW_CASES[name]
    seen, outp = refprog.run_raw_case(refprog.REF_RAW2RGB, str(tmp_path), isputil.CONFIG_FULL, name)
    opt = lambda k, d: int(flags[flags.index(k) + 1]) if k in flags else d  # noqa: E731
    cfg = O.isp_config_from_json(isputil.CONFIG_FULL, opt("--output_bpp", 8), opt("--demosaic_filter", 2), opt("--resize", 1),
                                 int("--disable_tone_curve" in flags), opt("--black_level_offset", 0))
    got = refprog.png_pixels_bgr(outp)
    assert np.array_equal(got, O.isp_run(cfg, seen)), name
    digest = hashlib.sha256(repr((got.shape, str(got.dtype))).encode() + got.tobytes()).hexdigest()
    assert digest == json.load(open(refprog.GOLDEN))["raw2rgb"][name]
