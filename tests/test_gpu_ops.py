"""Operator-level parity: the C-ABI entry points a maintainer binds one call site at a time —
bicubicRemapToSpherical, combineLazyNovelViews, flattenLayersDeghostPreferBase, offsetHorizontalWrap,
featherAlphaChannel, poleToSideFlowThread, sharpenThread — each on the GPU vs the oracle, bit-exact, including the
shapes the frame pipeline never produces (odd sizes, other feather radii, 3/4-channel combinations)."""
import numpy as np
import pytest

import rigutil
from surround360_amd import render as R

pytestmark = pytest.mark.gpu

EQR_W, EQR_H, CAM = 1008, 504, 512


@pytest.fixture(scope="module")
def env(tmp_path_factory, rig_json, oracle, s360lib):
    d = tmp_path_factory.mktemp("rig_ops")
    path = rigutil.scaled_rig_json(rig_json, str(d / "rig_small.json"), CAM / 2048.0)
    side, top, bottom = rigutil.frame_inputs(path, CAM)
    flags = dict(eqr_width=EQR_W, eqr_height=EQR_H, enable_top=1, enable_bottom=1, final_eqr_width=0, final_eqr_height=0)
    rig = R.RigDescription(path)
    ctx = R.Context(rig, R.make_params(**flags))
    cams, ids = oracle.load_rig(path)
    of = oracle.Frame(cams, oracle.make_params(**flags))
    of.render(side, top, bottom)  # fills the oracle's intermediates used as operator inputs below

    def side_cam(idx):  # (product s360_camera, oracle camera) of side camera idx
        c = rig.rig_side_only[idx]
        return c, cams[ids.index(c.id.decode())]

    def top_cam():
        return rig.top_camera(), cams[ids.index(rig.get_top_camera_id())]

    yield dict(ctx=ctx, of=of, rig=rig, side_cam=side_cam, top_cam=top_cam, side=side, top=top, bottom=bottom)
    ctx.close()


def _same(name, got, want):
    assert got.shape == want.shape, "%s: shape %s vs %s" % (name, got.shape, want.shape)
    if got.dtype == np.float32:
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "%s: %d mismatching values, max |d| %g" % (
            name, int((got != want).sum()), float(np.abs(got - want).max()))
    else:
        dd = got.astype(np.int32) - want.astype(np.int32)
        assert not dd.any(), "%s: %d mismatching bytes, max |d| %d" % (name, int((dd != 0).sum()), int(np.abs(dd).max()))


def _noise(rng, h, w, c):
    # smooth-ish random image: blocks of 4x4 so that bicubic taps see structure, plus per-pixel noise
    base = rng.integers(0, 256, size=((h + 3) // 4, (w + 3) // 4, c), dtype=np.uint8)
    img = np.repeat(np.repeat(base, 4, axis=0), 4, axis=1)[:h, :w].astype(np.int32)
    img += rng.integers(-20, 21, size=img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("cam_idx,sc,dc,dw,dh", [(0, 3, 4, 219, 213), (5, 3, 3, 128, 97), (5, 4, 4, 65, 9)])
def test_bicubic_remap_to_spherical(env, oracle, cam_idx, sc, dc, dw, dh):
    """ImageWarper.cpp:143-197 for a side camera: odd destination sizes (partial 64x8 tiles), 3- and 4-channel."""
    cam, ocam = env["side_cam"](cam_idx)
    src = env["side"][cam_idx]
    if sc == 4:
        src = np.dstack([src, np.full(src.shape[:2], 255, np.uint8)])
    fov = 77.8 * np.pi / 180.0
    l, r, t, b = 0.3, -0.5, fov / 2, -fov / 2
    got = env["ctx"].bicubic_remap_to_spherical(src, cam, dw, dh, dc, l, r, t, b)
    want = oracle.bicubic_remap_to_spherical(ocam, src, dw, dh, dc, l, r, t, b)
    _same("bicubicRemapToSpherical", got, want)
    assert got.any()


def test_bicubic_remap_pole_camera(env, oracle):
    """The top fisheye into the polar cap (TRSP:662-668): the source box of a destination tile is large here, so both
    the LDS-staged path and the gather fallback of the remap kernel are exercised."""
    cam, ocam = env["top_cam"]()
    got = env["ctx"].bicubic_remap_to_spherical(env["top"], cam, 300, 70, 4, 2 * np.pi, 0.0, np.pi / 2, np.pi / 2 - 0.6)
    want = oracle.bicubic_remap_to_spherical(ocam, env["top"], 300, 70, 4, 2 * np.pi, 0.0, np.pi / 2, np.pi / 2 - 0.6)
    _same("bicubicRemapToSpherical(top)", got, want)


def test_spherical_warp_map(env, oracle):
    """The warp map itself (ImageWarper.cpp:151-173). Camera::pixel runs in double on the device with the device's
    sin/cos/atan2, the oracle uses the host libm: 1e-3 px is far below the 1/32-px quantisation cv::remap applies to it
    (the remapped images of the tests above are bit-exact)."""
    cam, ocam = env["side_cam"](3)
    m = env["ctx"].spherical_warp_map(cam, 97, 61, 0.4, -0.4, 0.5, -0.5)
    want = oracle.spherical_warp_map(ocam, 97, 61, 0.4, -0.4, 0.5, -0.5)
    assert m.shape == want.shape == (61, 97, 2)
    both = np.isfinite(m) & np.isfinite(want)
    assert both.mean() > 0.5 and np.array_equal(np.isfinite(m), np.isfinite(want))
    assert np.abs(m[both] - want[both]).max() < 1e-3


@pytest.mark.parametrize("pair", [0, 7])
def test_combine_lazy_novel_views(env, pair):
    """NovelView.cpp:101-255 / TRSP:273-289 on the oracle's own overlap images and flows of one camera pair."""
    of = env["of"]
    il, ir = of.get_u8("overlap_l", pair), of.get_u8("overlap_r", pair)
    f_lr, f_rl = of.get_f32("flow_l_to_r", pair), of.get_f32("flow_r_to_l", pair)
    gl, gr = env["ctx"].combine_lazy_novel_views(il, ir, f_lr, f_rl)
    wl, wr = of.combine_lazy_novel_views(il, ir, f_lr, f_rl)
    _same("chunk L", gl, wl)
    _same("chunk R", gr, wr)
    assert gl[..., :3].any()


@pytest.mark.parametrize("w,h", [(256, 64), (333, 47), (4, 3)])
def test_flatten_layers_deghost_prefer_base(env, oracle, w, h):
    rng = np.random.default_rng(w * 1000 + h)
    base, top = _noise(rng, h, w, 4), _noise(rng, h, w, 4)
    base[..., 3] = rng.choice(np.array([0, 1, 128, 254, 255], np.uint8), size=(h, w))
    top[..., 3] = rng.choice(np.array([0, 3, 200, 255], np.uint8), size=(h, w))
    _same("flatten", env["ctx"].flatten_layers_deghost_prefer_base(base, top), oracle.flatten_layers(base, top))


@pytest.mark.parametrize("offset", [0.0, 5.0, -3.0, 170.5, -299.25])
@pytest.mark.parametrize("ch", [3, 4])
def test_offset_horizontal_wrap(env, oracle, offset, ch):
    rng = np.random.default_rng(11)
    img = _noise(rng, 37, 301, ch)
    _same("offsetHorizontalWrap", env["ctx"].offset_horizontal_wrap(img, offset), oracle.offset_horizontal_wrap(img, offset))


@pytest.mark.parametrize("w,h,e", [(256, 128, 31), (300, 77, 31), (301, 40, 31), (200, 90, 5), (64, 64, 1), (130, 66, 17)])
def test_feather_alpha_channel(env, oracle, w, h, e):
    """CvUtil.cpp:140-157: e = 31 with an even width takes the register kernels, everything else the generic tiles."""
    rng = np.random.default_rng(w + 7 * h + 13 * e)
    img = _noise(rng, h, w, 4)
    a = np.full((h, w), 255, np.uint8)
    a[: h // 5] = 0
    a[:, w - w // 7:] = 0
    a[h // 2: h // 2 + 3, w // 3: w // 3 + 5] = 0  # a hole
    a[rng.integers(0, h, 6), rng.integers(0, w, 6)] = rng.integers(0, 255, 6).astype(np.uint8)
    img[..., 3] = a
    _same("featherAlphaChannel e=%d" % e, env["ctx"].feather_alpha_channel(img, e), oracle.feather_alpha_channel(img, e))


@pytest.mark.parametrize("amount,h,w", [(0.25, 96, 200), (1.0, 96, 200), (0.25, 131, 203), (0.25, 70, 61)])
def test_sharpen(env, oracle, amount, h, w):
    """iirLowPass + sharpenWithIirLowPass (Filter.h:40-127); sizes that are not multiples of the kernels' 64-position
    tiles, 8-step groups and 16-chain waves."""
    rng = np.random.default_rng(5)
    img = _noise(rng, h, w, 3)
    _same("sharpen", env["ctx"].sharpen(img, amount), oracle.sharpen(img, amount))


def test_pole_to_side_flow(env):
    """poleToSideFlowThread (TRSP:388-561) as one call: feather + extend, DOWN-hinted flow, ramped warp, crop."""
    of = env["of"]
    side = of.get_u8("side_pano_l")
    pole = of.get_u8("top_spherical")
    got, gflow = env["ctx"].pole_to_side_flow(side, pole, want_flow=True)
    want, wflow = of.pole_to_side_flow(side, pole, want_flow=True)
    _same("pole flow", gflow, wflow)
    _same("warped pole", got, want)


def test_operator_argument_errors(env):
    ctx = env["ctx"]
    cam, _ = env["side_cam"](0)
    with pytest.raises(R.S360Error):
        ctx.bicubic_remap_to_spherical(np.zeros((8, 8, 4), np.uint8), cam, 16, 16, 3, 0.1, -0.1, 0.1, -0.1)  # 4 -> 3 channels
    with pytest.raises(R.S360Error):
        ctx.feather_alpha_channel(np.zeros((8, 8, 4), np.uint8), 41)  # erode size above the supported range
    with pytest.raises(R.S360Error):
        ctx.feather_alpha_channel(np.zeros((8, 8, 4), np.uint8), 16)  # even: GaussianBlur would refuse the kernel size


def test_frame_calls_refuse_sizes_no_frame_can_have(rig_json, s360lib):
    """Flags whose frame cannot exist — negative, zero or absurd eqr / final sizes, a projection, overlap or strip that
    would be empty, flow images below 2 x 2 after the entry downscale: the context is created (its operator-level entry
    points do not depend on the frame geometry) and every s360_frame_* call fails with S360_ERR_INVALID_ARG, where the
    reference runs into OpenCV assertions inside the frame (TRSP aborts). Before these checks the first render of such
    a context wrote outside its buffers."""
    from surround360_amd import synth
    rig = R.RigDescription(rig_json)
    img = np.zeros((16, 16, 3), np.uint8)
    i0, i1 = synth.flow_pair(24, 20, seed=3)
    for kw in (dict(eqr_width=-14, eqr_height=64), dict(eqr_width=0, eqr_height=64), dict(eqr_width=28, eqr_height=0),
               dict(eqr_width=28, eqr_height=-5), dict(eqr_width=1400000, eqr_height=64), dict(eqr_width=140, eqr_height=2000000000),
               dict(eqr_width=28, eqr_height=2),    # cam_image_height 0
               dict(eqr_width=14, eqr_height=64),   # overlap 2 px wide: 1 px after the x0.5 entry
               dict(eqr_width=140, eqr_height=5),   # 2 rows
               dict(eqr_width=140, eqr_height=64, final_eqr_width=-1, final_eqr_height=64),
               dict(eqr_width=140, eqr_height=64, final_eqr_width=64, final_eqr_height=-1),
               dict(eqr_width=140, eqr_height=64, final_eqr_width=64, final_eqr_height=1),
               dict(eqr_width=140, eqr_height=64, final_eqr_width=70000, final_eqr_height=64)):
        ctx = R.Context(rig, R.make_params(**kw))
        try:
            for call in (lambda: ctx.upload_frame([img]), lambda: ctx.render(), lambda: ctx.render_pairs(0, 2, False),
                         lambda: ctx.set_frame_slots(2), lambda: ctx.equirect_dev()):
                with pytest.raises(R.S360Error, match="eqr"):
                    call()
            assert ctx.compute_optical_flow(i0, i1, "pixflow_low", "LEFT").shape == (20, 24, 2)  # operators still work
        finally:
            ctx.close()
    ctx = R.Context(rig, R.make_params(eqr_width=28, eqr_height=14))  # the smallest frame of this rig: overlap 4 x 6
    g = ctx.geometry
    assert (g.cam_image_width, g.cam_image_height, g.overlap_image_width, g.num_novel_views) == (6, 6, 4, 2)
    ctx.close()


@pytest.mark.parametrize("eqr_w,eqr_h,cam,final,poles", [(28, 14, 64, (0, 0), 1), (42, 21, 64, (0, 0), 1), (70, 33, 64, (0, 0), 0),
                                                         (28, 300, 64, (0, 0), 1), (28, 14, 8, (3, 2), 1), (140, 70, 4, (300, 300), 1),
                                                         (504, 252, 96, (0, 0), 1)])  # (pole projections that minify by 7)
def test_tiny_frames_equal_the_oracle(tmp_path, rig_json, oracle, s360lib, eqr_w, eqr_h, cam, final, poles):
    """Whole frames at the small end of what s360_create accepts — overlap images 4 px wide, 6-row projections, 4 x 4
    cameras magnified 35 times, a 3 x 2 output, feathers as large as the images allow: every pyramid is a single level
    and every tile kernel runs partial tiles only. Stereo equirect byte for byte against the oracle."""
    path = rigutil.scaled_rig_json(rig_json, str(tmp_path / "rig_tiny.json"), cam / 2048.0)
    side, top, bottom = rigutil.frame_inputs(path, cam)
    flags = dict(eqr_width=eqr_w, eqr_height=eqr_h, enable_top=poles, enable_bottom=poles, final_eqr_width=final[0],
                 final_eqr_height=final[1], sharpening=0.25 if poles else 0.0, side_alpha_feather_size=min(7, cam // 2),
                 std_alpha_feather_size=3)
    cams, _ = oracle.load_rig(path)
    want, _ = oracle.Frame(cams, oracle.make_params(**flags)).render(side, top, bottom)
    ctx = R.Context(R.RigDescription(path), R.make_params(**flags))
    try:
        ctx.upload_frame(side, top, bottom)
        ctx.render()
        _same("tiny frame %dx%d" % (eqr_w, eqr_h), ctx.download_equirect(), want)
    finally:
        ctx.close()
