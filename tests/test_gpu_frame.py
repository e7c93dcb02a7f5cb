"""Full-frame parity: renderStereoPanorama (TestRenderStereoPanorama.cpp:716-972) on the GPU vs the oracle,
stage by stage, on a scaled 17-camera rig. Byte/flow results must be bit-exact."""
import ctypes as C

import numpy as np
import pytest

import rigutil
from surround360_amd import render as R

pytestmark = pytest.mark.gpu

EQR_W, EQR_H, CAM = 1008, 504, 512


@pytest.fixture(scope="module")
def setup(tmp_path_factory, rig_json, oracle, s360lib):
    d = tmp_path_factory.mktemp("rig")
    path = rigutil.scaled_rig_json(rig_json, str(d / "rig_small.json"), CAM / 2048.0)
    side, top, bottom = rigutil.frame_inputs(path, CAM)
    flags = dict(eqr_width=EQR_W, eqr_height=EQR_H, enable_top=1, enable_bottom=1, final_eqr_width=960,
                 final_eqr_height=960)
    rig = R.RigDescription(path)
    ctx = R.Context(rig, R.make_params(**flags))
    ctx.keep_intermediates(True)
    cams, _ = oracle.load_rig(path)
    of = oracle.Frame(cams, oracle.make_params(**flags))
    want, _ = of.render(side, top, bottom)
    ctx.upload_frame(side, top, bottom)
    ctx.render()
    got = ctx.download_equirect()
    yield dict(ctx=ctx, of=of, got=got, want=want, side=side, top=top, bottom=bottom, path=path, flags=flags)
    ctx.close()


def _cmp(name, got, want):
    assert got.shape == want.shape, "%s: shape %s vs %s" % (name, got.shape, want.shape)
    if got.dtype == np.float32:
        ok = np.array_equal(got.view(np.uint32), want.view(np.uint32))
        assert ok, "%s: max abs diff %g, %d mismatching values" % (name, np.abs(got - want).max(),
                                                                  int((got != want).sum()))
    else:
        d = got.astype(np.int32) - want.astype(np.int32)
        assert not d.any(), "%s: %d mismatching bytes, max |d| %d" % (name, int((d != 0).sum()), int(np.abs(d).max()))


def test_geometry(setup):
    g, of = setup["ctx"].geometry, setup["of"]
    assert (g.cam_image_width, g.cam_image_height, g.overlap_image_width, g.num_novel_views) == (
        of.cam_image_width, of.cam_image_height, of.overlap_image_width, of.num_novel_views)
    assert (g.top_rows, g.bottom_rows) == (of.top_rows, of.bottom_rows)
    assert np.float32(g.verge_at_infinity_slab_displacement) == np.float32(of.verge_disp)
    assert np.float32(g.zero_parallax_novel_view_shift_pixels) == np.float32(of.zero_parallax_shift)


def test_projections(setup):
    for i in (0, 5, 13):
        _cmp("projection %d" % i, setup["ctx"].get_u8("projection", i), setup["of"].get_u8("projection", i))


def test_overlaps_and_side_flows(setup):
    for i in (0, 7, 13):
        _cmp("overlap_l", setup["ctx"].get_u8("overlap_l", i), setup["of"].get_u8("overlap_l", i))
        _cmp("overlap_r", setup["ctx"].get_u8("overlap_r", i), setup["of"].get_u8("overlap_r", i))
        _cmp("flow_l_to_r", setup["ctx"].get_f32("flow_l_to_r", i), setup["of"].get_f32("flow_l_to_r", i))
        _cmp("flow_r_to_l", setup["ctx"].get_f32("flow_r_to_l", i), setup["of"].get_f32("flow_r_to_l", i))


def test_side_panoramas(setup):
    _cmp("side_pano_l", setup["ctx"].get_u8("side_pano_l"), setup["of"].get_u8("side_pano_l"))
    _cmp("side_pano_r", setup["ctx"].get_u8("side_pano_r"), setup["of"].get_u8("side_pano_r"))


def test_pole_projections(setup):
    _cmp("top_spherical", setup["ctx"].get_u8("top_spherical"), setup["of"].get_u8("top_spherical"))
    _cmp("bottom_spherical", setup["ctx"].get_u8("bottom_spherical"), setup["of"].get_u8("bottom_spherical"))


def test_pole_flow_inputs_and_flows(setup):
    for u in range(4):
        _cmp("extended_side %d" % u, setup["ctx"].get_u8("extended_side", u), setup["of"].get_u8("extended_side", u))
        _cmp("extended_fisheye %d" % u, setup["ctx"].get_u8("extended_fisheye", u),
             setup["of"].get_u8("extended_fisheye", u))
        _cmp("flow_pole %d" % u, setup["ctx"].get_f32("flow_pole", u), setup["of"].get_f32("flow_pole", u))


def test_pole_warped(setup):
    for u in range(4):
        _cmp("pole_warped %d" % u, setup["ctx"].get_u8("pole_warped", u), setup["of"].get_u8("pole_warped", u))


def test_eyes_and_output(setup):
    _cmp("eye_l", setup["ctx"].get_u8("eye_l"), setup["of"].get_u8("eye_l"))
    _cmp("eye_r", setup["ctx"].get_u8("eye_r"), setup["of"].get_u8("eye_r"))
    _cmp("stereo equirect", setup["got"], setup["want"])
    assert setup["got"].shape == (960, 960, 3)
    assert setup["got"].std() > 5  # not a blank image


def test_second_frame_temporal(setup, oracle):
    """--prev_frame_data_dir semantics: frame 2 regularised against frame 1's device-resident state."""
    side, top, bottom = rigutil.frame_inputs(setup["path"], CAM, yaw_deg=1.5)
    want, _ = setup["of"].render(side, top, bottom, use_prev=True)
    ctx = setup["ctx"]
    ctx.upload_frame(side, top, bottom)
    ctx.render(use_prev=True)
    _cmp("flow_l_to_r t1", ctx.get_f32("flow_l_to_r", 3), setup["of"].get_f32("flow_l_to_r", 3))
    _cmp("flow_pole t1", ctx.get_f32("flow_pole", 2), setup["of"].get_f32("flow_pole", 2))
    _cmp("frame 2", ctx.download_equirect(), want)


def test_sharded_pairs_equal_single(setup):
    """Multi-GPU partition on one device: rendering pairs in two shards fills the same strips."""
    ctx = setup["ctx"]
    rig = R.RigDescription(setup["path"])
    c2 = R.Context(rig, R.make_params(**setup["flags"]))
    c2.upload_frame(setup["side"], setup["top"], setup["bottom"])
    c2.render_pairs(0, 5)
    c2.render_pairs(5, 14)
    c2.finish(15)
    _cmp("sharded output", c2.download_equirect(), setup["got"])
    c2.close()


def test_strip_buffer_view_for_the_gather(setup):
    """parallel.strips_tensor aliases the context's strip buffers (zero copy): what the RCCL strip gather writes
    into is what s360_frame_finish assembles. Checked by zeroing one pair's strips through the torch view."""
    import torch
    from surround360_amd import parallel
    rig = R.RigDescription(setup["path"])
    c2 = R.Context(rig, R.make_params(**setup["flags"]))
    try:
        c2.upload_frame(setup["side"], setup["top"], setup["bottom"])
        c2.render_pairs(0, 14)
        c2.synchronize()
        dev = torch.device("cuda", 0)
        strips = parallel.strips_tensor(c2, dev)
        g = c2.geometry
        assert tuple(strips.shape) == (2, 14, g.cam_image_height, setup["flags"]["eqr_width"] // 14, 4)
        ptr, _ = c2.strip_ptr(0)
        assert strips.data_ptr() == ptr
        ext = torch.cuda.ExternalStream(c2.stream, device=dev)
        with torch.cuda.stream(ext):
            strips[:, 3].zero_()
        c2.finish(15)
        out = c2.download_equirect()
        ref = setup["got"]
        assert out.shape == ref.shape
        assert (out != ref).any(), "zeroing a pair's strips through the torch view must change the panorama"
        # and with nothing touched the sharded path reproduces the single-GPU frame (test_sharded_pairs_equal_single)
    finally:
        c2.close()


def test_frame_slots_batch_equals_single(setup):
    """s360_frame_render_batch: three different frames in three slots of one context — the 84 side flows in one batch
    of the flow kernels, the 12 pole flows in another — must give, slot by slot, the bytes of s360_frame_render; a
    second batch with use_prev continues three independent temporal chains."""
    rig = R.RigDescription(setup["path"])
    yaws = (0.0, 0.7, 1.4)
    f0 = [rigutil.frame_inputs(setup["path"], CAM, yaw_deg=y) for y in yaws]
    f1 = [rigutil.frame_inputs(setup["path"], CAM, yaw_deg=y + 0.3) for y in yaws]
    want = []
    for a, b in zip(f0, f1):
        c1 = R.Context(rig, R.make_params(**setup["flags"]))
        c1.upload_frame(*a)
        c1.render()
        first = c1.download_equirect()
        c1.upload_frame(*b)
        c1.render(use_prev=True)
        want.append((first, c1.download_equirect(), c1.get_f32("flow_pole", 1), c1.get_f32("flow_r_to_l", 6)))
        c1.close()
    cb = R.Context(rig, R.make_params(**setup["flags"]))
    try:
        cb.set_frame_slots(3)
        cb.set_sweep_mode("throughput")
        for k in range(3):
            cb.select_frame_slot(k)
            cb.upload_frame(*f0[k])
        cb.render_batch()
        for k in range(3):
            cb.select_frame_slot(k)
            _cmp("batched slot %d" % k, cb.download_equirect(), want[k][0])
            cb.upload_frame(*f1[k])
        cb.render_batch(use_prev=True)
        for k in range(3):
            cb.select_frame_slot(k)
            _cmp("batched temporal slot %d" % k, cb.download_equirect(), want[k][1])
            _cmp("batched temporal flow_pole slot %d" % k, cb.get_f32("flow_pole", 1), want[k][2])
            _cmp("batched temporal flow_r_to_l slot %d" % k, cb.get_f32("flow_r_to_l", 6), want[k][3])
        with pytest.raises(R.S360Error):
            cb.select_frame_slot(3)
        # a third step for slots 0 and 2 only (s360_frame_render_slots: streams of unequal length): slot 1 keeps its frame
        keep = {}
        cb.select_frame_slot(1)
        keep[1] = cb.download_equirect()
        c1 = R.Context(rig, R.make_params(**setup["flags"]))
        try:
            for k in (0, 2):
                for i, f in enumerate((f0[k], f1[k], f0[k])):
                    c1.upload_frame(*f)
                    c1.render(use_prev=i > 0)
                keep[k] = c1.download_equirect()
        finally:
            c1.close()
        for k in (0, 2):
            cb.select_frame_slot(k)
            cb.upload_frame(*f0[k])
        cb.render_slots([0, 2], use_prev=True)
        for k in range(3):
            cb.select_frame_slot(k)
            _cmp("subset step, slot %d" % k, cb.download_equirect(), keep[k])
        for bad in ([2, 0], [0, 0], [3], [-1]):
            with pytest.raises(R.S360Error):
                cb.render_slots(bad)
    finally:
        cb.close()


def test_pipelined_batches_equal_plain_batches(setup):
    """s360_set_frame_pipelining on a context of frame slots (round 6): three steps of two slots enqueued back to back — step k's
    pole stage and composite on the second stream beside step k+1's side stage, every slot chained to its own previous frame from
    the second step on — give, slot by slot and step by step, the bytes of the same batches on one stream; a step for a subset of
    the slots (s360_frame_render_slots) included."""
    rig = R.RigDescription(setup["path"])
    steps = [[rigutil.frame_inputs(setup["path"], CAM, yaw_deg=0.9 * s + 0.3 * k) for s in range(2)] for k in range(3)]
    outs = []
    for pipelined in (False, True):
        c = R.Context(rig, R.make_params(**setup["flags"]))
        try:
            c.set_frame_slots(2)
            c.set_sweep_mode("throughput")
            c.set_frame_pipelining(pipelined)
            got = []
            for k in range(3):
                for s in range(2):
                    c.select_frame_slot(s)
                    c.upload_frame(*steps[k][s])
                c.render_batch(use_prev=k > 0)  # no synchronisation between the steps
                if k == 1:  # (a fetch in the middle of the chain: the step behind it is enqueued over it)
                    c.select_frame_slot(1)
                    got.append(c.download_equirect())
            c.select_frame_slot(0)
            c.upload_frame(*steps[0][0])
            c.render_slots([0], use_prev=True)  # a fourth step for slot 0 alone
            for s in range(2):
                c.select_frame_slot(s)
                got.append(c.download_equirect())
                got.append(c.get_f32("flow_pole", 2))
                got.append(c.get_f32("flow_l_to_r", 5))
            outs.append(got)
        finally:
            c.close()
    assert len(outs[0]) == len(outs[1]) == 7
    for i, (a, b) in enumerate(zip(outs[1], outs[0])):
        _cmp("pipelined batches, item %d" % i, a, b)
    assert not np.array_equal(outs[0][1], outs[0][4])  # the two slots hold different frames


def test_frame_slots_batch_sharpened(setup):
    """--sharpening 0.25 in a batch: the IIR passes of all slots' eyes run in ONE set of launches (a frame's 2 x 4096 row
    chains alone are half a wave per SIMD); slot by slot the bytes of the frame rendered alone."""
    rig = R.RigDescription(setup["path"])
    flags = dict(setup["flags"], sharpening=0.25)
    frames = [rigutil.frame_inputs(setup["path"], CAM, yaw_deg=y) for y in (0.0, 1.1, 2.3)]
    cb = R.Context(rig, R.make_params(**flags))
    c1 = R.Context(rig, R.make_params(**flags))
    try:
        cb.set_frame_slots(3)
        cb.set_sweep_mode("throughput")
        for k in range(3):
            cb.select_frame_slot(k)
            cb.upload_frame(*frames[k])
        cb.render_batch()
        for k in range(3):
            c1.upload_frame(*frames[k])
            c1.render()
            cb.select_frame_slot(k)
            _cmp("sharpened batched slot %d" % k, cb.download_equirect(), c1.download_equirect())
        c1.set_sharpening(0.0)
        c1.render()
        assert not np.array_equal(c1.download_equirect(), cb.download_equirect())  # (the sharpening pass does something)
    finally:
        cb.close()
        c1.close()


def test_native_rccl_gather_single_rank(setup):
    """The native strip gather (comm.cpp: grouped ncclSend/ncclRecv on the context stream) on the one GPU there is:
    a one-rank communicator, the gather as a no-op between render_pairs and finish, and one real RCCL send+recv of
    the rank to itself between two strip slots (checked through the torch view of the strip buffer)."""
    import torch
    from surround360_amd import parallel
    rig = R.RigDescription(setup["path"])
    c2 = R.Context(rig, R.make_params(**setup["flags"]))
    try:
        c2.comm_init_rank(R.Context.comm_get_unique_id(), 0, 1)
        c2.upload_frame(setup["side"], setup["top"], setup["bottom"])
        c2.render_pairs(0, 14)
        c2.gather_strips([0, 14], 0)
        c2.finish(15)
        _cmp("one-rank sharded frame", c2.download_equirect(), setup["got"])
        strips = parallel.strips_tensor(c2, torch.device("cuda", 0))
        before = strips[0, 3].clone()
        assert not torch.equal(strips[0, 3], strips[0, 9])
        c2.comm_loopback(3, 9)
        c2.synchronize()
        assert torch.equal(strips[0, 9], before) and torch.equal(strips[0, 3], before)
        with pytest.raises(R.S360Error):
            c2.gather_strips([0, 7, 14], 0)  # bounds for two ranks on a one-rank communicator
        c2.comm_destroy()
        with pytest.raises(R.S360Error):
            c2.gather_strips([0, 14], 0)
    finally:
        c2.close()


def test_partitioned_temporal_state(setup, oracle):
    """s360_frame_set_partition + s360_frame_set_prev_side for a block of pairs: what a rank of the sharded frame
    does when the previous frame's state comes from files. The block's flows must equal the full frame's."""
    side, top, bottom = rigutil.frame_inputs(setup["path"], CAM, yaw_deg=1.5)
    ctx = setup["ctx"]
    ctx.upload_frame(setup["side"], setup["top"], setup["bottom"])
    ctx.render()
    prev = {i: (ctx.get_f32("flow_l_to_r", i), ctx.get_f32("flow_r_to_l", i), ctx.get_u8("overlap_l", i),
                ctx.get_u8("overlap_r", i)) for i in range(4, 9)}
    ctx.upload_frame(side, top, bottom)
    ctx.render(use_prev=True)
    want = {i: (ctx.get_f32("flow_l_to_r", i), ctx.get_f32("flow_r_to_l", i)) for i in range(4, 9)}
    rig = R.RigDescription(setup["path"])
    c2 = R.Context(rig, R.make_params(**setup["flags"]))
    try:
        c2.set_partition(4, 9)
        for i, (fl, fr, ol, orr) in prev.items():
            check = R._capi.check
            check(R.lib().s360_frame_set_prev_side(c2.h, i, R._p(fl), R._p(fr), R._p(ol), R._p(orr)), c2.h)
        with pytest.raises(R.S360Error):
            R._capi.check(R.lib().s360_frame_set_prev_side(c2.h, 2, R._p(prev[4][0]), R._p(prev[4][1]), R._p(prev[4][2]),
                                                           R._p(prev[4][3])), c2.h)
        c2.upload_frame(side, top, bottom)
        c2.render_pairs(4, 9, use_prev=True)
        for i in range(4, 9):
            _cmp("partitioned temporal flow_l_to_r %d" % i, c2.get_f32("flow_l_to_r", i), want[i][0])
            _cmp("partitioned temporal flow_r_to_l %d" % i, c2.get_f32("flow_r_to_l", i), want[i][1])
    finally:
        c2.close()


@pytest.mark.parametrize("fmt,fw,fh", [("video", 96, 80), ("photo", 64, 64)])
def test_cubemap(setup, fmt, fw, fh):
    """Stereo cubemap of the rendered frame (convertSphericalToCubemapBicubicRemap + stackOutputCubemapFaces,
    ImageWarper.cpp:95-141, CvUtil.cpp:117-138, TRSP:917-935): byte-exact, both layouts, non-square faces."""
    ctx = setup["ctx"]
    ctx.upload_frame(setup["side"], setup["top"], setup["bottom"])
    ctx.render()
    setup["of"].render(setup["side"], setup["top"], setup["bottom"])
    got = ctx.cubemap(fw, fh, fmt)
    want = setup["of"].cubemap(fw, fh, fmt)
    assert got.shape == ((4 * fh, 3 * fw, 3) if fmt == "video" else (12 * fh, fw, 3))
    _cmp("cubemap " + fmt, got, want)
    assert got.std() > 5
    with pytest.raises(R.S360Error):
        ctx.cubemap(fw, fh, "cross")


def test_pipelined_readers_wait_for_finish(setup):
    """With frame pipelining on, s360_frame_cubemap and the eye getters launch kernels that read the panoramas the
    finish stream composites: they must be ordered after it (they used to read a half-composited image)."""
    rig = R.RigDescription(setup["path"])
    ref = R.Context(rig, R.make_params(**setup["flags"]))
    pip = R.Context(rig, R.make_params(**setup["flags"]))
    try:
        pip.set_frame_pipelining(True)
        for c in (ref, pip):
            c.upload_frame(setup["side"], setup["top"], setup["bottom"])
        for _ in range(3):
            ref.render()
            want_cube, want_eye = ref.cubemap(96, 80, "video"), ref.get_u8("eye_r")
            pip.render()  # no synchronisation before the readers
            _cmp("pipelined cubemap", pip.cubemap(96, 80, "video"), want_cube)
            pip.render()
            _cmp("pipelined eye_r", pip.get_u8("eye_r"), want_eye)
    finally:
        ref.close()
        pip.close()


def test_pole_removal_two_frames(tmp_path, rig_json, oracle, s360lib):
    """--enable_pole_removal (combineBottomImagesWithPoleRemoval, PoleRemoval.cpp:32-188 + TRSP:569-634): the two
    bottom cameras merged through a flow, red masks, circle cuts and feathers; second frame with temporal state."""
    path = rigutil.scaled_rig_json(rig_json, str(tmp_path / "rig_small.json"), CAM / 2048.0)
    flags = dict(eqr_width=EQR_W, eqr_height=EQR_H, enable_top=0, enable_bottom=1, enable_pole_removal=1,
                 final_eqr_width=0, final_eqr_height=0)
    rig = R.RigDescription(path)
    ctx = R.Context(rig, R.make_params(**flags))
    cams, ids = oracle.load_rig(path)
    of = oracle.Frame(cams, oracle.make_params(**flags))
    b2 = of.bottom2_index()
    assert rig.get_bottom_camera2_id() == ids[b2] != rig.get_bottom_camera_id()
    assert np.float32(s360lib.s360_camera_usable_pixels_radius(C.byref(rig.rig[b2]))) == np.float32(of.usable_pixels_radius(b2))
    try:
        for k, yaw in enumerate((0.0, 1.5)):
            side, top, bottom, imgs, m1, m2 = rigutil.pole_removal_inputs(path, CAM, yaw_deg=yaw)
            of.set_pole_removal(imgs[ids[b2]], m1, m2)
            want, _ = of.render(side, None, bottom, use_prev=k > 0)
            ctx.upload_frame(side, None, bottom)
            ctx.upload_pole_removal(imgs[ids[b2]], m1, m2)
            ctx.render(use_prev=k > 0)
            got = ctx.download_equirect()
            _cmp("bottom_image f%d" % k, ctx.get_u8("bottom_image"), of.get_u8("bottom_image"))
            _cmp("bottom_image2 f%d" % k, ctx.get_u8("bottom_image2"), of.get_u8("bottom_image2"))
            _cmp("flow_bottom_secondary f%d" % k, ctx.get_f32("flow_bottom_secondary"), of.get_f32("flow_bottom_secondary"))
            _cmp("bottom_spherical f%d" % k, ctx.get_u8("bottom_spherical"), of.get_u8("bottom_spherical"))
            _cmp("pole removal frame %d" % k, got, want)
        assert np.abs(ctx.get_f32("flow_bottom_secondary")).max() > 0.5
    finally:
        ctx.close()


def test_frame_pipelining_equals_sequential(setup, oracle):
    """s360_set_frame_pipelining: three frames of a video stream (different inputs, temporal regularisation) enqueued
    back to back with the pole stage on the second stream must give the bytes of the one-stream sequence."""
    frames = [rigutil.frame_inputs(setup["path"], CAM, yaw_deg=0.4 * k) for k in range(3)]  # the world turns slowly
    rig = R.RigDescription(setup["path"])
    outs = []
    for pipelined in (False, True):
        ctx = R.Context(rig, R.make_params(**setup["flags"]))
        ctx.set_frame_pipelining(pipelined)
        for k, (side, top, bottom) in enumerate(frames):
            ctx.upload_frame(side, top, bottom)
            ctx.render(use_prev=k > 0)  # no synchronisation between the frames
        last = ctx.download_equirect()
        flows = [ctx.get_f32("flow_pole", u) for u in range(4)] + [ctx.get_f32("flow_l_to_r", 3)]
        outs.append((last, flows))
        ctx.close()
    _cmp("pipelined equirect", outs[1][0], outs[0][0])
    for a, b in zip(outs[1][1], outs[0][1]):
        _cmp("pipelined temporal flow", a, b)
