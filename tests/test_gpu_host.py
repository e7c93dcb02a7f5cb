"""Process-level drop-in: host/TestRenderStereoPanorama with the reference's flags, directory layout and state
files (TestRenderStereoPanorama.cpp:44-70, 201-255, 413-452, 961), two consecutive frames with
--prev_frame_data_dir, output PNG byte-exact against the oracle's renderStereoPanorama."""
import json
import os
import subprocess

import numpy as np
import pytest

import refprog
from PIL import Image

import rigutil

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (developer switch of tests/conftest.py: with S360_TEST_EMULATED_LIB=1 the programs linked against the emulated library)
HOST_DIR = os.path.join(ROOT, "tools", "emu") if os.environ.get("S360_TEST_EMULATED_LIB") == "1" else os.path.join(ROOT, "host")
EQR_W, EQR_H, CAM = 1008, 504, 512


def _write_frame(imgs_dir, rig_path, frame, side, top, bottom):
    cams = json.load(open(rig_path))["cameras"]
    side_ids = [c["id"] for c in cams if "side" in c.get("group", "")]
    other = [c for c in cams if "side" not in c.get("group", "")]
    top_id = max(other, key=lambda c: c["forward"][2])["id"]
    bot_id = min(other, key=lambda c: c["forward"][2])["id"]
    for cid, img in list(zip(side_ids, side)) + [(top_id, top), (bot_id, bottom)]:
        d = os.path.join(imgs_dir, cid)
        os.makedirs(d, exist_ok=True)
        Image.fromarray(np.ascontiguousarray(img[:, :, ::-1])).save(os.path.join(d, frame + ".png"))  # BGR -> RGB


def test_two_frames_through_the_binary(tmp_path, rig_json, oracle, s360lib):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])
    exe = os.path.join(HOST_DIR, "TestRenderStereoPanorama")
    rig_path = rigutil.scaled_rig_json(rig_json, str(tmp_path / "rig_small.json"), CAM / 2048.0)
    imgs, out = str(tmp_path / "rgb"), str(tmp_path / "out")
    os.makedirs(out)
    frames = {"000000": rigutil.frame_inputs(rig_path, CAM), "000001": rigutil.frame_inputs(rig_path, CAM, yaw_deg=1.5)}
    for f, (side, top, bottom) in frames.items():
        _write_frame(imgs, rig_path, f, side, top, bottom)
    flags = dict(eqr_width=EQR_W, eqr_height=EQR_H, enable_top=1, enable_bottom=1, final_eqr_width=960,
                 final_eqr_height=960)
    cams, _ = oracle.load_rig(rig_path)
    of = oracle.Frame(cams, oracle.make_params(**flags))
    prev = "NONE"
    for f in ("000000", "000001"):
        eqr = os.path.join(out, "eqr_%s.png" % f)
        cmd = [exe, "--rig_json_file", rig_path, "--imgs_dir=" + imgs, "--frame_number", f, "--output_data_dir", out,
               "--prev_frame_data_dir", prev, "--output_equirect_path", eqr, "--enable_top", "--enable_bottom",
               "--eqr_width", str(EQR_W), "--eqr_height", str(EQR_H), "--final_eqr_width", "960",
               "--final_eqr_height=960", "--side_flow_alg", "pixflow_low", "--polar_flow_alg", "pixflow_low",
               "--sharpening", "0.0", "--logbuflevel", "-1", "--stderrthreshold", "0", "--v", "1",
               "--log_dir", out]  # the glog flags batch_process_video.py passes
        cube = os.path.join(out, "cube_%s.png" % f)
        if f == "000001":
            cmd += ["--output_cubemap_path", cube, "--cubemap_width", "96", "--cubemap_height", "96",
                    "--cubemap_format", "video"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert "Runtime breakdown" in r.stderr
        got = np.asarray(Image.open(eqr))[:, :, ::-1]  # RGB -> BGR
        side, top, bottom = frames[f]
        want, _ = of.render(side, top, bottom, use_prev=(prev != "NONE"))
        assert got.shape == want.shape == (960, 960, 3)
        d = got.astype(np.int32) - want.astype(np.int32)
        assert not d.any(), "frame %s: %d mismatching bytes" % (f, int((d != 0).sum()))
        if f == "000001":  # TRSP:917-935
            got_c = np.asarray(Image.open(cube))[:, :, ::-1]
            want_c = of.cubemap(96, 96, "video")
            assert got_c.shape == want_c.shape == (384, 288, 3) and np.array_equal(got_c, want_c)
        # the state files the next frame (or a --resume) needs, under the reference's names
        for i in (0, 13):
            assert os.path.exists(os.path.join(out, "flow", f, "flowLtoR_%d.bin" % i))
            assert os.path.exists(os.path.join(out, "flow", f, "flowRtoL_%d.bin" % i))
            assert os.path.exists(os.path.join(out, "debug", f, "flow_images", "overlap_%d_L.png" % i))
        for eye in ("top_left", "top_right", "bottom_left", "bottom_right"):
            assert os.path.exists(os.path.join(out, "flow", f, "flow_%s.bin" % eye))
            assert os.path.exists(os.path.join(out, "debug", f, "flow_images", "extendedSideSpherical_%s.png" % eye))
            assert os.path.exists(os.path.join(out, "debug", f, "flow_images", "extendedFisheyeSpherical_%s.png" % eye))
        # flow .bin is the reference's format: int32 rows, int32 cols, (fx, fy) float32 pairs (CvUtil.cpp:159-199)
        raw = np.fromfile(os.path.join(out, "flow", f, "flowLtoR_0.bin"), dtype=np.int32, count=2)
        assert (raw[0], raw[1]) == (of.cam_image_height, of.overlap_image_width)
        prev = f


def test_stream_mode_equals_chained_processes(tmp_path, rig_json, oracle, s360lib):
    """--num_frames 3: one process renders three consecutive frames as a stream (device-resident temporal state, frame
    pipelining, next frame decoded/uploaded and previous frame downloaded/encoded while the current one renders). Every
    equirect must equal the oracle's frame-by-frame chain, and the state written after the last frame must let a
    classic one-frame process continue the chain (--prev_frame_data_dir)."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])
    exe = os.path.join(HOST_DIR, "TestRenderStereoPanorama")
    rig_path = rigutil.scaled_rig_json(rig_json, str(tmp_path / "rig_small.json"), CAM / 2048.0)
    imgs, out = str(tmp_path / "rgb"), str(tmp_path / "out")
    os.makedirs(out)
    names = ["000007", "000008", "000009", "000010"]
    frames = {f: rigutil.frame_inputs(rig_path, CAM, yaw_deg=0.6 * k) for k, f in enumerate(names)}
    for f, (side, top, bottom) in frames.items():
        _write_frame(imgs, rig_path, f, side, top, bottom)
    flags = dict(eqr_width=EQR_W, eqr_height=EQR_H, enable_top=1, enable_bottom=1, final_eqr_width=960,
                 final_eqr_height=960, sharpening=0.25)
    cams, _ = oracle.load_rig(rig_path)
    of = oracle.Frame(cams, oracle.make_params(**flags))
    common = ["--rig_json_file", rig_path, "--imgs_dir", imgs, "--output_data_dir", out, "--enable_top", "--enable_bottom",
              "--eqr_width", str(EQR_W), "--eqr_height", str(EQR_H), "--final_eqr_width", "960", "--final_eqr_height", "960",
              "--sharpening", "0.25", "--v", "1"]
    r = subprocess.run([exe, "--frame_number", names[0], "--num_frames", "3", "--output_equirect_path",
                        os.path.join(out, "eqr_%s.png")] + common, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "stream of 3 frames" in r.stderr
    for k, f in enumerate(names[:3]):
        want, _ = of.render(*frames[f], use_prev=k > 0)
        got = np.asarray(Image.open(os.path.join(out, "eqr_%s.png" % f)))[:, :, ::-1]
        assert got.shape == want.shape and np.array_equal(got, want), "stream frame %s differs" % f
    assert os.path.exists(os.path.join(out, "flow", names[2], "flow_top_left.bin"))
    assert not os.path.exists(os.path.join(out, "flow", names[1]))  # state only after the last frame of the stream
    # a one-frame process resumes from the stream's last frame
    eqr = os.path.join(out, "resume.png")
    r = subprocess.run([exe, "--frame_number", names[3], "--prev_frame_data_dir", names[2], "--output_equirect_path", eqr] +
                       common, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    want, _ = of.render(*frames[names[3]], use_prev=True)
    assert np.array_equal(np.asarray(Image.open(eqr))[:, :, ::-1], want)
    # --num_gpus beyond the box's GPUs is refused, not silently reduced
    r = subprocess.run([exe, "--frame_number", names[0], "--num_gpus", "2", "--output_equirect_path", eqr] + common,
                       capture_output=True, text=True)
    import ctypes
    if s360lib.s360_device_count() < 2:
        assert r.returncode != 0 and "not that many HIP devices" in r.stderr
    else:
        assert r.returncode == 0, r.stderr
        want0, _ = oracle.Frame(cams, oracle.make_params(**flags)).render(*frames[names[0]])
        assert np.array_equal(np.asarray(Image.open(eqr))[:, :, ::-1], want0)


def test_bad_command_lines(tmp_path):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])
    exe = os.path.join(HOST_DIR, "TestRenderStereoPanorama")
    r = subprocess.run([exe, "--rig_json_file", "x.json"], capture_output=True, text=True)
    assert r.returncode != 0 and "missing required command line argument" in r.stderr
    r = subprocess.run([exe, "--no_such_flag", "1"], capture_output=True, text=True)
    assert r.returncode != 0 and "unknown command line flag" in r.stderr


def test_pole_removal_through_the_binary(tmp_path, rig_json, oracle, s360lib):
    """--enable_pole_removal --bottom_pole_masks_dir: two frames, the second regularised against the first one's
    flow_bottom_secondary.bin / bottomImage{,2}.png (PoleRemoval.cpp:95-126), output byte-exact vs the oracle."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])
    exe = os.path.join(HOST_DIR, "TestRenderStereoPanorama")
    rig_path = rigutil.scaled_rig_json(rig_json, str(tmp_path / "rig_small.json"), CAM / 2048.0)
    imgs_dir, out, masks = str(tmp_path / "rgb"), str(tmp_path / "out"), str(tmp_path / "masks")
    os.makedirs(out)
    os.makedirs(masks)
    flags = dict(eqr_width=EQR_W, eqr_height=EQR_H, enable_top=0, enable_bottom=1, enable_pole_removal=1,
                 final_eqr_width=0, final_eqr_height=0)
    cams, ids = oracle.load_rig(rig_path)
    of = oracle.Frame(cams, oracle.make_params(**flags))
    b2id = ids[of.bottom2_index()]
    cj = json.load(open(rig_path))["cameras"]
    other = [c for c in cj if "side" not in c.get("group", "")]
    bot_id = [c["id"] for c in other if c["id"] != b2id and c["forward"][2] < 0][0]
    prev = "NONE"
    for f, yaw in (("000000", 0.0), ("000001", 1.5)):
        side, top, bottom, imgs, m1, m2 = rigutil.pole_removal_inputs(rig_path, CAM, yaw_deg=yaw)
        _write_frame(imgs_dir, rig_path, f, side, top, bottom)
        d = os.path.join(imgs_dir, b2id)
        os.makedirs(d, exist_ok=True)
        Image.fromarray(np.ascontiguousarray(imgs[b2id][:, :, ::-1])).save(os.path.join(d, f + ".png"))
        Image.fromarray(np.ascontiguousarray(m1[:, :, ::-1])).save(os.path.join(masks, bot_id + ".png"))
        Image.fromarray(np.ascontiguousarray(m2[:, :, ::-1])).save(os.path.join(masks, b2id + ".png"))
        eqr = os.path.join(out, "eqr_%s.png" % f)
        cmd = [exe, "--rig_json_file", rig_path, "--imgs_dir", imgs_dir, "--frame_number", f, "--output_data_dir", out,
               "--prev_frame_data_dir", prev, "--output_equirect_path", eqr, "--enable_bottom", "--enable_pole_removal",
               "--bottom_pole_masks_dir", masks, "--eqr_width", str(EQR_W), "--eqr_height", str(EQR_H),
               "--final_eqr_width", "0", "--final_eqr_height", "0", "--poleremoval_flow_alg", "pixflow_low"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        of.set_pole_removal(imgs[b2id], m1, m2)
        want, _ = of.render(side, None, bottom, use_prev=(prev != "NONE"))
        got = np.asarray(Image.open(eqr))[:, :, ::-1]
        assert got.shape == want.shape and np.array_equal(got, want), "frame %s differs" % f
        for name in ("flow/%s/flow_bottom_secondary.bin", "debug/%s/flow_images/bottomImage.png",
                     "debug/%s/flow_images/bottomImage2.png"):
            assert os.path.exists(os.path.join(out, name % f))
        prev = f
    # --enable_pole_removal without masks is an error, like requireArg at TRSP:571
    r = subprocess.run([c for c in cmd if c not in ("--bottom_pole_masks_dir", masks)], capture_output=True, text=True)
    assert r.returncode != 0 and "bottom_pole_masks_dir" in r.stderr


def test_optical_flow_harness(tmp_path, oracle, s360lib):
    """host/TestOpticalFlow --mode test (TestOpticalFlow.cpp:50-143): both flow directions of one pair, bit-exact
    against the oracle, "RUNTIME (sec)" logged per repetition, flows written in the reference's .bin format."""
    from surround360_amd import synth, render as R
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])
    exe = os.path.join(HOST_DIR, "TestOpticalFlow")
    i0, i1 = synth.flow_pair(300, 260, seed=11)
    Image.fromarray(np.ascontiguousarray(i0[:, :, [2, 1, 0, 3]])).save(str(tmp_path / "left.png"))   # BGRA -> RGBA
    Image.fromarray(np.ascontiguousarray(i1[:, :, [2, 1, 0]])).save(str(tmp_path / "right.png"))     # no alpha: 255 added
    i1 = i1.copy()
    i1[:, :, 3] = 255
    r = subprocess.run([exe, "--mode", "test", "--test_dir", str(tmp_path), "--left_img", "left.png", "--right_img",
                        "right.png", "--flow_alg", "pixflow_low", "--repetitions", "2"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stderr.count("RUNTIME (sec) = ") == 2
    for name, a, b, hint in (("flowLtoR", i0, i1, "LEFT"), ("flowRtoL", i1, i0, "RIGHT")):
        got = R.read_flow_from_file(str(tmp_path / "disparity" / (name + "_pixflow_low.bin")))
        want = oracle.compute_optical_flow(a, b, "pixflow_low", hint)
        assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32)), name


def test_raw2rgb_binary(tmp_path, oracle, s360lib):
    """host/Raw2Rgb (Raw2Rgb.cpp, soft-ISP path): 16-bit greyscale PNG and headerless .raw inputs, 8- and 16-bit PNG
    outputs, bit-exact against the oracle; "Runtime = ... ms" logged."""
    import struct
    import zlib
    import isputil
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])
    exe = os.path.join(HOST_DIR, "Raw2Rgb")
    w, h = 160, 96
    raw = isputil.bayer_frame(w, h, seed=4)
    cfgj = json.loads(isputil.CONFIG_FULL)
    cfgj["CameraIsp"]["width"], cfgj["CameraIsp"]["height"] = w, h
    cfg_path = str(tmp_path / "isp.json")
    open(cfg_path, "w").write(json.dumps(cfgj))
    Image.fromarray(raw, "I;16").save(str(tmp_path / "in.png"))
    raw.tofile(str(tmp_path / "in.raw"))

    def png16(path):
        data = open(path, "rb").read()
        pos, idat, ihdr = 8, b"", None
        while pos < len(data):
            n, t = struct.unpack(">I4s", data[pos:pos + 8])
            if t == b"IHDR":
                ihdr = struct.unpack(">IIBBBBB", data[pos + 8:pos + 8 + n])
            elif t == b"IDAT":
                idat += data[pos + 8:pos + 8 + n]
            pos += 12 + n
        ww, hh = ihdr[:2]
        rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(hh, 1 + ww * 6)
        rows = refprog.png_unfilter(rows, 6).reshape(hh, ww, 3, 2).astype(np.uint16)  # (the writer filters its scanlines)
        return ((rows[..., 0] << 8) | rows[..., 1])[..., ::-1]  # RGB -> BGR

    for inp, bpp, dm, extra in (("in.png", 16, 2, []), ("in.raw", 8, 0, ["--disable_tone_curve"]),
                                ("in.png", 8, 2, ["--resize", "2", "--black_level_offset=20"])):
        out = str(tmp_path / ("out_%s_%d.png" % (inp[-3:], bpp)))
        r = subprocess.run([exe, "--input_image_path", str(tmp_path / inp), "--output_image_path", out,
                            "--isp_config_path", cfg_path, "--output_bpp", str(bpp), "--demosaic_filter", str(dm)] + extra,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert "Runtime = " in r.stderr
        rs = 2 if "--resize" in extra else 1
        want = oracle.isp_run(oracle.isp_config_from_json(json.dumps(cfgj), bpp, dm, rs, int("--disable_tone_curve" in extra),
                                                          20 if rs == 2 else 0), raw)
        got = png16(out) if bpp == 16 else np.asarray(Image.open(out))[:, :, ::-1]
        assert np.array_equal(got, want), (inp, bpp)
    # --accelerate [--fast]: CameraIspPipe's arithmetic (Raw2Rgb.cpp:427-440) against its restatement (oracle/isp_pipe.h, not pinned)
    for bpp, fast, extra in ((16, False, []), (8, True, ["--black_level_offset=20"])):
        out = str(tmp_path / ("out_acc_%d.png" % bpp))
        r = subprocess.run([exe, "--input_image_path", str(tmp_path / "in.png"), "--output_image_path", out, "--isp_config_path",
                            cfg_path, "--output_bpp", str(bpp), "--accelerate"] + (["--fast"] if fast else []) + extra,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        want = oracle.isp_pipe_run(oracle.isp_config_from_json(json.dumps(cfgj), bpp, 2, 1, 0, 20 if extra else 0), raw, fast=fast)
        got = png16(out) if bpp == 16 else np.asarray(Image.open(out))[:, :, ::-1]
        assert np.array_equal(got, want), ("accelerate", bpp)
    r = subprocess.run([exe, "--input_image_path", str(tmp_path / "in.png"), "--output_image_path", str(tmp_path / "x.png"),
                        "--isp_config_path", cfg_path, "--accelerate", "--resize", "2"], capture_output=True, text=True)
    assert r.returncode != 0 and "resize" in r.stderr
