"""Process-level drop-in: host/TestRenderStereoPanorama with the reference's flags, directory layout and state
files (TestRenderStereoPanorama.cpp:44-70, 201-255, 413-452, 961), two consecutive frames with
--prev_frame_data_dir, output PNG byte-exact against the oracle's renderStereoPanorama."""
import json
import os
import subprocess

import numpy as np
import pytest
from PIL import Image

import rigutil

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
    exe = os.path.join(ROOT, "host", "TestRenderStereoPanorama")
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


def test_bad_command_lines(tmp_path):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "-s"])
    exe = os.path.join(ROOT, "host", "TestRenderStereoPanorama")
    r = subprocess.run([exe, "--rig_json_file", "x.json"], capture_output=True, text=True)
    assert r.returncode != 0 and "missing required command line argument" in r.stderr
    r = subprocess.run([exe, "--no_such_flag", "1"], capture_output=True, text=True)
    assert r.returncode != 0 and "unknown command line flag" in r.stderr
