"""The whole product library without a GPU. tools/libs360_emu.so is every source of surround360_amd/csrc — the C ABI,
the frame pipeline, the flow engine, the ISP, all HIP kernels — compiled for the CPU over tools/hip_wave_shim (workgroups
as OS threads, lanes as coroutines, DPP / ballot / shuffles / barriers as rendezvous; kernels run to completion at
launch, copies are memcpy), and tools/emu/* are the host programs of host/ linked against it. They get the same
command lines as the real programs get on the GPU box (tests/test_gpu_zz_refprogram.py, test_gpu_zz_unpacker.py,
test_gpu_host.py) and must write the same files: here that is checked against the REFERENCE'S OWN PROGRAMS' digests
(tests/golden/refprogram_golden.json) and against the oracle.

What this covers: the logic of the HIP sources — indexing, tiling, batching, hand-offs, state handling, file formats.
What it cannot: arithmetic that the device does differently from the host (its libm for double exp / atan2, v_sqrt,
denormal handling) and anything about timing; those are what the -m gpu tests are for. It is NOT a fallback: the product
never loads this library (surround360_amd/_capi.py loads libs360.so or fails)."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest
from PIL import Image

import oracle_lib as O
import refprog
import rigutil

ROOT = refprog.ROOT
EMU = os.path.join(ROOT, "tools", "emu")


@pytest.fixture(scope="module")
def emu_programs():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools"), "-s", "libs360_emu.so", "emu_programs"])
    return EMU


@pytest.mark.parametrize("name", list(refprog.CASES))
def test_emulated_program_writes_what_the_reference_program_writes(tmp_path, emu_programs, name):
    """Two chained frames; sharpening + cubemap + pixflow_search_20; three chained frames sharpened; pole removal: stereo equirects,
    cubemap, 28 + 4 flows per frame, overlap / pole / pole-removal state images — 551 files over the four cases, digest for digest."""
    rig = rigutil.scaled_rig_json(os.path.join(ROOT, "tests", "golden", "rig_17cam.json"), str(tmp_path / "rig_small.json"),
                                  refprog.CAM / 2048.0)
    out = refprog.run_case(os.path.join(emu_programs, "TestRenderStereoPanorama"), str(tmp_path), rig, name)
    got = refprog.digests(out, name)
    golden = json.load(open(refprog.GOLDEN))[name]
    assert sorted(got) == sorted(golden)
    differing = sorted(k for k in golden if got[k] != golden[k])
    assert not differing, "%d of %d files differ from the reference program's: %s" % (len(differing), len(golden), differing[:12])


@pytest.mark.parametrize("exe_name,name,flags", [
    ("TestRenderStereoPanorama_ops_hip_emu", "sharpen_cubemap_search", ["--side_flow_alg", "pixflow_search_20_hip", "--polar_flow_alg", "pixflow_low_hip"]),
    ("TestRenderStereoPanorama_ops_hip_emu", "pole_removal", ["--side_flow_alg", "pixflow_low_hip", "--polar_flow_alg", "pixflow_low_hip",
                                                              "--poleremoval_flow_alg", "pixflow_low_hip"])])
def test_reference_program_with_the_integration_binding(tmp_path, emu_programs, exe_name, name, flags):
    """INTEGRATION.md section 1, executed. The REFERENCE'S OWN TestRenderStereoPanorama — its sources where they lie under
    /root/reference, compiled over the stand-ins of oracle/ref_shim — with the `PixFlowHip` subclass of its
    OpticalFlowInterface and the two extra names in its flow factory injected from oracle/ref_binding/ (nothing of the
    reference is modified or copied), linked against the library (here: its sources on the CPU emulation). With
    `--side_flow_alg pixflow_low_hip --polar_flow_alg pixflow_low_hip` every flow of the reference's frame — its 14 pair
    threads and 4 pole threads calling one shared context, temporal regularisation through the reference's own state
    files — is computed by libs360, and the program writes exactly the files the unmodified reference program writes.
    The `_ops_` program goes further down INTEGRATION.md's table: the reference's bicubicRemapToSpherical,
    flattenLayersDeghostPreferBase, offsetHorizontalWrap and featherAlphaChannel are replaced as well (its ImageWarper.cpp /
    CvUtil.cpp compiled with those four renamed, oracle/ref_binding/operators_hip.cpp in their place): projections, blends,
    shifts, feathers and flows of the reference's frame all run through the C ABI."""
    exe = os.path.join(ROOT, "oracle", "_ref", exe_name)
    if os.path.isdir("/root/reference/surround360_render/source"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "_ref/" + exe_name])
    elif not os.path.exists(exe):
        pytest.skip("needs /root/reference to build the reference program from (make -C oracle ref_binding ref_binding_ops)")
    rig = rigutil.scaled_rig_json(os.path.join(ROOT, "tests", "golden", "rig_17cam.json"), str(tmp_path / "rig_small.json"),
                                  refprog.CAM / 2048.0)
    out = refprog.run_case(exe, str(tmp_path), rig, name, more_args=flags)
    got = refprog.digests(out, name)
    golden = json.load(open(refprog.GOLDEN))[name]
    assert sorted(got) == sorted(golden)
    differing = sorted(k for k in golden if got[k] != golden[k])
    assert not differing, "%d of %d files differ from the unmodified reference program's: %s" % (len(differing), len(golden), differing[:12])


@pytest.mark.parametrize("name", list(refprog.RAW_CASES))
def test_reference_raw2rgb_with_the_integration_binding(tmp_path, emu_programs, name):
    """INTEGRATION.md section 3, executed: the reference's own Raw2Rgb program with the `CameraIspGpu` subclass of its
    CameraIsp (oracle/ref_binding/CameraIspGpu.h) force-included and used in place of the base class — its flags, its JSON
    reader, its PNG input and output — develops the image through s360_isp_* (here: the library's CPU emulation) and
    writes the pixels the unmodified reference program writes."""
    import hashlib
    import isputil
    exe = os.path.join(ROOT, "oracle", "_ref", "Raw2Rgb_hip_emu")
    if os.path.isdir("/root/reference/surround360_render/source"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "_ref/Raw2Rgb_hip_emu"])
    elif not os.path.exists(exe):
        pytest.skip("needs /root/reference to build the reference program from (make -C oracle ref_binding_isp)")
    _, outp = refprog.run_raw_case(exe, str(tmp_path), isputil.CONFIG_FULL, name)
    a = refprog.png_pixels_bgr(outp)
    digest = hashlib.sha256(repr((a.shape, str(a.dtype))).encode() + a.tobytes()).hexdigest()
    assert digest == json.load(open(refprog.GOLDEN))["raw2rgb"][name]


def _build_ref(target):
    exe = os.path.join(ROOT, "oracle", "_ref", target)
    if os.path.isdir("/root/reference/surround360_render/source"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "_ref/" + target])
    elif not os.path.exists(exe):
        pytest.skip("needs /root/reference to build the reference program from (make -C oracle ref_binding_pipe)")
    return exe


def check_pipe_raw_case(exe, host_exe, work, name):
    """The reference's Raw2Rgb --accelerate on the library == host/Raw2Rgb --accelerate == the oracle's restatement."""
    import isputil
    depth, flags = refprog.PIPE_RAW_CASES[name]
    seen, outp = refprog.run_raw_case(exe, os.path.join(work, "ref"), isputil.CONFIG_FULL, name)
    a = refprog.png_pixels_bgr(outp)
    _, outh = refprog.run_raw_case(host_exe, os.path.join(work, "host"), isputil.CONFIG_FULL, name)
    assert np.array_equal(a, refprog.png_pixels_bgr(outh)), name
    bpp = int(flags[flags.index("--output_bpp") + 1])
    cfg = O.isp_config_from_json(isputil.CONFIG_FULL, bpp, 2, 1, int("--disable_tone_curve" in flags),
                                 20 if "--black_level_offset" in flags else 0)
    want = O.isp_pipe_run(cfg, seen, fast="--fast" in flags)
    assert a.shape == want.shape and a.dtype == want.dtype and np.array_equal(a, want), name


@pytest.mark.parametrize("name", list(refprog.PIPE_RAW_CASES))
def test_reference_raw2rgb_accelerated_on_the_library(tmp_path, emu_programs, name):
    """The reference's ACCELERATED ISP with nothing of it changed: Raw2Rgb.cpp (-DUSE_HALIDE) and CameraIspPipe.h compiled where
    they lie, the four functions Halide would generate (CameraIspGen8 / 16 / Fast8 / Fast16) provided by
    oracle/ref_binding/halide_shim over s360_isp_pipe_generated. The reference's own code builds the vignette tables, the tone
    table, the CCM and the parameter list; the library (here: its CPU emulation) only runs the pipeline. The picture equals
    host/Raw2Rgb --accelerate's and the oracle's — which pins everything of the accelerated path EXCEPT the generated
    arithmetic itself (oracle/isp_pipe.h says what that leaves open)."""
    check_pipe_raw_case(_build_ref("Raw2Rgb_pipe_hip_emu"), os.path.join(emu_programs, "Raw2Rgb"), str(tmp_path), name)


@pytest.mark.parametrize("bits", [12, 8])
def test_reference_unpacker_on_the_library(tmp_path, emu_programs, bits):
    """The reference's own Unpacker (Unpacker.cpp, BinaryFootageFile.cpp, RawConverter.cpp, CameraIspPipe.h, unmodified) over the
    same four functions: container parsing, 12-bit unpacking, one CameraIspPipe per frame, 16-bit PNGs, camN renaming — the
    files host/Unpacker writes and the oracle's pixels."""
    from test_gpu_zz_unpacker import check_unpacker
    check_unpacker(_build_ref("Unpacker_hip_emu"), tmp_path, O, bits, soft=False)


@pytest.mark.parametrize("script,seed,cases,ok", [("random_flow.py", 11, 12, "0 differ"), ("random_ops.py", 11, 60, "0 differ"),
                                                  ("random_isp.py", 11, 40, "0 differ")])
def test_random_calls_equal_the_oracle(emu_programs, script, seed, cases, ok):
    """A seeded slice of the randomised differential campaigns of tools/fuzz (DESIGN.md section 2 has the full counts): flows of
    random sizes / masks / hints / algorithms / previous-frame state through both sweep kernels, the other operator entry
    points with random shapes and arguments, the ISP with random configurations — emulated library against the oracle."""
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz", script), os.path.join(ROOT, "tools", "libs360_emu.so"),
                        str(seed), str(cases)], capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("done") and ok in last and "DIFFER" not in r.stdout, r.stdout[-1500:]


def test_emulated_stream_mode_equals_the_reference_programs_chain(tmp_path, emu_programs):
    """--num_frames 3 (one process, device-resident temporal state, frame pipelining) against the equirects the
    reference's program writes when it is run three times, chained with --prev_frame_data_dir."""
    rig = rigutil.scaled_rig_json(os.path.join(ROOT, "tests", "golden", "rig_17cam.json"), str(tmp_path / "rig_small.json"),
                                  refprog.CAM / 2048.0)
    name = "three_frames_sharpened"
    out = refprog.run_stream(os.path.join(emu_programs, "TestRenderStereoPanorama"), str(tmp_path), rig, name)
    golden = json.load(open(refprog.GOLDEN))[name]
    for f in refprog.CASES[name][0]:
        assert refprog._digest_png(os.path.join(out, "eqr_%s.png" % f)) == golden["eqr_%s" % f], f


@pytest.mark.parametrize("name", list(refprog.RAW_CASES))
def test_emulated_raw2rgb_writes_what_the_reference_program_writes(tmp_path, emu_programs, name):
    import isputil
    _, outp = refprog.run_raw_case(os.path.join(emu_programs, "Raw2Rgb"), str(tmp_path), isputil.CONFIG_FULL, name)
    a = refprog.png_pixels_bgr(outp)
    digest = hashlib.sha256(repr((a.shape, str(a.dtype))).encode() + a.tobytes()).hexdigest()
    assert digest == json.load(open(refprog.GOLDEN))["raw2rgb"][name]


@pytest.mark.parametrize("bits,soft", [(12, False), (8, False), (12, True)], ids=["12-pipe", "8-pipe", "12-soft_isp"])
def test_emulated_unpacker(tmp_path, emu_programs, bits, soft):
    from test_gpu_zz_unpacker import check_unpacker
    check_unpacker(os.path.join(emu_programs, "Unpacker"), tmp_path, O, bits, soft)


@pytest.mark.parametrize("soft", [False, True], ids=["pipe", "soft_isp"])
def test_emulated_renderer_fed_from_the_capture_containers(tmp_path, emu_programs, soft):
    """host/TestRenderStereoPanorama --bin_list --isp_dir ("the ISP feeding the GPU directly from .bin") against
    Unpacker -> PNG files -> renderer: the same equirects, two chained frames and the two frames as a stream."""
    from test_gpu_zz_unpacker import check_bin_list
    check_bin_list(os.path.join(emu_programs, "Unpacker"), os.path.join(emu_programs, "TestRenderStereoPanorama"), tmp_path, soft=soft,
                   chain=not soft,  # (the soft ISP: one frame — only the ISP differs between the two)
                   two_devices_env=None if soft else dict(os.environ, EMU_DEVICES="2"))


def test_emulated_optical_flow_harness(tmp_path, emu_programs):
    """host/TestOpticalFlow --mode test (BASELINE configs[1]'s harness) on a small pair: both directions against the oracle."""
    from surround360_amd import synth
    i0, i1 = synth.flow_pair(150, 120, seed=11)
    Image.fromarray(np.ascontiguousarray(i0[:, :, [2, 1, 0, 3]])).save(str(tmp_path / "left.png"))
    Image.fromarray(np.ascontiguousarray(i1[:, :, [2, 1, 0, 3]])).save(str(tmp_path / "right.png"))
    r = subprocess.run([os.path.join(emu_programs, "TestOpticalFlow"), "--mode", "test", "--test_dir", str(tmp_path), "--left_img",
                        "left.png", "--right_img", "right.png", "--flow_alg", "pixflow_low", "--repetitions", "1"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert r.stderr.count("RUNTIME (sec) = ") == 1
    for name, a, b, hint in (("flowLtoR", i0, i1, "LEFT"), ("flowRtoL", i1, i0, "RIGHT")):
        p = str(tmp_path / "disparity" / (name + "_pixflow_low.bin"))
        hdr = np.fromfile(p, dtype=np.int32, count=2)
        got = np.fromfile(p, dtype=np.float32, offset=8).reshape(int(hdr[0]), int(hdr[1]), 2)
        want = O.compute_optical_flow(a, b, "pixflow_low", hint)
        assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32)), name


_EMU_COMMON = dict(S360_TEST_EMULATED_LIB="1", TEST_FLOW_SIZES="150x120,97x131")


def test_emulated_sweep_kernels_give_the_same_flows(emu_programs):
    """tests/test_gpu_zz_variants.py's comparison through the whole flow path of the emulated library: both algorithms, both
    directions, a band of masked rows — the throughput kernel's digest equals the latency kernel's."""
    from test_gpu_zz_variants import _flows_digest
    assert _flows_digest(TEST_SWEEP_MODE="throughput", **_EMU_COMMON) == _flows_digest(**_EMU_COMMON)


def test_operator_level_gpu_tests_pass_on_the_emulated_library(emu_programs):
    """tests/test_gpu_ops.py (every operator of the C ABI against the oracle), tests/test_gpu_isp.py and tests/test_gpu_png.py (the
    device PNG encoder against libpng / zlib), unchanged, in a
    process whose Python binding points at the emulated library (tests/conftest.py: S360_TEST_EMULATED_LIB=1)."""
    import sys
    e = dict(os.environ, S360_TEST_EMULATED_LIB="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_ops.py"),
                        os.path.join(ROOT, "tests", "test_gpu_isp.py"), os.path.join(ROOT, "tests", "test_gpu_png.py"),
                        "-q", "-m", "gpu", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, env=e, timeout=1800, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " passed" in r.stdout and "failed" not in r.stdout and "skipped" not in r.stdout, r.stdout[-500:]


@pytest.mark.skipif(os.environ.get("S360_RUN_8K_REFPROGRAM") != "1", reason="half an hour of CPU: set S360_RUN_8K_REFPROGRAM=1")
def test_emulated_program_equals_the_reference_program_at_8k(tmp_path, emu_programs):
    """BASELINE configs[2] (17 cameras of 2048x2048, eqr 8400x4096 -> 8192x8192, top + bottom) through the emulated HIP
    program and through the reference's own program: the stereo equirect, the 32 flow files and the 36 state images.
    Measured in this container (8 cores): reference program 91 s, emulated library 1119 s; 0 of 201 326 592 equirect
    bytes, 0 of 32 flow files and 0 of 36 state images differ."""
    rig = os.path.join(ROOT, "tests", "golden", "rig_17cam.json")
    imgs8 = refprog.frame_images(rig, 0, size=2048)
    imgs = str(tmp_path / "rgb")
    for cid, img in imgs8.items():
        os.makedirs(os.path.join(imgs, cid))
        Image.fromarray(np.ascontiguousarray(img[:, :, ::-1])).save(os.path.join(imgs, cid, "000000.png"), compress_level=1)
    outs = {}
    for tag, exe in (("ref", refprog.REF_EXE), ("emu", os.path.join(emu_programs, "TestRenderStereoPanorama"))):
        out = str(tmp_path / tag)
        os.makedirs(os.path.join(out, "debug", "000000", "flow_images"))
        os.makedirs(os.path.join(out, "flow", "000000"))
        subprocess.check_call([exe, "--rig_json_file", rig, "--imgs_dir", imgs, "--frame_number", "000000", "--output_data_dir", out,
                               "--prev_frame_data_dir", "NONE", "--output_equirect_path", os.path.join(out, "eqr.png"),
                               "--eqr_width", "8400", "--eqr_height", "4096", "--final_eqr_width", "8192", "--final_eqr_height", "8192",
                               "--enable_top", "--enable_bottom", "--sharpening", "0.0"], timeout=9000)
        outs[tag] = out
    Image.MAX_IMAGE_PIXELS = None
    assert np.array_equal(np.asarray(Image.open(os.path.join(outs["ref"], "eqr.png"))), np.asarray(Image.open(os.path.join(outs["emu"], "eqr.png"))))
    fdir = os.path.join("flow", "000000")
    for f in sorted(os.listdir(os.path.join(outs["ref"], fdir))):
        assert open(os.path.join(outs["ref"], fdir, f), "rb").read() == open(os.path.join(outs["emu"], fdir, f), "rb").read(), f
    idir = os.path.join("debug", "000000", "flow_images")
    for f in sorted(os.listdir(os.path.join(outs["ref"], idir))):
        assert np.array_equal(np.asarray(Image.open(os.path.join(outs["ref"], idir, f))), np.asarray(Image.open(os.path.join(outs["emu"], idir, f)))), f


def test_emulated_program_reads_jpeg_camera_images(tmp_path, emu_programs):
    """The side cameras' images as .jpg (the extension is sniffed from the camera directory like getImageFileExtension,
    SystemUtil.h:96-105; top / bottom stay .png, TestRenderStereoPanorama.cpp:602,652): the frame equals the oracle's
    frame of the decoded images."""
    rig = rigutil.scaled_rig_json(os.path.join(ROOT, "tests", "golden", "rig_17cam.json"), str(tmp_path / "rig_small.json"),
                                  refprog.CAM / 2048.0)
    imgs = refprog.frame_images(rig, 0)
    side_ids, top_id, bottoms = refprog.rig_ids(rig)
    idir, out = str(tmp_path / "rgb"), str(tmp_path / "out")
    decoded = {}
    for cid, img in imgs.items():
        os.makedirs(os.path.join(idir, cid))
        if cid in side_ids:
            p = os.path.join(idir, cid, "000000.jpg")
            Image.fromarray(np.ascontiguousarray(img[:, :, ::-1])).save(p, quality=90, subsampling=2)
            decoded[cid] = np.ascontiguousarray(np.asarray(Image.open(p).convert("RGB"))[:, :, ::-1])
        else:
            Image.fromarray(np.ascontiguousarray(img[:, :, ::-1])).save(os.path.join(idir, cid, "000000.png"))
            decoded[cid] = img
    os.makedirs(os.path.join(out, "debug", "000000", "flow_images"))
    os.makedirs(os.path.join(out, "flow", "000000"))
    eqr = os.path.join(out, "eqr.png")
    r = subprocess.run([os.path.join(emu_programs, "TestRenderStereoPanorama"), "--rig_json_file", rig, "--imgs_dir", idir,
                        "--frame_number", "000000", "--output_data_dir", out, "--output_equirect_path", eqr, "--enable_top",
                        "--enable_bottom", "--eqr_width", str(refprog.EQR_W), "--eqr_height", str(refprog.EQR_H),
                        "--final_eqr_width", str(refprog.FINAL), "--final_eqr_height", str(refprog.FINAL)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    cams, _ = O.load_rig(rig)
    of = O.Frame(cams, O.make_params(eqr_width=refprog.EQR_W, eqr_height=refprog.EQR_H, final_eqr_width=refprog.FINAL,
                                     final_eqr_height=refprog.FINAL, enable_top=1, enable_bottom=1))
    want, _ = of.render([decoded[c] for c in side_ids], decoded[top_id], decoded[bottoms[0]])
    assert np.array_equal(np.asarray(Image.open(eqr))[:, :, ::-1], want)


@pytest.mark.skipif(os.environ.get("S360_RUN_BENCH_DRY") != "1", reason="four minutes of CPU: set S360_RUN_BENCH_DRY=1")
def test_bench_script_dry_run(emu_programs):
    """bench.py from its first line to its JSON line on the emulated library at toy sizes (S360_TEST_EMULATED_LIB=1: NOT a
    measurement): timed region with 2 contexts x 2 frame slots, the check of every timed frame, the isolated / single-frame
    / sharpening / flow-pair / video-stream / end-to-end-files legs, the reference program as the CPU baseline, the ISP leg."""
    import sys
    e = dict(os.environ, S360_TEST_EMULATED_LIB="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--slots", "2", "--inflight", "2",
                        "--video-frames", "3"], capture_output=True, text=True, env=e, timeout=3000, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert "dry_run" in d and "errors" not in d
    assert d["checked"] is True and d["cpu_baseline"]["checked_against_gpu"] is True and d["config2_flow_pair"]["checked"] is True
    assert d["isp"]["checked"] is True
    assert d["video_stream"]["frames"] >= 12 and "spill_ms_per_frame" in d["video_stream"]
    e2e = d["end_to_end_files"]  # the host program from PNG files to PNG files, its last frame against the in-process chain
    assert "error" not in e2e and e2e["last_frame_equals_in_process_stream"] is True and "host_thread_ms_per_frame" in e2e
    vb = d["video_streams_batched"]  # batched chained streams, every stream's last frame against the stream rendered alone
    assert "error" not in vb and vb["checked"] is True and vb["checked_streams"] == vb["streams"] == 4 and vb["distinct_last_frames"] >= 3
    assert e2e["single_invocation"]["equals_stream_frame"] is True and e2e["batched_streams"]["first_frame_equals_stream"] is True
