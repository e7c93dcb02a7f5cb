"""The HIP program against the REFERENCE'S OWN PROGRAM. tests/golden/refprogram_golden.json holds SHA-256 digests of every
file the reference's TestRenderStereoPanorama — compiled from /root/reference over oracle/ref_shim, see
tests/test_cpu_refprogram.py, which also proves those outputs equal the oracle's — writes for the cases of
tests/refprog.py (two chained frames; sharpening + cubemap + pixflow_search_20; pole removal over two frames).
host/TestRenderStereoPanorama, running the HIP library, gets the same integer-generated inputs and the same flags here
and must produce the same stereo equirects, cubemap, 28 + 4 flow files per frame and state images, digest for digest.
(No reference and no oracle involved at run time. Sorted last: added after round 2's GPU minutes were spent.)"""
import json
import os
import subprocess

import pytest

import refprog
import rigutil

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(refprog.CASES))
def test_hip_program_writes_what_the_reference_program_writes(tmp_path, name, s360lib):
    subprocess.check_call(["make", "-C", os.path.join(refprog.ROOT, "host"), "-s"])
    rig = rigutil.scaled_rig_json(os.path.join(refprog.ROOT, "tests", "golden", "rig_17cam.json"),
                                  str(tmp_path / "rig_small.json"), refprog.CAM / 2048.0)
    out = refprog.run_case(refprog.HOST_EXE, str(tmp_path), rig, name)
    got = refprog.digests(out, name)
    golden = json.load(open(refprog.GOLDEN))[name]
    missing = sorted(k for k in golden if k not in got)
    assert not missing, "files the reference program writes and this one does not: %s" % missing
    differing = sorted(k for k in golden if got[k] != golden[k])
    assert not differing, "%d of %d files differ from the reference program's: %s" % (len(differing), len(golden), differing[:12])


def test_hip_stream_mode_equals_the_reference_programs_chain(tmp_path, s360lib):
    """--num_frames 3 (one process: device-resident temporal state, frame pipelining, overlapped I/O) against the
    equirects the reference's program writes when it is run three times, chained with --prev_frame_data_dir."""
    subprocess.check_call(["make", "-C", os.path.join(refprog.ROOT, "host"), "-s"])
    rig = rigutil.scaled_rig_json(os.path.join(refprog.ROOT, "tests", "golden", "rig_17cam.json"),
                                  str(tmp_path / "rig_small.json"), refprog.CAM / 2048.0)
    name = "three_frames_sharpened"
    out = refprog.run_stream(refprog.HOST_EXE, str(tmp_path), rig, name)
    golden = json.load(open(refprog.GOLDEN))[name]
    for f in refprog.CASES[name][0]:
        assert refprog._digest_png(os.path.join(out, "eqr_%s.png" % f)) == golden["eqr_%s" % f], f


@pytest.mark.parametrize("name", list(refprog.RAW_CASES))
def test_hip_raw2rgb_writes_what_the_reference_program_writes(tmp_path, name, s360lib):
    """host/Raw2Rgb (the HIP ISP) against the digests of the reference's own Raw2Rgb program for the same inputs and flags."""
    import hashlib
    import isputil
    subprocess.check_call(["make", "-C", os.path.join(refprog.ROOT, "host"), "-s"])
    _, outp = refprog.run_raw_case(refprog.HOST_RAW2RGB, str(tmp_path), isputil.CONFIG_FULL, name)
    a = refprog.png_pixels_bgr(outp)
    digest = hashlib.sha256(repr((a.shape, str(a.dtype))).encode() + a.tobytes()).hexdigest()
    assert digest == json.load(open(refprog.GOLDEN))["raw2rgb"][name]


def test_two_streams_in_one_process(tmp_path, s360lib):
    """host/TestRenderStereoPanorama --num_frames 3 --num_streams 2 (BASELINE configs[4] on N GPUs: one stream per GPU; on a
    one-GPU box both streams share the device, each with a context, host thread and temporal state of its own) writes file for
    file what two separate invocations with the two frame ranges write, and the first stream's frames are the reference
    program's chain (tests/test_cpu_parallel.py runs the same check on the emulation with two devices)."""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(refprog.ROOT, "host"), "-s"])
    refprog.check_two_streams(os.path.join(refprog.ROOT, "host", "TestRenderStereoPanorama"), tmp_path)
