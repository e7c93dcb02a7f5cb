"""INTEGRATION.md section 1 on the hardware: the REFERENCE'S OWN TestRenderStereoPanorama (oracle/_ref/
TestRenderStereoPanorama_ops_hip: its sources compiled where they lie under /root/reference over oracle/ref_shim, with the
PixFlowHip subclass, the two extra factory names and the four free functions of oracle/ref_binding/ injected, linked against
surround360_amd/libs360.so; built by __graft_entry__.build() where the reference exists, travels to the GPU box prebuilt)
run with --side_flow_alg pixflow_low_hip --polar_flow_alg pixflow_low_hip: the reference's frame with every projection,
blend, shift, feather and flow computed on the GPU — the flows by the reference's own 14 + 4 threads sharing one context —
held to the digests of the unmodified reference program.
tests/test_cpu_library_emulation.py runs the same program against the library's CPU emulation (green).
A GATE since round 4: the driver's MI355X passed all five cases at the end of round 3 (GPUTEST_r03.json: 5 xpassed), so a
difference is a failure of the suite now."""
import json
import os

import pytest

import refprog
import rigutil

pytestmark = pytest.mark.gpu

EXE = os.path.join(refprog.ROOT, "oracle", "_ref", "TestRenderStereoPanorama_ops_hip")


@pytest.mark.parametrize("name,flags", [
    ("two_frames", ["--side_flow_alg", "pixflow_low_hip", "--polar_flow_alg", "pixflow_low_hip"]),
    ("pole_removal", ["--side_flow_alg", "pixflow_low_hip", "--polar_flow_alg", "pixflow_low_hip", "--poleremoval_flow_alg", "pixflow_low_hip"])])
def test_reference_program_with_the_integration_binding_on_the_gpu(tmp_path, name, flags, s360lib):
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/TestRenderStereoPanorama_ops_hip is built where /root/reference exists (make -C oracle ref_binding)")
    rig = rigutil.scaled_rig_json(os.path.join(refprog.ROOT, "tests", "golden", "rig_17cam.json"),
                                  str(tmp_path / "rig_small.json"), refprog.CAM / 2048.0)
    out = refprog.run_case(EXE, str(tmp_path), rig, name, timeout=120, more_args=flags)
    got = refprog.digests(out, name)
    golden = json.load(open(refprog.GOLDEN))[name]
    differing = sorted(k for k in golden if got.get(k) != golden[k])
    assert not differing, "%d of %d files differ from the unmodified reference program's: %s" % (len(differing), len(golden), differing[:12])


@pytest.mark.parametrize("name", list(refprog.RAW_CASES))
def test_reference_raw2rgb_with_the_integration_binding_on_the_gpu(tmp_path, name, s360lib):
    """the reference's Raw2Rgb with the CameraIspGpu subclass (oracle/ref_binding/CameraIspGpu.h)."""
    import hashlib
    import isputil
    exe = os.path.join(refprog.ROOT, "oracle", "_ref", "Raw2Rgb_hip")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/Raw2Rgb_hip is built where /root/reference exists (make -C oracle ref_binding_isp)")
    _, outp = refprog.run_raw_case(exe, str(tmp_path), isputil.CONFIG_FULL, name)
    a = refprog.png_pixels_bgr(outp)
    digest = hashlib.sha256(repr((a.shape, str(a.dtype))).encode() + a.tobytes()).hexdigest()
    assert digest == json.load(open(refprog.GOLDEN))["raw2rgb"][name]


@pytest.mark.parametrize("name", list(refprog.PIPE_RAW_CASES))
def test_reference_raw2rgb_accelerated_on_the_gpu(tmp_path, name, oracle, s360lib):
    """The reference's Raw2Rgb --accelerate / CameraIspPipe.h, unmodified, over oracle/ref_binding/halide_shim (the four functions
    Halide would generate, written over s360_isp_pipe_generated): tests/test_cpu_library_emulation.py has the description."""
    from test_cpu_library_emulation import check_pipe_raw_case
    exe = os.path.join(refprog.ROOT, "oracle", "_ref", "Raw2Rgb_pipe_hip")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/Raw2Rgb_pipe_hip is built where /root/reference exists (make -C oracle ref_binding_pipe)")
    check_pipe_raw_case(exe, os.path.join(refprog.ROOT, "host", "Raw2Rgb"), str(tmp_path), name)


@pytest.mark.parametrize("bits", [12, 8])
def test_reference_unpacker_on_the_gpu(tmp_path, bits, oracle, s360lib):
    """The reference's own Unpacker with its CameraIspPipe on the library."""
    from test_gpu_zz_unpacker import check_unpacker
    exe = os.path.join(refprog.ROOT, "oracle", "_ref", "Unpacker_hip")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/Unpacker_hip is built where /root/reference exists (make -C oracle ref_binding_pipe)")
    check_unpacker(exe, tmp_path, oracle, bits, soft=False)
