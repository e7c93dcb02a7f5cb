"""The ISP. Soft ISP: the oracle restatement (oracle/isp.h) against the reference's own CameraIsp.h
compiled from /root/reference over a container-only OpenCV stand-in (oracle/_ref/libref_isp.so), and against the
committed outputs of that library (tests/golden/isp_golden.npz): every ISP arithmetic operation checked there is the
reference's source, executed. Accelerated ISP (round 5): the oracle restatement (oracle/isp_pipe.h) against the reference's
Halide GENERATOR, camera_isp/CameraIspGen.cpp, executed as it lies over a lazy evaluator of the Halide front end
(oracle/ref_shim/halide_eval/Halide.h) under the reference's own CameraIspPipe.h (oracle/_ref/libref_isppipe.so), and against
that library's committed outputs (tests/golden/isp_pipe_golden.npz)."""
import os

import numpy as np
import pytest

import isputil

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "isp_golden.npz")
REF_CONFIG_DIR = "/root/reference/surround360_render/res/config/isp"

CASES = [  # (config, w, h, bpp, demosaic, resize, disable_tone_curve, black_level_offset)
    ("full", 128, 96, 8, 2, 1, 0, 0), ("full", 128, 96, 16, 2, 1, 0, 0), ("full", 128, 96, 8, 0, 1, 0, 0),
    ("full", 130, 70, 16, 2, 2, 0, 25), ("full", 160, 128, 8, 2, 4, 0, 0), ("full", 96, 64, 16, 0, 1, 1, 0),
    ("minimal", 70, 50, 8, 2, 1, 0, 0), ("minimal", 70, 50, 16, 0, 2, 0, 0),
    ("empty", 64, 64, 8, 2, 1, 0, 0), ("empty", 61, 47, 16, 2, 1, 1, 3),
    ("grbg", 100, 84, 8, 2, 1, 0, 0), ("grbg", 100, 84, 16, 0, 1, 0, 0), ("grbg", 256, 192, 16, 2, 8, 0, 0),
]


def _case_id(c):
    return "%s-%dx%d-bpp%d-dm%d-r%d-t%d-o%d" % c


def _raw(case):
    name, w, h = case[:3]
    return isputil.bayer_frame(w, h, seed=w * 7 + h)


@pytest.fixture(scope="module")
def ref(oracle):
    if oracle.ref_isp_lib() is None:
        pytest.skip("oracle/_ref/libref_isp.so not built (needs /root/reference)")
    return oracle


@pytest.mark.parametrize("case", CASES, ids=_case_id)
def test_restatement_equals_compiled_reference(ref, case):
    name, w, h, bpp, dm, rs, tone, off = case
    js, raw = isputil.CONFIGS[name], _raw(case)
    got = ref.isp_run(ref.isp_config_from_json(js, bpp, dm, rs, tone, off), raw)
    want = ref.ref_isp_run(js, raw, bpp, dm, rs, tone, off)
    assert got.shape == want.shape and got.dtype == want.dtype
    assert np.array_equal(got, want), "%d of %d samples differ" % ((got != want).sum(), got.size)
    assert got.std() > 3  # an image, not a constant


@pytest.mark.parametrize("case", CASES, ids=_case_id)
def test_restatement_equals_golden(oracle, case):
    """The same comparison where /root/reference is absent: outputs of oracle/_ref committed by make_isp_golden.py."""
    name, w, h, bpp, dm, rs, tone, off = case
    g = np.load(GOLDEN)
    got = oracle.isp_run(oracle.isp_config_from_json(isputil.CONFIGS[name], bpp, dm, rs, tone, off), _raw(case))
    assert np.array_equal(got, g[_case_id(case)])


@pytest.mark.skipif(not os.path.isdir(REF_CONFIG_DIR), reason="reference checkout not present")
@pytest.mark.parametrize("cfg", ["cmosis_fujinon.json", "cmosis_sunex.json", "passthrough.json"])
def test_shipped_configurations(ref, cfg):
    """The reference's own configurations, read where they lie."""
    js = open(os.path.join(REF_CONFIG_DIR, cfg)).read()
    raw = isputil.bayer_frame(192, 144, seed=5)
    for bpp in (8, 16):
        got = ref.isp_run(ref.isp_config_from_json(js, bpp), raw)
        assert np.array_equal(got, ref.ref_isp_run(js, raw, bpp)), (cfg, bpp)


# ---- the accelerated ISP: oracle/isp_pipe.h against the executed generator ------------------------------------------------------------
PIPE_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "isp_pipe_golden.npz")
PIPE_CASES = [  # (config, w, h, bpp, fast, disable_tone_curve, black_level_offset, unpacker); the first eleven = tests/test_gpu_isp.py's
    ("full", 128, 96, 16, 0, 0, 0, 0), ("full", 130, 70, 8, 0, 0, 0, 0), ("full", 96, 64, 16, 1, 0, 0, 0), ("full", 200, 136, 8, 1, 0, 25, 0),
    ("empty", 61, 47, 16, 0, 1, 3, 0), ("minimal", 70, 50, 8, 0, 0, 0, 0), ("grbg", 100, 84, 16, 0, 0, 0, 0), ("grbg", 64, 64, 16, 1, 0, 0, 0),
    ("full", 640, 480, 16, 0, 0, 0, 0), ("empty", 333, 257, 8, 0, 0, 0, 0),
    ("full", 128, 96, 8, 0, 1, 40, 0), ("minimal", 71, 51, 16, 1, 0, 0, 0), ("empty", 64, 64, 8, 1, 1, 0, 0), ("grbg", 33, 17, 8, 0, 0, 7, 0),
    ("full", 128, 96, 16, 0, 0, 0, 1), ("minimal", 70, 50, 16, 0, 0, 0, 1),  # Unpacker's call sequence (16 bits, full pipeline)
]


def _pipe_id(c):
    return "%s-%dx%d-bpp%d-fast%d-t%d-o%d-u%d" % c


def _pipe_raw(case):
    name, w, h = case[:3]
    return isputil.bayer_frame(w, h, seed=w + 5 * h, pattern="RGGB" if name == "full" else "GBRG")  # (as tests/test_gpu_isp.py)


@pytest.fixture(scope="module")
def refpipe(oracle):
    if oracle.ref_isp_pipe_lib() is None:
        pytest.skip("oracle/_ref/libref_isppipe.so not built (needs /root/reference)")
    return oracle


@pytest.mark.parametrize("case", PIPE_CASES, ids=_pipe_id)
def test_pipe_restatement_equals_executed_generator(refpipe, case):
    """oracle/isp_pipe.h == CameraIspGen.cpp executed under CameraIspPipe.h, bit for bit: 8 / 16 bits x full / fast, both patterns the
    pipeline knows (and two it runs as GBRG), odd sizes, tone curve off, black-level offsets, Raw2Rgb's and Unpacker's call order."""
    name, w, h, bpp, fast, tone, off, unp = case
    if w * h > 100000 and os.environ.get("S360_RUN_SLOW") != "1":
        pytest.skip("half a minute of evaluation: its output is in the golden file, which the next test holds the restatement to")
    js, raw = isputil.CONFIGS[name], _pipe_raw(case)
    got = refpipe.isp_pipe_run(refpipe.isp_config_from_json(js, bpp, 2, 1, tone, off), raw, fast=bool(fast))
    want = refpipe.ref_isp_pipe_run(js, raw, bpp, bool(fast), tone, off, unpacker=bool(unp))
    assert got.shape == want.shape and got.dtype == want.dtype
    assert np.array_equal(got, want), "%d of %d samples differ" % ((got != want).sum(), got.size)
    assert want.std() > 3


@pytest.mark.parametrize("case", PIPE_CASES, ids=_pipe_id)
def test_pipe_restatement_equals_golden(oracle, case):
    """The same where /root/reference is absent: the executed generator's outputs, committed by tests/golden/make_isp_pipe_golden.py."""
    name, w, h, bpp, fast, tone, off, unp = case
    g = np.load(PIPE_GOLDEN)
    got = oracle.isp_pipe_run(oracle.isp_config_from_json(isputil.CONFIGS[name], bpp, 2, 1, tone, off), _pipe_raw(case), fast=bool(fast))
    assert np.array_equal(got, g[_pipe_id(case)])


@pytest.mark.skipif(not os.path.isdir(REF_CONFIG_DIR), reason="reference checkout not present")
@pytest.mark.parametrize("cfg", ["cmosis_fujinon.json", "cmosis_sunex.json", "passthrough.json"])
def test_pipe_shipped_configurations(refpipe, cfg):
    js = open(os.path.join(REF_CONFIG_DIR, cfg)).read()
    raw = isputil.bayer_frame(192, 144, seed=5)
    for bpp, fast in ((8, 0), (16, 0), (16, 1)):
        got = refpipe.isp_pipe_run(refpipe.isp_config_from_json(js, bpp), raw, fast=bool(fast))
        assert np.array_equal(got, refpipe.ref_isp_pipe_run(js, raw, bpp, bool(fast))), (cfg, bpp, fast)


@pytest.mark.skipif(not os.path.isdir(REF_CONFIG_DIR), reason="reference checkout not present")
def test_pipe_comparison_resolves_a_rounding_choice(refpipe):
    """The comparison above is sharp enough to see ONE rounding decision: the evaluator run with a constant float division kept as a
    division (the one reading of Halide this repo takes from memory: its simplifier turns x / 5.0f into x * 0.2f) no longer equals
    the restatement — in a handful of samples, by one count of the tone table's input."""
    import subprocess
    import sys
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import oracle_lib as O, isputil, test_cpu_isp as T\n"
            "case = T.PIPE_CASES[0]; js = isputil.CONFIGS[case[0]]; raw = T._pipe_raw(case)\n"
            "a = O.ref_isp_pipe_run(js, raw, 16, False); b = O.isp_pipe_run(O.isp_config_from_json(js, 16), raw)\n"
            "print(int((a != b).sum()), a.size)" % os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HALIDE_EVAL_TRUE_DIVISION="1")
    n, size = map(int, subprocess.check_output([sys.executable, "-c", code], env=env).split())
    assert 0 < n < size // 20, (n, size)


# (radius, threshold, darkness threshold, does the reference's pass change pixels?)
STUCK = [(1, 5, 0.5, False), (1, 2, 0.9, False), (2, 9, 0.9, False), (1, 9, 0.9, False),  # 2 <= threshold <= region: no-op
         (1, 1, 0.5, True), (1, 0, 0.9, True), (1, 10, 0.9, True), (2, 26, 0.5, True), (1, -3, 0.9, True),
         (2, 1, 2.0, True), (3, 0, 0.4, True)]  # (darkness 2.0: every pixel is filtered)


@pytest.mark.parametrize("radius,thr,dark,active", STUCK)
def test_stuck_pixel_removal_equals_compiled_reference(ref, radius, thr, dark, active):
    """removeStuckPixels with a non-zero radius through the reference's own CameraIsp.h, compiled: the restatement equals
    it, and the loop condition of CameraIsp.h:1090-1092 (a comparison of size_t values) does what include/s360.h says it
    does — for 2 <= stuckPixelThreshold <= the region's population the pass changes nothing (the same pixels as with
    radius 0), outside that range it is a median filter of the dark regions."""
    raw = isputil.bayer_frame(96, 72, seed=9)
    raw[10:14, 20:24] = 65535  # a few hot sites in dark surroundings
    raw[40:60, 10:30] //= 8
    js = isputil.stuck_pixel_config(radius, thr, dark)
    got = ref.isp_run(ref.isp_config_from_json(js, 16), raw)
    want = ref.ref_isp_run(js, raw, 16)
    assert np.array_equal(got, want), "%d samples differ" % (got != want).sum()
    off = ref.ref_isp_run(isputil.stuck_pixel_config(0, thr, dark), raw, 16)
    assert np.array_equal(want, off) == (not active)


def test_library_accepts_the_stuck_pixel_configurations(oracle, s360lib):
    """libs360's host side: a configuration whose stuck-pixel pass is the reference's no-op derives the same tables as with
    radius 0; one whose pass filters is accepted too since round 4 (k_isp_stuck runs it; tests/test_gpu_isp.py, and on the
    emulation tests/test_cpu_library_emulation.py, hold it to the oracle for the cases above); a window wider than 15 x 15 is
    refused with a message."""
    from surround360_amd import isp as I
    ok = I.config_from_json(isputil.stuck_pixel_config(1, 5, 0.11), 16)
    assert (ok.stuck_pixel_radius, ok.stuck_pixel_threshold) == (2, 5) and abs(ok.stuck_pixel_darkness_threshold - 0.11) < 1e-6
    base = I.config_from_json(isputil.CONFIG_FULL, 16)
    for a, b in zip(I.config_tables(ok)[:2], I.config_tables(base)[:2]):
        assert np.array_equal(a, b)
    for thr in (1, 0, 10, -3):
        I.config_tables(I.config_from_json(isputil.stuck_pixel_config(1, thr, 0.5), 16))
    with pytest.raises(Exception, match="stuckPixelRadius"):
        I.config_tables(I.config_from_json(isputil.stuck_pixel_config(4, 1, 0.5), 16))


def test_tables(oracle):
    c = oracle.isp_config_from_json(isputil.CONFIG_FULL, 16)
    ccm, lut = oracle.isp_tables(c)
    assert ccm.shape == (3, 3) and lut.shape == (4096, 3)
    assert lut[0].max() <= lut[-1].min() and lut[-1].max() <= 65535.0 and (np.diff(lut[:, 1]) >= 0).all()
    c.disableToneCurve = 1
    _, lin = oracle.isp_tables(c)
    assert lin[-1, 0] == 65535.0 and lin[0, 0] == 0.0


def test_unsupported_modes_raise(oracle):
    raw = isputil.bayer_frame(32, 32)
    with pytest.raises(RuntimeError):
        oracle.isp_run(oracle.isp_config_from_json(isputil.CONFIG_MINIMAL, 8, 1), raw)  # DCT demosaic


# ---- host half of the product (libs360, no device needed): configuration reading and derived tables -----------------
def _cfg_pairs(oracle, name, **kw):
    from surround360_amd import isp as I
    js = isputil.CONFIGS[name]
    got = I.config_from_json(js, kw.get("bpp", 8), kw.get("dm", 2), kw.get("rs", 1), kw.get("tone", 0), kw.get("off", 0))
    want = oracle.isp_config_from_json(js, kw.get("bpp", 8), kw.get("dm", 2), kw.get("rs", 1), kw.get("tone", 0),
                                       kw.get("off", 0))
    return got, want


@pytest.mark.parametrize("name", sorted(isputil.CONFIGS))
def test_library_reads_configuration_like_the_constructor(oracle, s360lib, name):
    """s360_isp_config_from_json against the test plumbing's independent reading of the same JSON (which the compiled
    reference agrees with through test_restatement_equals_compiled_reference)."""
    got, want = _cfg_pairs(oracle, name, bpp=16, dm=0, rs=2, tone=1, off=7)
    pairs = [("black_level", "blackLevel"), ("clamp_min", "clampMin"), ("clamp_max", "clampMax"),
             ("white_balance_gain", "whiteBalanceGain"), ("ccm", "ccm"), ("gamma", "gamma"),
             ("low_key_boost", "lowKeyBoost"), ("high_key_boost", "highKeyBoost"), ("sharpening", "sharpening")]
    for a, b in pairs:
        assert list(getattr(got, a)) == list(getattr(want, b)), a
    for a, b in [("saturation", "saturation"), ("contrast", "contrast"), ("sharpening_support", "sharpeningSupport"),
                 ("noise_core", "noiseCore"), ("n_vignette_h", "nVignetteH"), ("n_vignette_v", "nVignetteV"),
                 ("stuck_pixel_radius", "stuckPixelRadius"), ("stuck_pixel_threshold", "stuckPixelThreshold"),
                 ("bayer_pattern", "bayerPattern"),
                 ("output_bpp", "outputBpp"), ("demosaic_filter", "demosaicFilter"), ("resize", "resize"),
                 ("disable_tone_curve", "disableToneCurve"), ("black_level_offset", "blackLevelOffset")]:
        assert getattr(got, a) == getattr(want, b), a
    for i in range(got.n_vignette_h):
        assert list(got.vignette_roll_off_h[i]) == list(want.vignetteRollOffH[i])
    for i in range(got.n_vignette_v):
        assert list(got.vignette_roll_off_v[i]) == list(want.vignetteRollOffV[i])


@pytest.mark.parametrize("name,bpp,tone", [("full", 8, 0), ("full", 16, 0), ("grbg", 16, 0), ("empty", 8, 1)])
def test_library_tables_equal_oracle(oracle, s360lib, name, bpp, tone):
    """Composite CCM and tone curve built on the host by libs360 (powf / tanf like the reference) == the oracle's, bit
    for bit."""
    from surround360_amd import isp as I
    got, want = _cfg_pairs(oracle, name, bpp=bpp, tone=tone)
    ccm, lut, _, _ = I.config_tables(got)
    occm, olut = oracle.isp_tables(want)
    assert np.array_equal(ccm.view(np.uint32), occm.view(np.uint32))
    assert np.array_equal(lut.view(np.uint32), olut.view(np.uint32))


def test_library_rejects_unsupported(s360lib):
    from surround360_amd import _capi, isp as I
    with pytest.raises(_capi.S360Error):
        I.config_from_json('{"CameraIsp": {"ccm": [[1, 0], [0, 1]]}}')
    with pytest.raises(_capi.S360Error):
        I.config_from_json('{"CameraIsp": {"bayerPattern": "XYZW"}}')
    for bad in ('{"CameraIsp": {"stuckPixelRadius": 1e999}}', '{"CameraIsp": {"stuckPixelThreshold": -1e300}}',
                '{"CameraIsp": {"stuckPixelRadius": "2"}}', '{"CameraIsp": {"saturation": null}}',
                '{"CameraIsp": {"gamma": [1, 2]}}', '{"CameraIsp": {"gamma": [1, 2, true]}}', '{"CameraIsp": {}} x'):
        with pytest.raises(_capi.S360Error):  # (an int no int holds was undefined behaviour before the range check)
            I.config_from_json(bad)
    c = I.config_from_json(isputil.CONFIG_MINIMAL, demosaic_filter=1)
    with pytest.raises(_capi.S360Error):
        I.config_tables(c)  # DCT demosaic
    c = I.config_from_json(isputil.CONFIG_MINIMAL, resize=3)
    with pytest.raises(_capi.S360Error):
        I.config_tables(c)


def test_library_vignette_curves(s360lib):
    """curveHAtPixel / curveVAtPixel: De Casteljau in float32, written out again here with numpy scalars."""
    from surround360_amd import isp as I
    import json
    cfg = I.config_from_json(isputil.CONFIG_FULL)
    w, h = 37, 53
    _, _, ch, cv = I.config_tables(cfg, w, h)
    f = np.float32

    def bez(pts, t):
        pts = [f(p) for p in pts]
        while len(pts) > 1:  # lerp(a, b, t) = a * (1 - t) + b * t, level by level == the reference's recursion
            pts = [f(f(a * f(f(1) - t)) + f(b * t)) for a, b in zip(pts[:-1], pts[1:])]
        return pts[0]
    j = json.loads(isputil.CONFIG_FULL)["CameraIsp"]
    md = f(max(w, h))
    for x in (0, 1, 17, 36):
        for k in range(3):
            assert ch[x, k] == bez([p[k] for p in j["vignetteRollOffH"]], f(f(x) / md))
    for y in (0, 26, 52):
        for k in range(3):
            assert cv[y, k] == bez([p[k] for p in j["vignetteRollOffV"]], f(f(y) / md))


# ---- the HIP kernels themselves, emulated on the CPU (developer tool: tools/hip_cpu_shim, tools/isp_emulate.cpp) ----
@pytest.mark.parametrize("case", [("full", 64, 48, 16, 2, 1, 0, 0), ("full", 70, 50, 8, 0, 1, 0, 0),
                                  ("grbg", 96, 64, 16, 2, 2, 0, 25),
                                  # several 64-position tiles + a remainder, more than 21 rows per wave / 32 rows per batch (the
                                  # tiled IIR passes); a width below one tile
                                  ("full", 150, 70, 16, 2, 1, 0, 0), ("full", 40, 34, 8, 2, 1, 0, 0)], ids=_case_id)
def test_kernels_emulated_on_cpu(oracle, s360lib, case):
    """isp_kernels.hip + isp.cpp compiled with g++ over a stand-in for the HIP runtime, one std::thread per GPU thread:
    the kernels' indexing and float arithmetic (IEEE on both sides, -ffp-contract=off) against the oracle, here where no
    GPU is attached. The GPU itself is tested by tests/test_gpu_isp.py."""
    import ctypes as C
    import subprocess
    from surround360_amd import isp as I
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "tools"), "-s", "libisp_emu.so"])
    emu = C.CDLL(os.path.join(root, "tools", "libisp_emu.so"))
    name, w, h, bpp, dm, rs, tone, off = case
    js, raw = isputil.CONFIGS[name], isputil.bayer_frame(w, h, seed=w + h)
    cfg = I.config_from_json(js, bpp, dm, rs, tone, off)
    got = np.zeros((h // rs, w // rs, 3), np.uint8 if bpp == 8 else np.uint16)
    err = C.create_string_buffer(256)
    assert emu.emu_isp_run(C.byref(cfg), raw.ctypes.data_as(C.c_void_p), w, h, got.ctypes.data_as(C.c_void_p), err, 256) == 0, err.value
    want = oracle.isp_run(oracle.isp_config_from_json(js, bpp, dm, rs, tone, off), raw)
    assert np.array_equal(got, want)


# ---- packed sensor frames (Unpacker.cpp:136-143; RawConverter.cpp:15-59) ---------------------------------------------
@pytest.mark.parametrize("bits,w,h", [(8, 64, 48), (12, 64, 48), (12, 130, 7), (8, 63, 5)])
def test_unpack_restatement_equals_compiled_reference(ref, bits, w, h):
    rng = np.random.default_rng(bits + w)
    frame = rng.integers(0, 256, ref.isp_packed_bytes(bits, w, h) + 2, dtype=np.uint8)
    a, b = ref.isp_unpack_frame(frame, bits, w, h), ref.ref_unpack_frame(frame, bits, w, h)
    assert np.array_equal(a, b)


def test_unpack_inverts_packing(oracle):
    raw = isputil.bayer_frame(64, 32, seed=3)
    got = oracle.isp_unpack_frame(isputil.pack_frame(raw, 12), 12, 64, 32)
    v = (raw >> 4).astype(np.uint32)
    assert np.array_equal(got, ((v << 4) | (v >> 8)).astype(np.uint16))  # 12 -> 16 bits by replicating the top bits
    got8 = oracle.isp_unpack_frame(isputil.pack_frame(raw, 8), 8, 64, 32)
    assert np.array_equal(got8, (raw >> 8).astype(np.uint16) * 0x101)


def test_packed_path_emulated_on_cpu(oracle, s360lib):
    import ctypes as C
    import subprocess
    from surround360_amd import isp as I
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "tools"), "-s", "libisp_emu.so"])
    emu = C.CDLL(os.path.join(root, "tools", "libisp_emu.so"))
    w, h = 64, 48
    for bits in (12, 8):
        frame = isputil.pack_frame(isputil.bayer_frame(w, h, seed=9), bits)
        cfg = I.config_from_json(isputil.CONFIG_GRBG_NOSHARP, 16)
        got = np.zeros((h, w, 3), np.uint16)
        err = C.create_string_buffer(256)
        assert emu.emu_isp_run_packed(C.byref(cfg), frame.ctypes.data_as(C.c_void_p), bits, w, h,
                                      got.ctypes.data_as(C.c_void_p), err, 256) == 0, err.value
        raw16 = oracle.isp_unpack_frame(frame, bits, w, h)
        want = oracle.isp_run(oracle.isp_config_from_json(isputil.CONFIG_GRBG_NOSHARP, 16), raw16)
        assert np.array_equal(got, want), bits
