"""Soft ISP: the oracle restatement (oracle/isp.h) against the reference's own CameraIsp.h
compiled from /root/reference over a container-only OpenCV stand-in (oracle/_ref/libref_isp.so), and against the
committed outputs of that library (tests/golden/isp_golden.npz). This is the one part of the oracle that is PINNED:
every ISP arithmetic operation checked here is the reference's source, executed."""
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
    c = oracle.isp_config_from_json(isputil.CONFIG_MINIMAL, 8)
    c.stuckPixelRadius = 2
    with pytest.raises(RuntimeError):
        oracle.isp_run(c, raw)
