"""Independent cross-checks of the oracle's OpenCV restatements (oracle/cvlite.h).

OpenCV itself is not available here, so nothing below is a bit-level pin. What it does pin is the
SEMANTICS — tap positions, border modes, coefficient formulas, half-pixel conventions — of every primitive against
third-party implementations that were written independently of this repository: torch's `interpolate` / `grid_sample`
(whose bicubic uses the same A = -0.75 kernel and half-pixel centres as cv::resize / cv::remap), scipy.ndimage
(median, correlate with mirror = BORDER_REFLECT_101, grey erosion) and closed-form numpy. Tolerances are the float
round-off / fixed-point quantisation of the primitive, stated per test."""
import numpy as np
import pytest
import scipy.ndimage as ndi
import torch
import torch.nn.functional as F


def _t(a):  # HxWxC float32 -> 1xCxHxW tensor
    a = np.asarray(a, np.float32)
    if a.ndim == 2:
        a = a[..., None]
    return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))[None]


def _n(t):
    return t[0].numpy().transpose(1, 2, 0)


@pytest.mark.parametrize("dw,dh", [(36, 27), (80, 60), (37, 29)])
def test_resize_linear_f32_vs_torch(oracle, dw, dh):
    """cv::resize INTER_LINEAR on CV_32F == torch bilinear, align_corners=False (no antialias): float round-off only."""
    rng = np.random.default_rng(10)
    src = rng.normal(size=(30, 40, 2)).astype(np.float32)
    want = _n(F.interpolate(_t(src), size=(dh, dw), mode="bilinear", align_corners=False))
    got = oracle.resize_linear_f32(src, dw, dh)
    assert np.abs(got - want).max() < 1e-4  # torch derives the source coordinate from a float32 scale


@pytest.mark.parametrize("dw,dh", [(44, 33), (36, 27), (81, 62)])
def test_resize_cubic_f32_vs_torch(oracle, dw, dh):
    """cv::resize INTER_CUBIC on CV_32FC2 (A = -0.75, replicated border taps) == torch bicubic, align_corners=False."""
    rng = np.random.default_rng(11)
    src = rng.normal(size=(30, 40, 2)).astype(np.float32)
    want = _n(F.interpolate(_t(src), size=(dh, dw), mode="bicubic", align_corners=False))
    got = oracle.resize_cubic_f32(src, dw, dh)
    assert np.abs(got - want).max() < 2e-4  # (float32 coordinate scale in torch)


@pytest.mark.parametrize("dw,dh", [(20, 15), (23, 17), (64, 48)])
def test_resize_cubic_u8_vs_torch(oracle, dw, dh):
    """8-bit cubic resize: the 11-bit fixed-point taps and the final rounding stay within 1 LSB of the float filter."""
    rng = np.random.default_rng(12)
    src = rng.integers(0, 256, (30, 40, 4), dtype=np.uint8)
    want = _n(F.interpolate(_t(src.astype(np.float32)), size=(dh, dw), mode="bicubic", align_corners=False))
    got = oracle.resize_cubic_u8(src, dw, dh).astype(np.float32)
    assert np.abs(got - np.clip(want, 0, 255)).max() <= 1.0
    assert (np.abs(got - np.clip(np.rint(want), 0, 255)) > 0).mean() < 0.06  # and almost always equal to the rounded one


def test_resize_cubic_u8_sse2_region_and_tail(oracle):
    """Exact x0.5 downscale: taps (-192, 1216, 1216, -192)/2048 make every result a multiple of 1/1024, so exact .5
    ties occur; the SSE2-covered elements round them half-to-even (cvtps2dq), the scalar tail (last pixel of an
    odd-width row) half-up. Checked against exact integer arithmetic."""
    rng = np.random.default_rng(13)
    src = rng.integers(0, 256, (64, 54, 4), dtype=np.uint8)  # -> 27 x 32, odd width: last pixel is the scalar tail
    got = oracle.resize_cubic_u8(src, 27, 32).astype(np.int64)
    c = np.array([-3, 19, 19, -3], np.int64)
    s = src.astype(np.int64)
    ys = np.clip(2 * np.arange(32)[:, None] - 1 + np.arange(4)[None, :], 0, 63)
    xs = np.clip(2 * np.arange(27)[:, None] - 1 + np.arange(4)[None, :], 0, 53)
    num = np.einsum("yaxbc,a,b->yxc", s[ys][:, :, xs], c, c)  # value * 1024
    half_up = np.clip((num + 512) >> 10, 0, 255)
    q, rem = num >> 10, num & 1023
    half_even = np.clip(q + ((rem > 512) | ((rem == 512) & (q & 1 == 1))), 0, 255)
    assert np.array_equal(got[:, :26], half_even[:, :26])
    assert np.array_equal(got[:, 26:], half_up[:, 26:])
    assert (half_even != half_up).any(), "the test image should contain ties"


def test_remap_cubic_vs_grid_sample(oracle):
    """cv::remap INTER_CUBIC, BORDER_CONSTANT(0) == torch grid_sample(bicubic, zeros, align_corners=True) up to the
    1/32-pixel coordinate quantisation (float source) and the 15-bit weights + rounding (8-bit source)."""
    rng = np.random.default_rng(14)
    h, w = 40, 50
    srcf = ndi.gaussian_filter(rng.normal(size=(h, w, 2)), (2, 2, 0)).astype(np.float32) * 4
    src8 = np.clip(ndi.gaussian_filter(rng.normal(size=(h, w, 4)), (2, 2, 0)) * 300 + 128, 0, 255).astype(np.uint8)
    yy, xx = np.meshgrid(np.arange(30, dtype=np.float32), np.arange(36, dtype=np.float32), indexing="ij")
    # on the 1/32 grid, so that only arithmetic (not coordinate quantisation) differs; reaches outside the image
    mx = np.round((xx * 1.31 - 2.2 + 0.4 * np.sin(yy / 5)) * 32) / 32
    my = np.round((yy * 1.27 - 1.6 + 0.3 * np.cos(xx / 7)) * 32) / 32
    mp = np.stack([mx, my], -1).astype(np.float32)
    grid = torch.from_numpy(np.stack([2 * mx / (w - 1) - 1, 2 * my / (h - 1) - 1], -1).astype(np.float32))[None]
    wantf = _n(F.grid_sample(_t(srcf), grid, mode="bicubic", padding_mode="zeros", align_corners=True))
    gotf = oracle.remap_cubic_f32(srcf, mp)
    assert np.abs(gotf - wantf).max() < 2e-5
    want8 = _n(F.grid_sample(_t(src8.astype(np.float32)), grid, mode="bicubic", padding_mode="zeros", align_corners=True))
    got8 = oracle.remap_cubic_u8(src8, mp).astype(np.float32)
    assert np.abs(got8 - np.clip(want8, 0, 255)).max() <= 1.0


@pytest.mark.parametrize("n,sigma", [(5, 0.25), (3, 0.5), (15, 8.0), (3, 1.0)])
def test_gaussian_blur_vs_scipy(oracle, n, sigma):
    """GaussianBlur == separable correlation with the normalised exp(-x^2 / 2 sigma^2) taps, BORDER_REFLECT_101
    (scipy 'mirror'); the evaluation order only moves the last bits."""
    rng = np.random.default_rng(15)
    src = rng.normal(size=(33, 41, 2)).astype(np.float32)
    x = np.arange(n) - (n - 1) / 2
    k = np.exp(-0.5 * x * x / (sigma * sigma))
    k /= k.sum()
    assert np.abs(oracle.gaussian_kernel(n, sigma) - k).max() < 1e-7
    want = ndi.correlate1d(ndi.correlate1d(src.astype(np.float64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
    got = oracle.gaussian_blur_f32(src, n, sigma)
    assert np.abs(got - want).max() < 2e-6


def test_gaussian_blur_row_order_is_left_to_right(oracle):
    """The 15-tap row pass accumulates left to right (OpenCV's generic RowFilter), the 5-tap one in symmetric pairs
    (SymmRowSmallFilter): replayed here in numpy float32 on a single row (the column pass of a 1-row image with
    BORDER_REFLECT_101 sees the same row 15 times: centre*k + pairs)."""
    rng = np.random.default_rng(16)
    row = rng.normal(size=(1, 64)).astype(np.float32)
    for n, sigma, pairs in ((15, 8.0, False), (5, 0.25, True)):
        k = oracle.gaussian_kernel(n, sigma)
        r = n // 2
        xi = np.abs(np.arange(-r, 64 + r))
        xi = np.where(xi > 63, 126 - xi, xi)
        mid = np.zeros(64, np.float32)
        for x in range(64):
            if pairs:
                s = np.float32(k[r] * row[0, xi[x + r]])
                for j in range(1, r + 1):
                    s = np.float32(s + np.float32(k[r + j] * np.float32(row[0, xi[x + r + j]] + row[0, xi[x + r - j]])))
            else:
                s = np.float32(0)
                for j in range(n):
                    s = np.float32(s + np.float32(k[j] * row[0, xi[x + j]]))
            mid[x] = s
        out = np.float32(k[r] * mid)  # 1-row image: every reflected row is row 0
        for j in range(1, r + 1):
            out = np.float32(out + np.float32(k[r + j] * np.float32(mid + mid)))
        got = oracle.gaussian_blur_f32(row, n, sigma)[0]
        assert np.array_equal(got.view(np.uint32), out.view(np.uint32)), (n, np.abs(got - out).max())


def test_sobel_median_erode_vs_scipy(oracle):
    rng = np.random.default_rng(17)
    img = rng.normal(size=(28, 35)).astype(np.float32)
    pad = np.pad(img, 1, mode="edge")  # BORDER_REPLICATE
    assert np.array_equal(oracle.sobel(img, 0), pad[1:-1, 2:] - pad[1:-1, :-2])
    assert np.array_equal(oracle.sobel(img, 1), pad[2:, 1:-1] - pad[:-2, 1:-1])
    f = rng.normal(size=(28, 35, 2)).astype(np.float32)
    m = oracle.median5(f)
    for c in range(2):  # exact selection: implementation independent
        assert np.array_equal(m[..., c], ndi.median_filter(f[..., c], size=5, mode="nearest"))
    # featherAlphaChannel = erode(MORPH_CROSS 2e+1, border = +inf) then the 8-bit Gaussian; checked on the erosion by
    # making the Gaussian a no-op is not possible, so compare the whole thing with a float restatement to 1 LSB
    a = (rng.random((60, 70)) > 0.08).astype(np.uint8) * 255
    a = ndi.grey_closing(a, size=3)
    rgba = np.zeros((60, 70, 4), np.uint8)
    rgba[..., 3] = a
    e = 7
    cross = np.zeros((2 * e + 1, 2 * e + 1), bool)
    cross[e, :] = cross[:, e] = True
    er = ndi.grey_erosion(a, footprint=cross, mode="constant", cval=255)
    x = np.arange(e) - (e - 1) / 2
    k = np.exp(-0.5 * x * x / ((e / 2.0) ** 2))
    k /= k.sum()
    want = ndi.correlate1d(ndi.correlate1d(er.astype(np.float64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
    got = oracle.feather_alpha_channel(rgba, e)[..., 3].astype(np.float64)
    assert np.abs(got - want).max() <= 2.5  # taps quantised to 1/256 per pass (their sum is 254..258), then rounding


def test_gauss_u8_ties_round_half_even_except_tail(oracle):
    """Column pass of the 8-bit Gaussian: SSE2 groups of 4 columns round exact .5 ties half-to-even, the last
    w % 4 columns (scalar FixedPtCastEx) half-up. ksize 3 / sigma 1.5 has taps (79, 98, 79)/256: rows of 0 and 128
    give (79*128*256 + ...)/65536 sums with exact ties."""
    e = 3
    k = np.rint(oracle.gaussian_kernel(e, e / 2.0) * 256).astype(np.int64)
    assert k.sum() == 256
    rng = np.random.default_rng(18)
    a = rng.integers(0, 256, (40, 43), dtype=np.uint8)
    a[10:30, 5:40] = 255  # an un-eroded plateau so that the erosion leaves structure behind
    rgba = np.zeros((40, 43, 4), np.uint8)
    rgba[..., 3] = a
    cross = np.zeros((2 * e + 1, 2 * e + 1), bool)
    cross[e, :] = cross[:, e] = True
    er = ndi.grey_erosion(a, footprint=cross, mode="constant", cval=255).astype(np.int64)
    row = ndi.correlate1d(er, k, axis=1, mode="mirror")
    num = ndi.correlate1d(row, k, axis=0, mode="mirror")
    q, rem = num >> 16, num & 0xFFFF
    half_even = np.clip(q + ((rem > 0x8000) | ((rem == 0x8000) & (q & 1 == 1))), 0, 255)
    half_up = np.clip((num + 0x8000) >> 16, 0, 255)
    got = oracle.feather_alpha_channel(rgba, e)[..., 3].astype(np.int64)
    assert np.array_equal(got[:, :40], half_even[:, :40]) and np.array_equal(got[:, 40:], half_up[:, 40:])


def test_gray_conversion_formula(oracle):
    """cvtColor BGRA2GRAY 8-bit: 14-bit fixed point of 0.114 B + 0.587 G + 0.299 R, via pixflow_entry's grey plane."""
    rng = np.random.default_rng(19)
    img = rng.integers(0, 256, (64, 64, 4), dtype=np.uint8)
    img[..., 3] = 255
    down, grey, alpha = oracle.pixflow_entry(img)[:3]
    d = down.astype(np.float64)
    want = (0.114 * d[..., 0] + 0.587 * d[..., 1] + 0.299 * d[..., 2]) / 255.0
    # grey is pre-blurred with the 5x5 sigma 0.25 Gaussian (centre tap 0.9993): compare loosely
    assert np.abs(grey - want).max() < 0.01
    assert np.abs(alpha - 1.0).max() < 1e-6
