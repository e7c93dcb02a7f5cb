"""The PixFlow sweep kernels without a GPU: surround360_amd/csrc/sweep_lock.hip and sweep_quad.hip compiled for the CPU
over tools/hip_wave_shim (workgroups = OS threads, lanes = coroutines, DPP / ballot / s_barrier as rendezvous, bands
chained through real atomics) and compared bit for bit with a plain raster-order loop of PixFlow.h:388-410
(tools/sweep_emulate.cpp). Covers what the GPU parity tests cover for the sweeps — both directions, masked pixels,
fully masked bands, first rows that need / do not need the band above, operands outside the fast path's range, the
persistent-wave ticket loop — plus something they cannot: the lanes of a wave run in forward, reverse and shuffled
order, so a wave-synchronous LDS hand-over that is not marked (S360_WAVE_SYNC) shows up as a mismatch."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tools", "sweep_emulate")


@pytest.fixture(scope="module")
def emulator():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools"), "-s", "sweep_emulate"])
    return EXE


def _run(exe, args, **env):
    e = dict(os.environ)
    e.update({k: str(v) for k, v in env.items()})
    r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, env=e, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(" 0 differ, error flag 0") == 2, r.stdout  # forward and backward sweep
    return r.stdout


@pytest.mark.parametrize("kernel", ["lock", "quad"])
@pytest.mark.parametrize("mask", ["none", "random", "bands", "rows0", "most"])
def test_sweeps_equal_the_raster_order_loop(emulator, kernel, mask):
    for fast in (0, 1):
        _run(emulator, [kernel, 37, 40, 2, 11, mask, fast, 1])


@pytest.mark.parametrize("kernel", ["lock", "quad"])
@pytest.mark.parametrize("order", ["rev", "shuffle"])
def test_lane_order_does_not_matter(emulator, kernel, order):
    _run(emulator, [kernel, 53, 35, 3, 12, "random", 1, 1], EMU_LANE_ORDER=order)
    _run(emulator, [kernel, 53, 35, 3, 13, "bands", 1, 1], EMU_LANE_ORDER=order)


@pytest.mark.parametrize("w,h", [(3, 2), (16, 16), (17, 33), (64, 17), (130, 21)])
def test_sizes(emulator, w, h):
    """Narrower than a band's skew, exactly one band, one row more than two bands, chunk boundaries, several chunks."""
    for kernel in ("lock", "quad"):
        _run(emulator, [kernel, w, h, 2, 14, "random", 1, 1])


def test_quad_tuning_switches_keep_the_result(emulator):
    """Two persistent waves taking all tickets; no row flags."""
    _run(emulator, ["quad", 70, 50, 2, 15, "bands", 1, 1], S360_QUAD_WAVES_PER_CU=2, EMU_CUS=1)
    _run(emulator, ["quad", 70, 50, 2, 15, "bands", 1, 1], S360_QUAD_WAVES_PER_CU=3, EMU_CUS=1, EMU_LANE_ORDER="shuffle")
    _run(emulator, ["quad", 70, 50, 2, 15, "bands", 1, 0])


@pytest.mark.parametrize("w,h", [(3, 2), (4, 5), (5, 16), (6, 17), (17, 33), (130, 21), (41, 70), (64, 65)])
def test_lock_sizes(emulator, w, h):
    """Widths below, at and above the first width with a steady range (5); bands of 8 rows (2 compute waves)."""
    _run(emulator, ["lock", w, h, 2, 22, "random", 1])
    _run(emulator, ["lock", w, h, 2, 23, "rows0", 0], EMU_LANE_ORDER="shuffle")


@pytest.mark.parametrize("w,h", [(3, 2), (16, 16), (31, 16), (32, 17), (33, 33), (48, 20), (47, 21), (64, 41), (200, 70)])
@pytest.mark.parametrize("lpp", [3, 4])
def test_quad_sizes(emulator, w, h, lpp):
    """Widths without, with exactly one and with several interior chunks (the first one is steps 16..31 / 32..47: w >= 32 /
    48); heights of exactly one band (16 / 20 rows), one row more, two bands and one row more."""
    _run(emulator, ["quad", w, h, 2, 33, "random", 1, 1], S360_QUAD_LPP=lpp)
    _run(emulator, ["quad", w, h, 2, 33, "bands", 1, 1], S360_QUAD_LPP=lpp, S360_QUAD_WAVES_PER_CU=2, EMU_CUS=1, EMU_LANE_ORDER="rev")


@pytest.mark.parametrize("mask", ["none", "random", "bands", "rows0", "most"])
@pytest.mark.parametrize("lpp", [3, 4])
def test_quad_lds_window(emulator, mask, lpp):
    """The bilinear taps of both rounds come from the LDS window of I1-gradient texels placed per chunk, and a wave one of
    whose taps leaves it gathers from global memory for that round (the generator's +-40 px outliers and +-2 px noise make
    both happen all the time). lpp: lanes per pixel — 4 (a DPP quad, 16 rows per wave) or 3 (20 rows per wave, values
    exchanged with row shifts, lane 15 of a DPP row a passive copy)."""
    _run(emulator, ["quad", 70, 50, 2, 51, mask, 1, 1], S360_QUAD_LPP=lpp)
    _run(emulator, ["quad", 53, 40, 2, 52, mask, 0, 0], S360_QUAD_LPP=lpp, EMU_LANE_ORDER="shuffle")
    _run(emulator, ["quad", 200, 70, 2, 53, mask, 1, 1], S360_QUAD_LPP=lpp, S360_QUAD_WAVES_PER_CU=2, EMU_CUS=1)


def test_quad_lds_window_on_smooth_flows(emulator):
    """Flows as the pipeline produces them (smooth, a few pixels of disparity): nearly every round is served by the
    window — the emulator's counters say how many — and the bits are those of the raster-order loop."""
    out = _run(emulator, ["quad", 160, 120, 2, 71, "smooth", 1, 1], EMU_LANE_ORDER="rev")
    stats = [ln for ln in out.splitlines() if ln.startswith("quad window:")]
    assert stats, out
    frac = float(stats[-1].split("fallback fraction")[1].split()[0])
    assert frac < 0.05, stats[-1]
