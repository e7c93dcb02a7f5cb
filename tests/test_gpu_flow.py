"""GPU parity of PixFlow::computeOpticalFlow (PixFlow.h:81-183) through the C ABI, bit-exact vs the oracle."""
import numpy as np
import pytest

from surround360_amd import render as R, synth

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def ctx(gpu_rig):
    c = R.Context(gpu_rig, R.make_params(eqr_width=1008, eqr_height=504))
    yield c
    c.close()


def test_flow_levels_bit_exact(ctx, oracle):
    i0, i1 = synth.flow_pair(200, 168, seed=3)
    want_final, want_levels = oracle.compute_optical_flow(i0, i1, "pixflow_low", "LEFT", want_levels=True)
    buf, n = ctx.debug_flow_levels(i0, i1, "pixflow_low", "LEFT")
    assert n == len(want_levels)
    off = 0
    for li, wl in enumerate(want_levels):
        got = buf[off:off + wl.size].reshape(wl.shape)
        off += wl.size
        assert np.array_equal(bits(got), bits(wl)), "level %d (coarsest first) differs: max abs %g" % (
            li, np.abs(got - wl).max())


@pytest.mark.parametrize("w,h,seed", [(160, 192, 1), (297, 444, 2), (333, 257, 5)])
def test_flow_bit_exact(ctx, oracle, w, h, seed):
    i0, i1 = synth.flow_pair(w, h, seed=seed)
    for hint, a, b in (("LEFT", i0, i1), ("RIGHT", i1, i0)):
        got = ctx.compute_optical_flow(a, b, "pixflow_low", hint)
        want = oracle.compute_optical_flow(a, b, "pixflow_low", hint)
        assert np.array_equal(bits(got), bits(want)), "max abs diff %g" % np.abs(got - want).max()


def test_flow_search20_bit_exact(ctx, oracle):
    i0, i1 = synth.flow_pair(180, 200, seed=11)
    for hint in ("LEFT", "DOWN", "UNKNOWN"):
        got = ctx.compute_optical_flow(i0, i1, "pixflow_search_20", hint)
        want = oracle.compute_optical_flow(i0, i1, "pixflow_search_20", hint)
        assert np.array_equal(bits(got), bits(want)), hint


def test_flow_temporal_bit_exact(ctx, oracle):
    a0, a1 = synth.flow_pair(192, 224, seed=21)
    b0, b1 = synth.flow_pair(192, 224, seed=21, max_disp=6.0)
    prev = oracle.compute_optical_flow(a0, a1, "pixflow_low", "LEFT")
    got = ctx.compute_optical_flow(b0, b1, "pixflow_low", "LEFT", prev_flow=prev, prev_i0=a0, prev_i1=a1)
    want = oracle.compute_optical_flow(b0, b1, "pixflow_low", "LEFT", prev_flow=prev, prev_i0=a0, prev_i1=a1)
    assert np.array_equal(bits(got), bits(want)), "max abs diff %g" % np.abs(got - want).max()


def test_flow_batch_matches_single(ctx):
    pairs = [synth.flow_pair(160, 176, seed=s) for s in (31, 32, 33)]
    i0 = np.stack([p[0] for p in pairs])
    i1 = np.stack([p[1] for p in pairs])
    batch = ctx.compute_optical_flow(i0, i1, "pixflow_low", "LEFT")
    for k, (a, b) in enumerate(pairs):
        single = ctx.compute_optical_flow(a, b, "pixflow_low", "LEFT")
        assert np.array_equal(bits(batch[k]), bits(single))


def test_unknown_algorithm_raises(ctx):
    i0, i1 = synth.flow_pair(160, 176, seed=1)
    with pytest.raises(R.VrCamException):
        ctx.compute_optical_flow(i0, i1, "no_such_flow")


def test_identical_images_near_zero_flow(ctx):
    i0, _ = synth.flow_pair(192, 192, seed=9)
    f = ctx.compute_optical_flow(i0, i0)
    assert np.abs(f).max() < 0.5


@pytest.mark.parametrize("mode", ["throughput", "latency"])
def test_sweep_modes_bit_exact(gpu_rig, oracle, mode):
    """Both sweep kernels (lockstep = latency, quad = throughput) reproduce the raster-order sweeps exactly,
    including sizes that are not multiples of the band height and the temporal path."""
    c = R.Context(gpu_rig, R.make_params(eqr_width=1008, eqr_height=504))
    c.set_sweep_mode(mode)
    try:
        for (w, h, seed) in [(297, 444, 2), (333, 257, 5), (160, 130, 9), (520, 300, 12)]:
            i0, i1 = synth.flow_pair(w, h, seed=seed)
            got = c.compute_optical_flow(i0, i1, "pixflow_low", "LEFT")
            want = oracle.compute_optical_flow(i0, i1, "pixflow_low", "LEFT")
            assert np.array_equal(bits(got), bits(want)), "%s %dx%d: max abs diff %g" % (mode, w, h, np.abs(got - want).max())
        with pytest.raises(R.S360Error):
            c.set_sweep_mode("fastest")
    finally:
        c.close()


def test_ieee_division_path_bit_exact(gpu_rig, oracle, monkeypatch):
    """S360_SWEEP_DIV=ieee disables the verified fast division / sqrt of the sweep kernels: same bits."""
    monkeypatch.setenv("S360_SWEEP_DIV", "ieee")
    c = R.Context(gpu_rig, R.make_params(eqr_width=1008, eqr_height=504))
    try:
        i0, i1 = synth.flow_pair(200, 168, seed=3)
        got = c.compute_optical_flow(i0, i1, "pixflow_low", "RIGHT")
        want = oracle.compute_optical_flow(i0, i1, "pixflow_low", "RIGHT")
        assert np.array_equal(bits(got), bits(want))
    finally:
        c.close()


@pytest.mark.parametrize("w,h", [(4, 4), (5, 7), (20, 49), (49, 60), (51, 33)])
def test_flow_below_the_pyramid_threshold(ctx, oracle, w, h):
    """Inputs whose x0.5 downscale is at or below kPyrMinImageSize = 24: buildPyramid (PixFlow.h:477-491) returns one
    level and the reference still runs — so does the library (round 2 rejected anything below 50 px)."""
    i0, i1 = synth.flow_pair(w, h, seed=100 * w + h)
    for alg in ("pixflow_low", "pixflow_search_20"):
        got = ctx.compute_optical_flow(i0, i1, alg, "LEFT")
        want = oracle.compute_optical_flow(i0, i1, alg, "LEFT")
        assert np.array_equal(bits(got), bits(want)), (alg, w, h)


def test_fourteen_threads_share_one_context(ctx, oracle):
    """TRSP:320-335 runs renderStereoPanoramaChunksThread on 14 std::threads, each with its own flow operator
    (NovelView.cpp:281-298). INTEGRATION.md lets all of them call ONE s360_ctx: entry points lock the context
    (SURVEY 8b "thread-safe per ctx"), so 14 concurrent callers with 14 different pairs — and different sizes, so that
    the shared scratch buffers are re-sized under their feet if the lock were missing — all get the oracle's flow."""
    import threading
    sizes = [(160 + 8 * (k % 5), 176 + 6 * (k % 3)) for k in range(14)]
    pairs = [synth.flow_pair(w, h, seed=700 + k) for k, (w, h) in enumerate(sizes)]
    hints = ["LEFT", "RIGHT"]
    got = [None] * 14
    errors = []
    start = threading.Barrier(14)

    def work(k):
        try:
            start.wait()
            for rep in range(2):  # two rounds: the second interleaves with other threads' first
                got[k] = ctx.compute_optical_flow(pairs[k][0], pairs[k][1], "pixflow_low", hints[k % 2])
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    th = [threading.Thread(target=work, args=(k,)) for k in range(14)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for k in range(14):
        want = oracle.compute_optical_flow(pairs[k][0], pairs[k][1], "pixflow_low", hints[k % 2])
        assert np.array_equal(bits(got[k]), bits(want)), "thread %d" % k


@pytest.mark.parametrize("bx", [32, 16, 8, 4])
def test_median_tile_shapes_give_the_same_flow(bx, oracle, rig_json, tmp_path):
    """medianBlur(flow, 5) runs with one of four tile shapes per pyramid level (median.hip, launch_median5_c2); S360_MEDIAN_BX forces
    one shape for every level. The variable is read once per process, so each shape runs in a process of its own; the flow must be
    the oracle's bit for bit whatever the shape."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (sizes: the replay of this file on the CPU emulation sets TEST_FLOW_SIZES; the entry downscale halves the image, and only
    # levels at least 64 wide take the tiled kernel, so the first level of a 150-wide pair still walks the forced shape)
    w, h = [int(v) for v in os.environ.get("TEST_FLOW_SIZES", "297x444").split(",")[0].split("x")]
    code = ("import os, sys, numpy as np; sys.path.insert(0, %r)\n"
            "from surround360_amd import _capi, render as R, synth\n"
            "if os.environ.get('S360_TEST_EMULATED_LIB') == '1':  # tests/conftest.py's developer switch, for this child too\n"
            "    _capi.LIB_PATH = os.environ.get('S360_TEST_EMULATED_LIB_PATH') or os.path.join(%r, 'tools', 'libs360_emu.so')\n"
            "c = R.Context(R.RigDescription(sys.argv[2]), R.make_params(eqr_width=1008, eqr_height=504))\n"
            "i0, i1 = synth.flow_pair(int(sys.argv[3]), int(sys.argv[4]), seed=2)\n"
            "np.save(sys.argv[1], c.compute_optical_flow(i0, i1, 'pixflow_low', 'LEFT'))\n" % (root, root))
    out = str(tmp_path / "flow.npy")
    subprocess.run([sys.executable, "-c", code, out, rig_json, str(w), str(h)], check=True,
                   env=dict(os.environ, S360_MEDIAN_BX=str(bx)), timeout=600)
    i0, i1 = synth.flow_pair(w, h, seed=2)
    want = oracle.compute_optical_flow(i0, i1, "pixflow_low", "LEFT")
    assert np.array_equal(bits(np.load(out)), bits(want))
