"""The library's whole PixFlow path without a GPU: FlowEngine (flow.hip) with every kernel it launches — entry resize, grey /
alpha / motion, pyramids, gradients, the coarse search, 15-tap blurs to sweep records, the banded sweeps, medians,
diffusion, level and final upscales — compiled for the CPU over tools/hip_wave_shim (tools/flow_emulate.cpp) and compared
bit for bit with the oracle's computeOpticalFlow, which is itself pinned to the reference's PixFlow.h
(tests/test_cpu_refpin.py). What the GPU parity tests check on hardware (tests/test_gpu_flow.py) is checked here for the
sources' logic: indexing, batching over shared images, the pyramid schedule, hand-offs, both sweep kernels."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from surround360_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HINT = {"UNKNOWN": 0, "RIGHT": 1, "DOWN": 2, "LEFT": 3, "UP": 4}


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tools"), "-s", "libflow_emu.so"])
    lib = C.CDLL(os.path.join(ROOT, "tools", "libflow_emu.so"))

    def run(images, alg, hint, pairs, mode, prev_images=None, prev_flows=None):
        imgs = np.ascontiguousarray(np.stack(images), np.uint8)
        n, h, w, _ = imgs.shape
        i0 = (C.c_int * len(pairs))(*[p[0] for p in pairs])
        i1 = (C.c_int * len(pairs))(*[p[1] for p in pairs])
        out = np.zeros((len(pairs), h, w, 2), np.float32)
        err = C.create_string_buffer(512)
        pi = np.ascontiguousarray(np.stack(prev_images), np.uint8) if prev_images is not None else None
        pf = np.ascontiguousarray(np.stack(prev_flows), np.float32) if prev_flows is not None else None
        rc = lib.emu_flow_batch(imgs.ctypes.data_as(C.c_void_p), n, w, h, alg.encode(), HINT[hint], len(pairs), i0, i1,
                                pi.ctypes.data_as(C.c_void_p) if pi is not None else None,
                                pf.ctypes.data_as(C.c_void_p) if pf is not None else None, mode,
                                out.ctypes.data_as(C.c_void_p), err, 512)
        assert rc == 0, err.value
        return out
    run.lib = lib
    return run


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("alg,hint,w,h,seed", [("pixflow_low", "LEFT", 128, 112, 7), ("pixflow_low", "DOWN", 201, 75, 11),
                                               ("pixflow_search_20", "RIGHT", 128, 112, 7), ("pixflow_search_20", "UP", 90, 150, 5),
                                               ("pixflow_low", "UNKNOWN", 97, 61, 2)])
@pytest.mark.parametrize("mode", [2, 3])
def test_flow_engine_equals_oracle(emu, alg, hint, w, h, seed, mode):
    i0, i1 = synth.flow_pair(w, h, seed=seed)
    got = emu([i0, i1], alg, hint, [(0, 1)], mode)[0]
    want = O.compute_optical_flow(i0, i1, alg, hint)
    assert np.array_equal(_bits(got), _bits(want)), "max abs diff %g" % np.abs(got - want).max()


@pytest.mark.parametrize("mode", [2, 3])
def test_batch_over_shared_images_masked_rows_and_previous_frame(emu, mode):
    """Three images, four flows over them in one batch (each image is I0 of one flow and I1 of another, as the 14 side
    pairs share their 28 images); the upper third of one image below the alpha threshold (the pole flows' case); then the
    same batch regularised against a previous frame (PixFlow.h:109-117, 185-193)."""
    w, h = 120, 100
    a, b = synth.flow_pair(w, h, seed=21)
    c, _ = synth.flow_pair(w, h, seed=22)
    b = b.copy()
    b[: h // 3, :, 3] = 0
    pairs = [(0, 1), (1, 0), (1, 2), (2, 0)]
    imgs = [a, b, c]
    first = emu(imgs, "pixflow_low", "LEFT", pairs, mode)
    for k, (p, q) in enumerate(pairs):
        want = O.compute_optical_flow(imgs[p], imgs[q], "pixflow_low", "LEFT")
        assert np.array_equal(_bits(first[k]), _bits(want)), (mode, k)
    nxt = [np.roll(im, 2, axis=1) for im in imgs]
    second = emu(nxt, "pixflow_low", "LEFT", pairs, mode, prev_images=imgs, prev_flows=list(first))
    for k, (p, q) in enumerate(pairs):
        want = O.compute_optical_flow(nxt[p], nxt[q], "pixflow_low", "LEFT", first[k], imgs[p], imgs[q])
        assert np.array_equal(_bits(second[k]), _bits(want)), (mode, k)


@pytest.mark.parametrize("sw,sh,dw,dh,B", [(333, 200, 300, 180, 16), (150, 97, 135, 87, 4), (70, 40, 63, 36, 3), (9, 7, 8, 6, 2)])
def test_pyramid_resize_tiles(emu, sw, sh, dw, dh, B):
    """The pyramid's tiled resize (k_resize_linear_f32c1_tiled) with many tiles per plane, several plane groups and the
    XCD-aware tile order active (>= 64 workgroups), ragged edges, and the smallest levels (which take the one-thread-per-
    pixel kernel) against the oracle's resize."""
    rng = np.random.RandomState(sw + dh)
    src = rng.rand(B, sh, sw).astype(np.float32)
    out = np.zeros((B, dh, dw), np.float32)
    assert emu.lib.emu_resize_linear_planes(src.ctypes.data_as(C.c_void_p), sw, sh, B, dw, dh, out.ctypes.data_as(C.c_void_p)) == 0
    for b in range(B):
        want = O.resize_linear_f32(src[b], dw, dh)
        assert np.array_equal(out[b].view(np.uint32), np.asarray(want, np.float32).reshape(dh, dw).view(np.uint32)), "plane %d" % b
