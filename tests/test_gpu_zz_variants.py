"""Kernel variants behind run-time switches, on hardware: a variant must give byte-identical flows. The switches are read
once per process, so each side of the comparison runs in a process of its own. Sorted last: these variants were brought
to bit-exactness on the CPU emulation (tests/test_cpu_sweep_emulation.py) after the round's GPU minutes were spent."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SNIPPET = r'''
import hashlib, os, sys
sys.path.insert(0, %(root)r)
import torch  # noqa: F401
if os.environ.get("S360_TEST_EMULATED_LIB") == "1":  # (tests/test_cpu_library_emulation.py: the same comparison on the CPU)
    from surround360_amd import _capi
    _capi.LIB_PATH = os.path.join(%(root)r, "tools", "libs360_emu.so")
from surround360_amd import render as R, synth
rig = R.RigDescription(os.path.join(%(root)r, "tests", "golden", "rig_17cam.json"))
ctx = R.Context(rig, R.make_params())
ctx.set_sweep_mode(os.environ.get("TEST_SWEEP_MODE", "latency"))
h = hashlib.sha1()
sizes = [tuple(int(v) for v in s.split("x")) for s in os.environ.get("TEST_FLOW_SIZES", "333x444,1214x700").split(",")]
for seed, (w, hh) in enumerate(sizes, 1):
    i0, i1 = synth.flow_pair(w, hh, seed=seed)
    i0[: hh // 3, :, 3] = 0  # a band of pixels below the alpha threshold, like the pole flows
    for alg in ("pixflow_low", "pixflow_search_20"):
        h.update(ctx.compute_optical_flow(i0, i1, alg, "LEFT").tobytes())
        h.update(ctx.compute_optical_flow(i1, i0, alg, "RIGHT").tobytes())
ctx.close()
print("SHA1", h.hexdigest())
'''


def _flows_digest(**env):
    """(env: the variant's switch; TEST_SWEEP_MODE; on the CPU also S360_TEST_EMULATED_LIB=1 and smaller TEST_FLOW_SIZES)"""
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", SNIPPET % {"root": ROOT}], capture_output=True, text=True, env=e, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return [ln for ln in r.stdout.splitlines() if ln.startswith("SHA1")][-1]


@pytest.fixture(scope="module")
def default_digest(s360lib):
    return _flows_digest()


def test_lock_kernel_with_peeled_steady_state(default_digest):
    assert _flows_digest(S360_LOCK_PEEL="1") == default_digest


@pytest.fixture(scope="module")
def default_throughput_digest(s360lib):
    return _flows_digest(TEST_SWEEP_MODE="throughput")


@pytest.mark.parametrize("level", ["1", "2"])
def test_quad_kernel_with_peeled_interior_chunks(default_throughput_digest, level):
    """S360_QUAD_PEEL=1 (specialised interior chunks, prefetch arrays in registers) and =2 (plus the round-2 texel exchange)
    against the default build of the throughput kernel."""
    assert _flows_digest(TEST_SWEEP_MODE="throughput", S360_QUAD_PEEL=level) == default_throughput_digest


@pytest.mark.parametrize("level", ["1", "2"])
def test_throughput_kernel_with_three_lanes_per_pixel(default_throughput_digest, level):
    """S360_SWEEP_TRI=1 (sweep_tri.hip) and =2 (with the round-2 texel exchange) against the default throughput kernel."""
    assert _flows_digest(TEST_SWEEP_MODE="throughput", S360_SWEEP_TRI=level) == default_throughput_digest


@pytest.mark.parametrize("nw", ["2", "8"])
def test_lock_kernel_with_other_workgroup_heights(default_digest, nw):
    """S360_LOCK_NW: 2 or 8 compute waves (8 or 32 rows) per workgroup instead of 4, with the peeled steps as well."""
    assert _flows_digest(S360_LOCK_NW=nw, S360_LOCK_PEEL="1") == default_digest
