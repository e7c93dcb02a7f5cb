"""The one kernel build left behind a run-time switch, on hardware: S360_QUAD_AHEAD=0 (throughput sweep requesting
the next chunk's LDS window between the chunks instead of four steps early) must give byte-identical flows to the default. The
switch is read once per process, so each side of the comparison runs in a process of its own. (Round 2's other variants
were timed by that round's bench, adopted or deleted: DESIGN.md section 5.)"""
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
def default_throughput_digest(s360lib):
    return _flows_digest(TEST_SWEEP_MODE="throughput")


def test_throughput_kernel_window_request_between_chunks(default_throughput_digest):
    assert _flows_digest(TEST_SWEEP_MODE="throughput", S360_QUAD_AHEAD="0") == default_throughput_digest


def test_throughput_equals_latency_kernel(default_throughput_digest):
    assert _flows_digest() == default_throughput_digest
