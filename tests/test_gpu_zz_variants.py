"""The two sweep kernels against each other on hardware, through the whole flow path and at sizes above the other tests': the
throughput kernel (banded, LDS window, persistent waves) must give byte-identical flows to the latency kernel. Each side
runs in a process of its own. (The kernel variants that round 2 left behind run-time switches were timed by that round's
bench and adopted or deleted, and so was this round's: DESIGN.md section 5.)"""
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


def test_throughput_equals_latency_kernel(default_throughput_digest):
    assert _flows_digest() == default_throughput_digest
