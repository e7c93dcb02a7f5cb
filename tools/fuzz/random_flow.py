"""s360_compute_optical_flow with random sizes (4..300 squared), alpha masks, direction hints, algorithms and previous-frame
state on an emulated build of the library, both sweep kernels, against the oracle bit for bit.
usage: python tools/fuzz/random_flow.py <libs360 build> <seed> <cases>"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
from surround360_amd import _capi
_capi.LIB_PATH = sys.argv[1]
from surround360_amd import render as R, synth
import numpy as np
import oracle_lib as O
random.seed(int(sys.argv[2])); n = int(sys.argv[3])
rig = R.RigDescription(ROOT + '/tests/golden/rig_17cam.json')
ctx = R.Context(rig, R.make_params(eqr_width=1008, eqr_height=504))
bad = 0
HINTS = ["UNKNOWN", "RIGHT", "DOWN", "LEFT", "UP"]
for i in range(n):
    w = random.choice([random.randint(4, 40), random.randint(4, 300), 16 * random.randint(1, 12) + random.choice([-1, 0, 1])])
    h = random.choice([random.randint(4, 40), random.randint(4, 300), 16 * random.randint(1, 12) + random.choice([-1, 0, 1]), 20 * random.randint(1, 8) + random.choice([-1, 0, 1])])
    w, h = max(w, 4), max(h, 4)
    rng = np.random.default_rng(1000 + i)
    i0, i1 = synth.flow_pair(max(w, 8), max(h, 8), seed=1000 + i)
    i0, i1 = np.ascontiguousarray(i0[:h, :w]), np.ascontiguousarray(i1[:h, :w])
    m = random.random()
    if m < 0.3:   # rows / columns of transparent pixels (pole-like masks)
        i1[rng.integers(0, h):, :, 3] = 0
    elif m < 0.5:
        i0[:, :rng.integers(1, w + 1), 3] = 0
    elif m < 0.6:
        i1[..., 3] = rng.integers(0, 256, (h, w), dtype=np.uint8)
    alg = random.choice(["pixflow_low", "pixflow_low", "pixflow_search_20"])
    hint = random.choice(HINTS)
    prev = None
    if random.random() < 0.4:
        pf = (rng.standard_normal((h, w, 2)) * 2).astype(np.float32)
        p0, p1 = synth.flow_pair(max(w, 8), max(h, 8), seed=2000 + i)
        prev = (pf, np.ascontiguousarray(p0[:h, :w]), np.ascontiguousarray(p1[:h, :w]))
    want = O.compute_optical_flow(i0, i1, alg, hint, *(prev or ()))
    for mode in ("latency", "throughput"):
        ctx.set_sweep_mode(mode)
        got = ctx.compute_optical_flow(i0, i1, alg, hint, *(prev or ()))
        if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
            bad += 1
            print("DIFFER", i, mode, w, h, alg, hint, "prev" if prev else "", "mask %.2f" % m, flush=True)
print("done: %d cases x 2 sweep kernels, %d differ" % (n, bad))
