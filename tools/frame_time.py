"""Developer tool: one 8K frame (BASELINE configs[2], the reference's 8k preset) at a time with the latency sweep kernel, and a
short stream with temporal state and frame pipelining, on a library given on the command line — for timing two builds of
libs360.so against each other on the same box:  python tools/frame_time.py [path/to/libs360.so] [frames]"""
import os
import sys
import time

import numpy as np
import torch  # noqa: F401  (its HIP runtime must load first)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from surround360_amd import _capi  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1]:
    _capi.LIB_PATH = os.path.abspath(sys.argv[1])
from surround360_amd import render as R, synth  # noqa: E402

FLAGS = dict(eqr_width=8400, eqr_height=4096, enable_top=1, enable_bottom=1, final_eqr_width=8192, final_eqr_height=8192,
             sharpening=0.25)
rig_path = os.path.join(ROOT, "tests", "golden", "rig_17cam.json")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
world = synth.World(4096, seed=360, device="cuda")
rr = synth.RigRenderer(rig_path, world, 2048)
frames = [rr.frame_numpy(yaw_deg=0.2 * k, disc_deg=10.0 + 0.5 * k) for k in range(6)]
del rr, world
torch.cuda.empty_cache()
rig = R.RigDescription(rig_path)
ctx = R.Context(rig, R.make_params(**FLAGS))
ctx.set_sweep_mode("latency")
ctx.upload_frame(*frames[0])
for _ in range(2):
    ctx.render(False)
ctx.synchronize()
t = time.perf_counter()
for _ in range(4):
    ctx.render(False)
ctx.synchronize()
single = 1e3 * (time.perf_counter() - t) / 4
ctx.profile_enable(True)
for _ in range(2):
    ctx.render(False)
ctx.synchronize()
pr = ctx.profile_get()
ctx.profile_enable(False)
ctx.set_frame_pipelining(True)
t = None
for k in range(n):
    if k == 6:
        ctx.synchronize()
        t = time.perf_counter()
    ctx.upload_frame(*frames[k % 6 if (k // 6) % 2 == 0 else 5 - k % 6])
    ctx.render(k > 0)
ctx.synchronize()
stream = 1e3 * (time.perf_counter() - t) / (n - 6)
digest = int(np.asarray(ctx.download_equirect(), np.uint64).sum())
print("%s: single frame %.2f ms (sweeps %.2f ms), stream %.2f ms per frame, checksum %d" % (
    os.path.basename(_capi.LIB_PATH), single, pr.get("flow_sweep", (0, 0))[0] / 2, stream, digest))
print("  kernel families, ms per frame: " + ", ".join("%s %.3f" % (k, v[0] / 2) for k, v in sorted(pr.items(), key=lambda kv: -kv[1][0])))
ctx.close()
