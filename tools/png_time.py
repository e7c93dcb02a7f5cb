#!/usr/bin/env python
"""What the device PNG encoder costs per 8K frame (HIP events of the "png_encode" profile family around k_png_band / _layout /
_gather), the file's size, and the host side of s360_frame_download_png (transfer + CRC).  usage: python tools/png_time.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
from surround360_amd import render as R, synth  # noqa: E402

RIG = os.path.join(ROOT, "tests", "golden", "rig_17cam.json")
flags = dict(eqr_width=8400, eqr_height=4096, enable_top=1, enable_bottom=1, final_eqr_width=8192, final_eqr_height=8192, sharpening=0.25)
dev = torch.device("cuda", 0)
rr = synth.RigRenderer(RIG, synth.World(8192, seed=360, device=dev), 2048)
frame = rr.frame_numpy(yaw_deg=0.0, disc_deg=10.0)
del rr
torch.cuda.empty_cache()
ctx = R.Context(R.RigDescription(RIG), R.make_params(**flags))
ctx.upload_frame(*frame)
ctx.set_png_encode(True)
ctx.render(False)
buf = R.pinned_empty((int(R.lib().s360_frame_png_bound(ctx.h)),))
png = ctx.download_png(0, buf)
ctx.profile_enable(True)
n = 5
for _ in range(n):
    ctx.render(False)
ctx.synchronize()
pr = ctx.profile_get()
ctx.profile_enable(False)
t = time.perf_counter()
for _ in range(n):
    ctx.download_png(0, buf)
host = (time.perf_counter() - t) / n
print(json.dumps({"png_encode_ms_per_frame": pr["png_encode"][0] / n, "launch_groups": pr["png_encode"][1] / n, "file_bytes": int(png.size),
                  "download_png_host_ms": 1e3 * host, "frame_ms_kernels": sum(v[0] for v in pr.values()) / n}))
