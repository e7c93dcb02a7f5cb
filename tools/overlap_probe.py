"""Developer tool: how does the cap on the throughput sweep's persistent waves (S360_QUAD_WAVES_PER_CU, read at every launch)
change what two overlapped contexts deliver? The sweep's waves hold 204-226 VGPRs each: two of them per SIMD leave the other
context's kernels ~60 registers per lane on that SIMD, i.e. nothing — with fewer sweep waves resident the other context's
stencil / warp kernels can actually run beside the sweeps. One process, the bench's shape (contexts x slots, one submitting
thread per context), frames/s per cap.   python tools/overlap_probe.py [contexts] [slots] [caps, comma-separated]"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from surround360_amd import render as R, synth  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 2
S = int(sys.argv[2]) if len(sys.argv) > 2 else 22
caps = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "8,6,5,4,3,8").split(",")]
STEPS = int(os.environ.get("PROBE_STEPS", "6"))
FLAGS = dict(eqr_width=8400, eqr_height=4096, enable_top=1, enable_bottom=1, final_eqr_width=8192, final_eqr_height=8192, sharpening=0.25)
rig_path = os.path.join(ROOT, "tests", "golden", "rig_17cam.json")
world = synth.World(4096, seed=360, device="cuda")
rr = synth.RigRenderer(rig_path, world, 2048)
frames = [rr.frame_numpy(yaw_deg=0.2 * k, disc_deg=10.0 + 0.5 * k) for k in range(8)]
del rr, world
torch.cuda.empty_cache()
rig = R.RigDescription(rig_path)
ctxs = [R.Context(rig, R.make_params(**FLAGS)) for _ in range(F)]
for k, c in enumerate(ctxs):
    c.set_frame_slots(S)
    for j in range(S):
        c.select_frame_slot(j)
        c.upload_frame(*frames[(k * S + j) % len(frames)])
    c.set_sweep_mode("throughput")
pools = [ThreadPoolExecutor(max_workers=1) for _ in range(F)]


def run(steps):
    futs = [pools[i % F].submit(ctxs[i % F].render_batch, False) for i in range(steps)]
    for f in futs:
        f.result()
    for c in ctxs:
        c.synchronize()


run(2 * F)
for cap in caps:
    os.environ["S360_QUAD_WAVES_PER_CU"] = str(cap)
    run(F)
    t = time.perf_counter()
    run(STEPS * F)
    dt = time.perf_counter() - t
    print("%d contexts x %d slots, %2d sweep waves per CU: %.2f frames/s (%.1f ms per batch)" % (F, S, cap, STEPS * F * S / dt, 1e3 * dt / (STEPS * F)), flush=True)
for c in ctxs:
    c.close()
