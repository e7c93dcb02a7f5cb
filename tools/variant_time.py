#!/usr/bin/env python
"""A/B timing of kernel variants that are selected by an environment variable read once per process (e.g.
S360_LOCK_PEEL=1): renders bench.py's first synthetic 8K frame alone with the latency sweep kernel and prints one JSON
line {ms, sweep_ms, us_per_diagonal_step, sha1}. Run it once per variant (the variable set in the environment) and
compare; equal sha1 = byte-identical stereo equirects. bench.py's "variants" leg does exactly that, each run in a
process of its own.
  S360_LOCK_PEEL=1 python tools/variant_time.py --json"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import torch
    from surround360_amd import render as R, synth
    rig_path = os.path.join(ROOT, "tests", "golden", "rig_17cam.json")
    torch.cuda.set_device(args.device)
    dev = torch.device("cuda", args.device)
    world = synth.World(4096, seed=360, device=dev)
    frame = synth.RigRenderer(rig_path, world, 2048).frame_numpy(yaw_deg=0.0, disc_deg=10.0)
    del world
    torch.cuda.empty_cache()
    flags = dict(eqr_width=8400, eqr_height=4096, enable_top=1, enable_bottom=1, final_eqr_width=8192, final_eqr_height=8192)
    ctx = R.Context(R.RigDescription(rig_path), R.make_params(**flags), device=args.device)
    try:
        ctx.set_sweep_mode("latency")
        ctx.upload_frame(*frame)
        ctx.render(False)
        ctx.synchronize()
        t = time.perf_counter()
        for _ in range(args.reps):
            ctx.render(False)
        ctx.synchronize()
        ms = 1e3 * (time.perf_counter() - t) / args.reps
        ctx.profile_enable(True)
        for _ in range(args.reps):
            ctx.render(False)
        ctx.synchronize()
        prof = ctx.profile_get()
        sweep_ms = prof.get("flow_sweep", (0.0, 0))[0] / args.reps
        res = {"ms": round(ms, 3), "sweep_ms": round(sweep_ms, 3),
               "sha1": hashlib.sha1(ctx.download_equirect().tobytes()).hexdigest(),
               "env": {k: v for k, v in os.environ.items() if k.startswith("S360_")}}
    finally:
        ctx.close()
    print(json.dumps(res) if args.json else res)


if __name__ == "__main__":
    main()
