#!/usr/bin/env python
"""A/B timing of kernel variants that are selected by an environment variable read once per process (e.g.
S360_LOCK_PEEL=1). --slots 1 (default): bench.py's first synthetic 8K frame alone with the latency sweep kernel;
--slots S > 1: a batch of S frames in one context with the throughput sweep kernel (what a bench step is). Prints one
JSON line {ms (per frame), sweep_ms, sha1}. Run it once per variant (the variable set in the environment) and compare;
equal sha1 = byte-identical stereo equirects. bench.py's "variants" leg does exactly that, each run in a process of its
own, after it has released its own contexts.
  S360_LOCK_PEEL=1 python tools/variant_time.py --json
  S360_QUAD_PEEL=1 python tools/variant_time.py --json --slots 12"""
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
    ap.add_argument("--slots", type=int, default=1)
    args = ap.parse_args()
    import torch
    from surround360_amd import render as R, synth
    rig_path = os.path.join(ROOT, "tests", "golden", "rig_17cam.json")
    torch.cuda.set_device(args.device)
    dev = torch.device("cuda", args.device)
    world = synth.World(4096, seed=360, device=dev)
    rr = synth.RigRenderer(rig_path, world, 2048)
    S = max(1, args.slots)
    frames = [rr.frame_numpy(yaw_deg=0.2 * k, disc_deg=10.0 + 0.5 * k) for k in range(min(S, 3))]
    del rr, world
    torch.cuda.empty_cache()
    flags = dict(eqr_width=8400, eqr_height=4096, enable_top=1, enable_bottom=1, final_eqr_width=8192, final_eqr_height=8192)
    ctx = R.Context(R.RigDescription(rig_path), R.make_params(**flags), device=args.device)
    try:
        if S > 1:
            ctx.set_frame_slots(S)
            for j in range(S):
                ctx.select_frame_slot(j)
                ctx.upload_frame(*frames[j % len(frames)])
            ctx.set_sweep_mode("throughput")
            go = lambda: ctx.render_batch(False)  # noqa: E731
        else:
            ctx.set_sweep_mode("latency")
            ctx.upload_frame(*frames[0])
            go = lambda: ctx.render(False)  # noqa: E731
        go()
        ctx.synchronize()
        t = time.perf_counter()
        for _ in range(args.reps):
            go()
        ctx.synchronize()
        ms = 1e3 * (time.perf_counter() - t) / args.reps / S
        ctx.profile_enable(True)
        for _ in range(args.reps):
            go()
        ctx.synchronize()
        prof = ctx.profile_get()
        sweep_ms = prof.get("flow_sweep", (0.0, 0))[0] / args.reps / S
        h = hashlib.sha1()
        for j in range(min(S, 3)):  # (the other slots hold the same three frames again)
            if S > 1:
                ctx.select_frame_slot(j)
            h.update(ctx.download_equirect().tobytes())
        res = {"slots": S, "ms_per_frame": round(ms, 3), "sweep_ms_per_frame": round(sweep_ms, 3), "sha1": h.hexdigest(),
               "env": {k: v for k, v in os.environ.items() if k.startswith("S360_")}}
    finally:
        ctx.close()
    print(json.dumps(res) if args.json else res)


if __name__ == "__main__":
    main()
