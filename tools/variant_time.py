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
    rig_path = os.path.join(ROOT, "tests", "golden", "rig_17cam.json")
    flags = dict(eqr_width=8400, eqr_height=4096, enable_top=1, enable_bottom=1, final_eqr_width=8192, final_eqr_height=8192)
    emulated = os.environ.get("S360_TEST_EMULATED_LIB") == "1"  # (developer check of this tool's control flow without a GPU)
    if emulated:
        from surround360_amd import _capi
        _capi.LIB_PATH = os.path.join(ROOT, "tools", "libs360_emu.so")
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import rigutil
        rig_path = rigutil.scaled_rig_json(rig_path, "/tmp/variant_time_rig_small.json", 256 / 2048.0)
        flags.update(eqr_width=504, eqr_height=252, final_eqr_width=480, final_eqr_height=480)
    from surround360_amd import render as R, synth
    if not emulated:
        torch.cuda.set_device(args.device)
    dev = torch.device("cpu") if emulated else torch.device("cuda", args.device)
    world = synth.World(512 if emulated else 4096, seed=360, device=dev)
    rr = synth.RigRenderer(rig_path, world, 256 if emulated else 2048)
    S = max(1, args.slots)
    frames = [rr.frame_numpy(yaw_deg=0.2 * k, disc_deg=10.0 + 0.5 * k) for k in range(min(S, 3))]
    del rr, world
    if not emulated:
        torch.cuda.empty_cache()
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
               "kernel_ms_per_frame": {k: round(v[0] / args.reps / S, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])},
               "env": {k: v for k, v in os.environ.items() if k.startswith("S360_")}}
    finally:
        ctx.close()
    print(json.dumps(res) if args.json else res)


if __name__ == "__main__":
    main()
