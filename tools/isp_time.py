#!/usr/bin/env python
"""ISP timing on the GPU box (run from the repo root): ms per 2048x2048 raw image through s360_isp_process (host buffers
on both sides), equality with the oracle at that size, and a whole 8K frame fed from 17 raw images through
s360_frame_upload_raw (ISP on the upload stream, nothing leaves the device) followed by one render.
  python tools/isp_time.py            human-readable lines
  python tools/isp_time.py --json     one JSON line (bench.py's "isp" leg runs this in a process of its own)"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()
    import torch  # noqa: F401  (the HIP runtime before libs360, like tests/conftest.py)
    import isputil
    emulated = os.environ.get("S360_TEST_EMULATED_LIB") == "1"  # (developer check of this tool's control flow without a GPU)
    if emulated:
        from surround360_amd import _capi
        _capi.LIB_PATH = os.path.join(ROOT, "tools", "libs360_emu.so")
    from surround360_amd import isp as I, render as R
    js = isputil.CONFIG_FULL
    SZ = 256 if emulated else 2048
    raws = [isputil.bayer_frame(SZ, SZ, seed=k) for k in range(3)]
    res = {"input": "16-bit Bayer frames 2048x2048, every configuration key set, IIR sharpening on "
                    "(CameraIsp.h through Raw2Rgb.cpp:441-456)"}
    for key, bpp, dm, pipe in (("ms_per_image_bpp16_edge_aware", 16, 2, 0), ("ms_per_image_bpp8_edge_aware", 8, 2, 0),
                               ("ms_per_image_bpp8_bilinear", 8, 0, 0), ("ms_per_image_bpp16_pipe", 16, 2, 1),
                               ("ms_per_image_bpp8_pipe_fast", 8, 2, 2)):  # pipe: CameraIspPipe's arithmetic (Unpacker, --accelerate)
        isp = I.CameraIsp(I.config_from_json(js, bpp, dm, pipe=pipe), device=args.device)
        isp.get_image(raws[0])
        t = time.perf_counter()
        for k in range(6):
            out = isp.get_image(raws[k % 3])
        res[key] = round(1e3 * (time.perf_counter() - t) / 6, 3)
        isp.close()
        if not args.json:
            print("2048x2048 bpp%d dm%d pipe%d: %.2f ms per image incl. PCIe both ways" % (bpp, dm, pipe, res[key]), flush=True)
    if not args.no_cpu:
        import oracle_lib as O
        t = time.perf_counter()
        want = O.isp_run(O.isp_config_from_json(js, 16, 2), raws[0])
        res["cpu_seconds_per_image"] = round(time.perf_counter() - t, 3)
        isp = I.CameraIsp(I.config_from_json(js, 16, 2), device=args.device)
        res["checked"] = bool(np.array_equal(isp.get_image(raws[0]), want))
        isp.close()
        if not args.json:
            print("oracle 2048x2048 bpp16 dm2: %.2f s; equal to GPU: %s" % (res["cpu_seconds_per_image"], res["checked"]))
        t = time.perf_counter()
        want = O.isp_pipe_run(O.isp_config_from_json(js, 16, 2), raws[0])
        res["pipe_cpu_seconds_per_image"] = round(time.perf_counter() - t, 3)
        isp = I.CameraIsp(I.config_from_json(js, 16, 2, pipe=I.PIPE), device=args.device)
        res["pipe_checked"] = bool(np.array_equal(isp.get_image(raws[0]), want))
        isp.close()
        if not args.json:
            print("oracle 2048x2048 bpp16 pipe: %.2f s; equal to GPU: %s" % (res["pipe_cpu_seconds_per_image"], res["pipe_checked"]))
    # a whole frame from raw images: 17 x upload_raw + render (latency sweep kernel), frames back to back
    rig_path = os.path.join(ROOT, "tests", "golden", "rig_17cam.json")
    flags = dict(eqr_width=8400, eqr_height=4096, enable_top=1, enable_bottom=1, final_eqr_width=8192, final_eqr_height=8192)
    if emulated:
        import rigutil
        rig_path = rigutil.scaled_rig_json(rig_path, "/tmp/isp_time_rig_small.json", SZ / 2048.0)
        flags.update(eqr_width=504, eqr_height=252, final_eqr_width=480, final_eqr_height=480)
    rig = R.RigDescription(rig_path)
    ctx = R.Context(rig, R.make_params(**flags), device=args.device)
    isp = I.CameraIsp(I.config_from_json(js, 16, 2), device=args.device)
    isp_pipe = I.CameraIsp(I.config_from_json(js, 16, 2, pipe=I.PIPE), device=args.device)  # what the reference's Unpacker runs
    try:
        def frame(isp=isp):
            for k in range(14):
                ctx.upload_raw(isp, k, raws[k % 3])
            ctx.upload_raw(isp, -1, raws[0])
            ctx.upload_raw(isp, -2, raws[1])
            ctx.render(False)
        frame()
        ctx.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            frame()
        ctx.synchronize()
        res["frame_from_raw_ms"] = round(1e3 * (time.perf_counter() - t) / 3, 2)
        res["frame_from_raw_note"] = ("17 x s360_frame_upload_raw (ISP on the upload stream, the result stays on the device) + "
                                      "one 8K frame render, latency sweep kernel, frames back to back")
        frame(isp_pipe)
        ctx.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            frame(isp_pipe)
        ctx.synchronize()
        res["frame_from_raw_pipe_ms"] = round(1e3 * (time.perf_counter() - t) / 3, 2)  # (the accelerated pipeline's arithmetic)
        if not args.json:
            print("8K frame from 17 raw images (ISP + render): %.1f ms; with CameraIspPipe's arithmetic: %.1f ms" % (
                res["frame_from_raw_ms"], res["frame_from_raw_pipe_ms"]))
    finally:
        isp.close()
        isp_pipe.close()
        ctx.close()
    if args.json:
        print(json.dumps(res))


if __name__ == "__main__":
    main()
