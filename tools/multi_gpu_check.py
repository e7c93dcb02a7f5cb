#!/usr/bin/env python
"""Hardware-day checklist for N > 1 GPUs (tools/gpu_multi.sh runs it): the steps of the multi-GPU path in the order in which
they depend on each other, each under its own timeout, each in a process of its own — a hang or a crash names its step instead
of taking the rest with it. No curve is measured here; `bench.py --gpus N` (last step) prints the one line the driver reads.
  1. which librccl the library resolves (S360_RCCL_LIB overrides), N visible devices
  2. s360_comm_init_all over N contexts (one per device) + s360_comm_loopback on every rank (a grouped send + recv of the rank
     to itself on its own stream: communicator, stream ordering and ncclGroupEnd without any peer traffic)
  3. a grouped exchange between the ranks: one small 8K-less frame sharded over the N GPUs by host/TestRenderStereoPanorama
     --num_gpus N (pairs + pole units, both RCCL exchanges), two chained frames, every file against the REFERENCE program's
     digests (tests/golden/refprogram_golden.json)
  4. python bench.py --gpus N --steps 4 --warmup 2   (no launcher: the script starts itself under torch.distributed.run, the
     form of the driver's recorded command; tests/test_cpu_parallel.py runs the same form on the emulation)
Usage: python tools/multi_gpu_check.py [N]     (N defaults to the number of devices)"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
EMU = os.environ.get("S360_TEST_EMULATED_LIB") == "1"  # developer dry walk of steps 1-3 on the CPU emulation (EMU_DEVICES=N)
if EMU:
    from surround360_amd import _capi as _c
    _c.LIB_PATH = os.path.join(ROOT, "tools", "libs360_emu.so")


def step1():
    from surround360_amd import _capi
    L = _capi.lib()
    print("devices:", L.s360_device_count())
    p = L.s360_comm_library_path()
    print("librccl:", p.decode() if p else "NOT FOUND: " + L.s360_last_error(None).decode())
    return 0 if p else 1


def step2(n):
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    import rigutil
    from surround360_amd import _capi, render as R
    rig_path = rigutil.scaled_rig_json(os.path.join(ROOT, "tests", "golden", "rig_17cam.json"), os.path.join(tempfile.mkdtemp(), "rig.json"), 256 / 2048.0)
    rig = R.RigDescription(rig_path)
    flags = dict(eqr_width=504, eqr_height=252, enable_top=1, enable_bottom=1)
    ctxs = [R.Context(rig, R.make_params(**flags), device=d) for d in range(n)]
    frame = rigutil.frame_inputs(rig_path, 256, yaw_deg=0.0)
    for c in ctxs:
        c.upload_frame(*frame)
        c.render()  # (strip buffers exist)
        c.synchronize()
    arr = (C.c_void_p * n)(*[c.h for c in ctxs])
    rc = _capi.lib().s360_comm_init_all(arr, n)
    if rc < 0:
        print("s360_comm_init_all:", _capi.lib().s360_last_error(None).decode())
        return 1
    with ThreadPoolExecutor(n) as ex:  # one host thread per rank, as for every grouped call of single-process RCCL
        list(ex.map(lambda c: (c.comm_loopback(3, 9), c.synchronize()), ctxs))
    for c in ctxs:
        c.comm_destroy()
        c.close()
    print("comm_init_all + loopback on %d ranks: ok" % n)
    return 0


def step3(n):
    import refprog
    import rigutil
    work = tempfile.mkdtemp(prefix="s360_multi_")
    rig = rigutil.scaled_rig_json(os.path.join(ROOT, "tests", "golden", "rig_17cam.json"), os.path.join(work, "rig_small.json"), refprog.CAM / 2048.0)
    exe = os.path.join(ROOT, "tools", "emu", "TestRenderStereoPanorama") if EMU else refprog.HOST_EXE
    out = refprog.run_case(exe, work, rig, "two_frames", more_args=["--num_gpus", str(n)], env={"S360_RCCL_VERBOSE": "1"}, timeout=300)
    got, want = refprog.digests(out, "two_frames"), json.load(open(refprog.GOLDEN))["two_frames"]
    bad = sorted(k for k in want if got.get(k) != want[k])
    print("sharded two-frame case on %d GPUs: %d files, %d differ %s" % (n, len(want), len(bad), bad[:6]))
    return 1 if bad else 0


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--step":
        k, n = int(sys.argv[2]), int(sys.argv[3])
        sys.exit({1: step1, 2: lambda: step2(n), 3: lambda: step3(n)}[k]())
    from surround360_amd import _capi
    n = int(sys.argv[1]) if len(sys.argv) > 1 else _capi.lib().s360_device_count()
    if n < 2:
        print("needs at least 2 GPUs (%d visible)" % n)
        sys.exit(2)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    steps = [("1 librccl + devices", [sys.executable, __file__, "--step", "1", str(n)], 120),
             ("2 comm_init_all + loopback", [sys.executable, __file__, "--step", "2", str(n)], 300),
             ("3 sharded frames vs the reference program's digests", [sys.executable, __file__, "--step", "3", str(n)], 600),
             ("4 bench.py --gpus %d" % n, [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "4", "--warmup", "2"], 1200)]
    if EMU:
        steps = steps[:3]
    for name, cmd, tmo in steps:
        print("==== step %s" % name, flush=True)
        try:
            r = subprocess.run(cmd, env=env, cwd=ROOT, timeout=tmo, capture_output=True, text=True)
            sys.stdout.write(r.stdout[-3000:])
            if r.returncode != 0:
                sys.stdout.write(r.stderr[-3000:])
                print("FAILED at step %s (rc %d)" % (name, r.returncode))
                sys.exit(1)
        except subprocess.TimeoutExpired:
            print("FAILED at step %s: no result in %d s" % (name, tmo))
            sys.exit(1)
    print("all steps passed on %d GPUs" % n)


if __name__ == "__main__":
    main()
