#!/usr/bin/env python
"""bench.py — stereo-equirect frames/sec at 8K (17-camera rig) on N MI355X.

One "step" = one pass of the hot path over one synthetic 17-camera frame: spherical reprojection, 28 side
PixFlow flows, novel-view strips, panorama assembly, 4 pole flows + warps, composite, final resize to
8192x8192 — everything renderStereoPanorama does between decoded inputs and the stacked equirect
(TestRenderStereoPanorama.cpp:716-972). Inputs are uploaded to HBM before the timed region.

Timed region (`value`): every rank renders independent frames of BASELINE.json configs[2]: `--inflight` contexts (2; one
HIP stream and one submitting host thread each) x `--slots` frame slots (22: the frames of a context go through ONE launch
sequence, their flows in the same batched kernels), every slot a DIFFERENT frame of the synthetic stream — 44 frames
resident in 251 GB of HBM (a single frame is latency-bound by PixFlow's raster-order sweeps and leaves most of the chip
idle; the pole flows' serial chain costs a batch the same whatever it holds: DESIGN.md sections 5 - 7). A step = one batch. Per-GPU work is fixed => "scaling": "weak"; `value` is the aggregate over all ranks. After the timed region
every context's equirect is downloaded and byte-compared with the render of the same inputs by one context alone
(`checked`).

Then, one context alone on the GPU (nothing else in flight, so kernel durations are isolated):
  * `roofline`: the dominant kernel of the timed region (the throughput-mode sweep) — algorithmic bytes per launch
    (48 B per pixel-level-sweep) / its average ISOLATED launch duration from HIP events on the library's
    stream / 8 TB/s; `aggregate_frac` is the same bytes over the wall time of the timed region (launches of up to
    `inflight` frames overlapping);
  * `single_frame`: configs[2] as a latency, per-kernel-family milliseconds, the warp/blend and flow-stencil kernels
    against the HBM roofline, and the same frame with the reference presets' sharpening 0.25;
  * `sharded_frame` (N > 1): configs[3] — pairs and pole units sharded over the ranks, the two native RCCL exchanges,
    composite on rank 0 — ms per frame, `rccl_ranks` as the communicator reports it (ncclCommCount), and each exchange's
    bytes / ms / GB/s against the 153 GB/s xGMI link (HIP events around ncclGroupStart..ncclGroupEnd); run by a child
    process per rank on a rendezvous of its own, so that a fault in that path cannot cost this line;
  * `config2_flow_pair`: BASELINE configs[1], one 2048x2048 pair, both directions, GPU vs the CPU oracle;
  * `video_stream`: configs[4] on one GPU — 190 frames of a rotating world with a moving disc, every frame
    regularised toward its predecessor's device-resident flows, inputs fed from page-locked host buffers on the upload
    stream while the previous frame renders, the finished frame fetched while the next one renders; with and without
    frame pipelining; steady state over frames 10-189, and what the reference's per-frame state files would add
    (`spill_ms_per_frame`); on N GPUs: one such stream per rank at the same time (a stream cannot use more than one GPU);
  * `end_to_end_files`: SURVEY 8d's "end-to-end incl. raw I/O" figure — the drop-in host program
    (host/TestRenderStereoPanorama --num_frames) rendering the first 14 frames of that stream from PNG files on disk to
    equirect PNG files on disk, its last frame compared with the same chain rendered through the C ABI in this process;
  * `cpu_baseline`: kind "reference" — the reference's own TestRenderStereoPanorama program (oracle/_ref) rendering the
    SAME 8K frame once as a process on the host cores (N=1 only), its equirect compared with the GPU's; where oracle/_ref
    is absent the CPU oracle port with the reference's thread shape (kind "port").

Prints ONE JSON line on rank 0. `python bench.py --gpus N` without a launcher starts itself under torch.distributed.run
(one rank per GPU, 127.0.0.1); under the driver's own torch.distributed.run it runs as the rank it is given.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

RIG = os.path.join(ROOT, "tests", "golden", "rig_17cam.json")
FLAGS_8K = dict(eqr_width=8400, eqr_height=4096, enable_top=1, enable_bottom=1, final_eqr_width=8192,
                final_eqr_height=8192, sharpening=0.25)  # the reference's "8k" preset, batch_process_video.py:194-199 (SHARPENNING = 0.25)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
SWEEP_BYTES_PER_PX = 48  # a sweep reads 40 B + writes 8 B per pixel-level


def host_cpus():
    """CPUs this process may use: os.cpu_count() cut down to the control group's quota (the GPU boxes are containers with 16 CPUs
    of a 256-thread host: /sys/fs/cgroup/cpu.max = 1600000 100000)."""
    n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max" and int(p) > 0:
            n = max(1, min(n, int(int(q) / int(p))))
    except Exception:  # noqa: BLE001
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = max(1, min(n, q // p))
        except Exception:  # noqa: BLE001
            pass
    return n


def pyramid_levels(w, h):
    """PixFlow.h:477-491 on the x0.5 downscaled input."""
    cw, ch = int(w * 0.5), int(h * 0.5)
    out = []
    while True:
        out.append((cw, ch))
        nw, nh = int(np.float32(cw) * np.float32(0.9) + np.float32(0.5)), int(
            np.float32(ch) * np.float32(0.9) + np.float32(0.5))
        if nh <= 24 or nw <= 24:
            break
        cw, ch = nw, nh
    return out


def flow_px_levels(geom, eqr_w):
    side = sum(w * h for w, h in pyramid_levels(geom.overlap_image_width, geom.cam_image_height))
    pole = sum(w * h for w, h in pyramid_levels(int(np.float32(eqr_w) * np.float32(1.2)), geom.top_rows))
    return side, pole


def sweep_algorithmic_bytes(geom, n_side_flows, n_pole_flows, eqr_w):
    side, pole = flow_px_levels(geom, eqr_w)
    return 2 * SWEEP_BYTES_PER_PX * (n_side_flows * side + n_pole_flows * pole)  # forward + backward sweep per level


def cpu_baseline_8k(side, top, bottom, rig_path=RIG, flags=None):
    """The oracle (kind "port") on the bench's own 8K frame, once, with the reference's thread shape."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O  # test infrastructure: used here only as the timed CPU baseline
    cams, _ = O.load_rig(rig_path)
    f = O.Frame(cams, O.make_params(**(FLAGS_8K if flags is None else flags)))
    t0 = time.time()
    out, _ = f.render(side, top, bottom, threaded=True)
    sec = time.time() - t0
    st = f.stage_seconds()
    return out, {"value": 1.0 / sec, "unit": "frames/s", "cores": 18, "kind": "port",
                 "host_cores_available": host_cpus(), "host_hardware_threads": os.cpu_count() or 1, "seconds_per_frame": round(sec, 2),
                 "stage_seconds": {k: round(v, 2) for k, v in st.items()},
                 "sample": "CPU restatement of Surround360 (OpenCV-free oracle, -O3, no FMA): ONE full 8K frame of the bench "
                           "workload (eqr 8400x4096 -> 8192x8192, top+bottom, pixflow_low), reference thread shape = 14 "
                           "camera/pair threads for projection, flow and novel views, then 4 pole-unit threads, "
                           "single-threaded inside every operator (TestRenderStereoPanorama.cpp:153-175, 320-372, 811-860); "
                           "cores = peak threads in use"}


REF_PROGRAM = os.path.join(ROOT, "oracle", "_ref", "TestRenderStereoPanorama")


ISP_JSON_NEUTRAL = json.dumps({"CameraIsp": {"bayerPattern": "GBRG"}})  # every other key at CameraIsp's defaults (CameraIsp.h:440-462)


def write_capture_container(path, isp_dir, cams_by_id, frames_by_id, bits=12):
    """The synthetic stream as a capture's .bin container (BinaryFootageFile.cpp: a 4096-byte metadata page, then the packed frames
    interleaved by camera) + one ISP configuration per camera serial: every B,G,R frame mosaiced (GBRG), 8 -> 12 bits, packed the way
    the sensor packs two samples into three bytes. Camera k of the container carries the k-th smallest serial, which makes it the
    rig's "cam<k>" (Unpacker.cpp:203-219). frames_by_id[f][id] = H x W x 3 uint8. Returns the packed bytes [f][k] as written."""
    from concurrent.futures import ThreadPoolExecutor
    n = len(cams_by_id)
    assert sorted(cams_by_id) == sorted("cam%d" % k for k in range(n))
    serials = [50000 + 11 * k for k in range(n)]
    os.makedirs(isp_dir, exist_ok=True)
    for sn in serials:
        with open(os.path.join(isp_dir, "%d.json" % sn), "w") as f:
            f.write(ISP_JSON_NEUTRAL)
    h, w = frames_by_id[0]["cam0"].shape[:2]

    def pack(job):
        img, serial = job
        raw = np.empty((h, w), np.uint8)  # GBRG: (0,0) G, (0,1) B, (1,0) R, (1,1) G; images are B,G,R
        raw[0::2, 0::2] = img[0::2, 0::2, 1]
        raw[0::2, 1::2] = img[0::2, 1::2, 0]
        raw[1::2, 0::2] = img[1::2, 0::2, 2]
        raw[1::2, 1::2] = img[1::2, 1::2, 1]
        out = np.zeros((h, w // 2, 3), np.uint8)  # 12-bit samples a = v8 << 4, b likewise: bytes a >> 4, (a & 15) | (b & 15) << 4, b >> 4
        out[..., 0] = raw[:, 0::2]
        out[..., 2] = raw[:, 1::2]
        fr = out.reshape(-1)
        fr[4:8] = np.array([serial], np.uint32).view(np.uint8)  # the camera stamps its serial number over the first samples
        return fr
    page = np.zeros(4096, np.uint8)
    page[:32] = np.array([0xfaceb00c, 1234, 0, 1, w, h, bits, n], np.uint32).view(np.uint8)
    written = []
    with open(path, "wb") as f, ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as ex:
        f.write(page.tobytes())
        for fr in frames_by_id:
            row = list(ex.map(pack, [(fr["cam%d" % k], serials[k]) for k in range(n)]))
            for b in row:
                f.write(b.tobytes())
            written.append(row)
    return written


def batched_streams_run(program, common, src_args, out, tag, first, nf, ns, device, timeout):
    """One --num_streams run of the host program on one GPU (the streams are the frame slots of one context); the record of what
    it printed (--v 1) and measured."""
    import re
    import subprocess
    cb = [program] + common + src_args + ["--frame_number", first, "--num_frames", str(nf), "--num_streams", str(ns), "--stream_gpus", "1",
                                         "--output_data_dir", out, "--prev_frame_data_dir", "NONE",
                                         "--output_equirect_path", os.path.join(out, tag + "_%s.png"), "--device", str(device),
                                         "--write_state=false", "--v", "1"]
    t1 = time.perf_counter()
    rb = subprocess.run(cb, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=timeout)
    wb = time.perf_counter() - t1
    if rb.returncode != 0:
        raise RuntimeError("rc %d: %s" % (rb.returncode, rb.stderr[-300:]))
    hb = re.search(r"host thread per step:\s+decode \+ upload ([0-9.]+)\s+wait for the encoders ([0-9.]+)\s+wait for the GPU \+ fetch ([0-9.]+)\s+enqueue ([0-9.]+)", rb.stderr)
    sm = re.search(r"steady state:\s+(\d+) frames of steps 1\.\.(\d+) in ([0-9.]+)\s+\(([0-9.]+) frames per second", rb.stderr)
    rec = {"streams": ns, "frames": nf, "process_wall_s": round(wb, 2), "frames_per_s_process": nf / wb,
           "host_thread_s_per_step": {"decode_and_upload": float(hb.group(1)), "wait_for_encoders": float(hb.group(2)),
                                      "wait_for_gpu_and_fetch": float(hb.group(3)), "enqueue_next_step": float(hb.group(4))} if hb else None}
    if sm:
        rec.update({"frames_per_s_steady": float(sm.group(4)), "steady_frames": int(sm.group(1)), "steady_seconds": float(sm.group(3)),
                    "steady_note": "the program's own clock: from the arrival of step 0's frames on the host to the last file written, "
                                   "the frames of steps 1..%s" % sm.group(2)})
    return rec


def make_bins_leg(get_ctx, rig_path, stream_frame, dry, local_rank):
    """The `batched_streams_from_bins` leg of end_to_end_files as a closure host_program_stream calls with its scratch directory."""

    def bins_leg(program, common, work, device, timeout):
        ctx = get_ctx()
        from PIL import Image
        from surround360_amd import isp as I
        ns, per = (2, 2) if dry else (8, 4)
        nf = ns * per
        cams = json.load(open(rig_path))["cameras"]
        side_ids = [c["id"] for c in cams if "side" in c.get("group", "")]
        other = [c for c in cams if "side" not in c.get("group", "")]
        top_id = max(other, key=lambda c: c["forward"][2])["id"]
        bot_id = min(other, key=lambda c: c["forward"][2])["id"]
        ids = side_ids + [top_id, bot_id]
        by_id = []
        for k in range(nf):
            side, top, bottom = stream_frame(k)
            by_id.append(dict(zip(ids, list(side) + [top, bottom])))
        binp, ispd, outb = os.path.join(work, "0.bin"), os.path.join(work, "isp"), os.path.join(work, "out_bins")
        os.makedirs(outb)
        t0 = time.perf_counter()
        written = write_capture_container(binp, ispd, ids, by_id)
        t_write = time.perf_counter() - t0
        rb = batched_streams_run(program, common, ["--bin_list", binp, "--isp_dir", ispd], outb, "bins", "000000", nf, ns, device, timeout)
        # check: the LAST frame of every stream against the same packed frames sent through the C ABI in this process
        # (s360_frame_upload_packed: the same ISP arithmetic, CameraIspPipe at 16 bits as Unpacker runs it), one stream
        # after the other in one context, latency sweep kernel, pixels fetched with s360_frame_download_equirect
        isp = I.CameraIsp(I.config_from_json(ISP_JSON_NEUTRAL, 16, pipe=I.PIPE), device=local_rank)
        hh, ww = by_id[0][ids[0]].shape[:2]
        ctx.set_sweep_mode("latency")
        bad = []
        Image.MAX_IMAGE_PIXELS = None
        for st_ in range(ns):
            for j in range(per):
                f = st_ * per + j
                for k, cid in enumerate(side_ids):
                    ctx.upload_packed(isp, k, written[f][int(cid[3:])], 12, ww, hh)
                ctx.upload_packed(isp, -1, written[f][int(top_id[3:])], 12, ww, hh)
                ctx.upload_packed(isp, -2, written[f][int(bot_id[3:])], 12, ww, hh)
                ctx.render(j > 0)
            want = ctx.download_equirect()
            got = np.asarray(Image.open(os.path.join(outb, "bins_%06d.png" % (st_ * per + per - 1))))[:, :, ::-1]
            if got.shape != want.shape or not np.array_equal(got, want):
                bad.append(st_)
        isp.close()
        rb.update({"checked": not bad, "mismatching_streams": bad, "container_bytes": os.path.getsize(binp),
                   "container_write_s": round(t_write, 2),
                   "output_png_bytes_per_frame": sum(os.path.getsize(os.path.join(outb, "bins_%06d.png" % k)) for k in range(nf)) // nf,
                   "check": "the last frame's file of every stream decoded by PIL (libpng) against s360_frame_download_equirect of "
                            "the same packed frames rendered through the C ABI in this process",
                   "note": "%d frames as %d streams of %d, one process, one context with %d frame slots: inputs straight from the "
                           "capture's .bin container (12-bit packed Bayer, mmap -> s360_frame_upload_packed -> ISP on the device), "
                           "outputs as PNG files filtered and deflated on the device (s360_frame_download_png): the host inflates "
                           "and deflates nothing" % (nf, ns, per, ns)})
        return rb
    return bins_leg


def host_program_stream(frames, rig_path, flags, program, device=0, timeout=420, scratch=None, bins_leg=None):
    """"state both the device-path fps and the end-to-end fps incl. raw I/O". The drop-in host program
    (host/TestRenderStereoPanorama, the reference's binary name / flags / file layout) renders `frames` consecutive
    frames as ONE stream from PNG files on disk to equirect PNG files on disk: 17 PNG decodes per frame, upload, render
    with device-resident temporal state, download, 8192x8192 PNG encode, everything overlapped as --num_frames does it.
    Returns the record (wall time of the whole process incl. HIP start-up and context creation, and the program's own
    per-frame figure from its --v 1 runtime breakdown)."""
    import re
    import shutil
    import subprocess
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    from PIL import Image
    if not os.path.exists(program):
        return {"error": "%s not built" % os.path.relpath(program, ROOT)}
    cams = json.load(open(rig_path))["cameras"]
    side_ids = [c["id"] for c in cams if "side" in c.get("group", "")]
    other = [c for c in cams if "side" not in c.get("group", "")]
    top_id = max(other, key=lambda c: c["forward"][2])["id"]
    bot_id = min(other, key=lambda c: c["forward"][2])["id"]
    work = tempfile.mkdtemp(prefix="s360_e2e_", dir=scratch)  # scratch: where the files live (None: the default temporary directory)
    try:
        imgs, out = os.path.join(work, "rgb"), os.path.join(work, "out")
        for cid in side_ids + [top_id, bot_id]:
            os.makedirs(os.path.join(imgs, cid))
        jobs = []
        for k, (side, top, bottom) in enumerate(frames):
            for cid, img in list(zip(side_ids, side)) + [(top_id, top), (bot_id, bottom)]:
                jobs.append((np.asarray(img), os.path.join(imgs, cid, "%06d.png" % k)))
        t0 = time.perf_counter()
        with ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as ex:  # (PIL's zlib releases the GIL)
            list(ex.map(lambda j: Image.fromarray(np.ascontiguousarray(j[0][:, :, ::-1])).save(j[1], compress_level=1), jobs))
        t_write = time.perf_counter() - t0
        in_bytes = sum(os.path.getsize(j[1]) for j in jobs)
        n = len(frames)
        os.makedirs(os.path.join(out, "debug", "%06d" % (n - 1), "flow_images"))
        os.makedirs(os.path.join(out, "flow", "%06d" % (n - 1)))
        cmd = [program, "--rig_json_file", rig_path, "--imgs_dir", imgs, "--frame_number", "000000", "--num_frames", str(n),
               "--output_data_dir", out, "--prev_frame_data_dir", "NONE", "--output_equirect_path", os.path.join(out, "eqr_%s.png"),
               "--sharpening", repr(float(flags.get("sharpening", 0.0))), "--device", str(device), "--write_state=false", "--v", "1"]
        for k in ("eqr_width", "eqr_height", "final_eqr_width", "final_eqr_height"):
            cmd += ["--" + k, str(flags[k])]
        cmd += [f for f in ("--enable_top", "--enable_bottom") if flags.get(f[2:])]
        t0 = time.perf_counter()
        r = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=timeout)
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            return {"error": "rc %d: %s" % (r.returncode, r.stderr[-300:])}
        outs = [os.path.join(out, "eqr_%06d.png" % k) for k in range(n)]
        missing = [os.path.basename(o) for o in outs if not os.path.exists(o)]
        if missing:
            return {"error": "missing outputs: %s" % missing}
        m = re.search(r"stream of\s+(\d+) frames:\s+([0-9.]+)\s+\(([0-9.]+) per frame", r.stderr)
        Image.MAX_IMAGE_PIXELS = None
        last = np.ascontiguousarray(np.asarray(Image.open(outs[-1]))[:, :, ::-1])
        rec = {"program": os.path.relpath(program, ROOT), "frames": n, "process_wall_s": round(wall, 2),
               "frames_per_s_process": n / wall,
               "input_png_bytes_per_frame": in_bytes // n, "output_png_bytes_per_frame": sum(os.path.getsize(o) for o in outs) // n,
               "dataset_write_s": round(t_write, 2),
               "note": "END TO END incl. file I/O (SURVEY 8d): the drop-in host program reading 17 PNG files per frame from disk "
                       "and writing one stereo equirect PNG per frame to disk, the frames as one stream (--num_frames: temporal "
                       "regularisation, device-resident state, decode / upload / render / download / encode overlapped; PNG "
                       "codec of host/png_io.hpp, inputs written by PIL at compression level 1). process_wall_s includes "
                       "process start, HIP initialisation and context creation; ms_per_frame_stream is the program's own "
                       "runtime breakdown (first render to last file, TRSP:964-971)"}
        if m:
            rec["ms_per_frame_stream"] = 1e3 * float(m.group(3))
            rec["frames_per_s_stream"] = 1.0 / max(float(m.group(3)), 1e-9)
        hm = re.search(r"host thread per frame:\s+decode ([0-9.]+)\s+upload\+enqueue ([0-9.]+)\s+wait\+fetch ([0-9.]+)\s+wait for the encoder ([0-9.]+)\s+other (-?[0-9.]+)\s+of ([0-9.]+)", r.stderr)
        if hm:  # where the program's host thread spends a frame (its --v 1 breakdown, averages over all frames)
            rec["host_thread_ms_per_frame"] = {"png_decode": 1e3 * float(hm.group(1)), "upload_and_enqueue": 1e3 * float(hm.group(2)),
                                               "wait_for_gpu_and_fetch": 1e3 * float(hm.group(3)),
                                               "wait_for_png_encoder": 1e3 * float(hm.group(4)), "other": 1e3 * float(hm.group(5)),
                                               "total": 1e3 * float(hm.group(6)), "note": "frames 3.. of the stream"}
        if n >= 6:  # steady state: from the moment frame 2's file is complete to the last file's (the first frames pay
            # for the spherical maps, buffer growth and kernel loading)
            t = [os.stat(o).st_mtime_ns * 1e-9 for o in outs]
            rec["ms_per_frame_steady"] = 1e3 * (t[-1] - t[2]) / (n - 3)
            rec["frames_per_s_steady"] = (n - 3) / max(t[-1] - t[2], 1e-9)
            rec["steady_note"] = "frames 3..%d: time between the completion of eqr_000002.png and of the last file" % (n - 1)
        # ---- ONE frame per process: how the caller the drop-in exists for runs it (batch_process_video.py:29-62 starts the
        # program once per frame, every frame with --prev_frame_data_dir and the state files written for the next one) ----
        try:
            def one(frame, prev):
                os.makedirs(os.path.join(out, "debug", frame, "flow_images"), exist_ok=True)
                os.makedirs(os.path.join(out, "flow", frame), exist_ok=True)
                c1 = [program, "--rig_json_file", rig_path, "--imgs_dir", imgs, "--frame_number", frame, "--output_data_dir", out,
                      "--prev_frame_data_dir", prev, "--output_equirect_path", os.path.join(out, "single_%s.png" % frame),
                      "--sharpening", repr(float(flags.get("sharpening", 0.0))), "--device", str(device), "--v", "1"]
                for k in ("eqr_width", "eqr_height", "final_eqr_width", "final_eqr_height"):
                    c1 += ["--" + k, str(flags[k])]
                c1 += [f for f in ("--enable_top", "--enable_bottom") if flags.get(f[2:])]
                t1 = time.perf_counter()
                r1 = subprocess.run(c1, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=timeout)
                w1 = time.perf_counter() - t1
                if r1.returncode != 0:
                    raise RuntimeError("rc %d: %s" % (r1.returncode, r1.stderr[-300:]))
                lines = [ln.strip() for ln in r1.stderr.splitlines() if re.match(r"^(load|previous|GPU render|state files|equirect PNG|TOTAL)", ln.strip())]
                return w1, lines
            w0, b0 = one("000000", "NONE")
            w1, b1 = one("000001", "000000")
            st = os.path.join(out, "flow", "000001")
            rec["single_invocation"] = {
                "wall_s": round(w1, 3), "first_frame_wall_s": round(w0, 3), "breakdown": b1,
                "state_bytes_read": sum(os.path.getsize(os.path.join(out, "flow", "000000", f)) for f in os.listdir(os.path.join(out, "flow", "000000"))) +
                sum(os.path.getsize(os.path.join(out, "debug", "000000", "flow_images", f)) for f in os.listdir(os.path.join(out, "debug", "000000", "flow_images"))),
                "state_files_written": len(os.listdir(st)) + len(os.listdir(os.path.join(out, "debug", "000001", "flow_images"))),
                "note": "one process per frame, process start to exit: frame 1 with --prev_frame_data_dir (28 + 4 flow files and 36 state "
                        "images of frame 0 read, frame 1's written: TRSP:201-255, 413-452), 17 PNG inputs, one 8192x8192 equirect PNG; "
                        "the stream figures above are what a caller gets that keeps the process (--num_frames)"}
            rec["single_invocation_s"] = rec["single_invocation"]["wall_s"]
            same = np.asarray(Image.open(os.path.join(out, "single_000001.png")))[:, :, ::-1]
            if n >= 2:
                chain = np.asarray(Image.open(outs[1]))[:, :, ::-1]
                rec["single_invocation"]["equals_stream_frame"] = bool(np.array_equal(same, chain))
        except Exception as e:  # noqa: BLE001
            rec["single_invocation"] = {"error": repr(e)}
        # ---- the same files as FOUR streams in one process on this GPU (--num_streams 4: the streams are the frame slots of one
        # context, frame k of all four in one launch sequence, each with its own temporal state); the equirects leave as PNGs the
        # DEVICE encoded (--device_png, the default: png.hip) ----
        common = ["--rig_json_file", rig_path, "--sharpening", repr(float(flags.get("sharpening", 0.0)))]
        for k in ("eqr_width", "eqr_height", "final_eqr_width", "final_eqr_height"):
            common += ["--" + k, str(flags[k])]
        common += [f for f in ("--enable_top", "--enable_bottom") if flags.get(f[2:])]
        try:
            ns = 4
            nf = (n // ns) * ns
            if nf >= 2 * ns:
                rb = batched_streams_run(program, common, ["--imgs_dir", imgs], out, "batched", "000000", nf, ns, device, timeout)
                first = np.asarray(Image.open(os.path.join(out, "batched_000000.png")))[:, :, ::-1]
                ref0 = np.asarray(Image.open(outs[0]))[:, :, ::-1]
                rb["first_frame_equals_stream"] = bool(np.array_equal(first, ref0))
                rb["output_png_bytes_per_frame"] = sum(os.path.getsize(os.path.join(out, "batched_%06d.png" % k)) for k in range(nf)) // nf
                rb["note"] = ("%d frames as %d streams of %d (segments of the frame range), one process, one context with %d frame slots; "
                              "17 PNG files per frame in (inflated by host threads), one PNG per frame out, filtered and deflated on the "
                              "device" % (nf, ns, nf // ns, ns))
                rec["batched_streams"] = rb
        except Exception as e:  # noqa: BLE001
            rec["batched_streams"] = {"error": repr(e)}
        # ---- the same streams fed from the capture's .bin containers (--bin_list: raw 12-bit Bayer frames, developed by the ISP
        # on the device — SURVEY 8f row 4) and written as device-encoded PNGs: no pixel is inflated or deflated on the host ----
        if bins_leg is not None:
            try:
                rec["batched_streams_from_bins"] = bins_leg(program, common, work, device, timeout)
            except Exception as e:  # noqa: BLE001
                import traceback
                rec["batched_streams_from_bins"] = {"error": repr(e), "trace": traceback.format_exc()[-400:]}
        return rec, last
    finally:
        shutil.rmtree(work, ignore_errors=True)


def cpu_baseline_reference(side, top, bottom, rig_path=RIG, flags=None, timeout=900):
    """kind "reference": the reference's OWN program — test/TestRenderStereoPanorama.cpp and its libraries compiled from the
    reference's sources over stand-ins for OpenCV / Eigen / folly / gflags / glog (oracle/_ref, built by build() where the
    reference exists; tests/test_cpu_refprogram.py) — rendering the bench's frame once as a process, the way
    batch_process_video.py runs it: PNG inputs from disk, its own threads, equirect + state files to disk. Returns
    (equirect BGR, record) or None when the program is not there or fails (the caller then times the oracle port)."""
    if not os.path.exists(REF_PROGRAM):
        return None
    import shutil
    import subprocess
    import tempfile
    from PIL import Image
    flags = dict(FLAGS_8K if flags is None else flags)
    cams = json.load(open(rig_path))["cameras"]
    side_ids = [c["id"] for c in cams if "side" in c.get("group", "")]
    other = [c for c in cams if "side" not in c.get("group", "")]
    top_id = max(other, key=lambda c: c["forward"][2])["id"]
    bot_id = min(other, key=lambda c: c["forward"][2])["id"]
    work = tempfile.mkdtemp(prefix="s360_refprog_")
    try:
        imgs, out = os.path.join(work, "rgb"), os.path.join(work, "out")
        for cid, img in list(zip(side_ids, side)) + [(top_id, top), (bot_id, bottom)]:
            os.makedirs(os.path.join(imgs, cid))
            Image.fromarray(np.ascontiguousarray(np.asarray(img)[:, :, ::-1])).save(os.path.join(imgs, cid, "000000.png"), compress_level=1)
        os.makedirs(os.path.join(out, "debug", "000000", "flow_images"))
        os.makedirs(os.path.join(out, "flow", "000000"))
        eqr = os.path.join(out, "eqr.png")
        cmd = [REF_PROGRAM, "--rig_json_file", rig_path, "--imgs_dir", imgs, "--frame_number", "000000", "--output_data_dir", out,
               "--prev_frame_data_dir", "NONE", "--output_equirect_path", eqr, "--sharpening", repr(float(flags.get("sharpening", 0.0)))]
        for k in ("eqr_width", "eqr_height", "final_eqr_width", "final_eqr_height"):
            cmd += ["--" + k, str(flags[k])]
        cmd += [f for f in ("--enable_top", "--enable_bottom") if flags.get(f[2:])]
        t0 = time.time()
        p = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        peak = 1
        while p.poll() is None:  # peak number of threads the program runs (its own std::thread fan-out)
            try:
                for ln in open("/proc/%d/status" % p.pid):
                    if ln.startswith("Threads:"):
                        peak = max(peak, int(ln.split()[1]))
            except OSError:
                pass
            if time.time() - t0 > timeout:
                p.kill()
                return None
            time.sleep(0.1)
        sec = time.time() - t0
        if p.returncode != 0:
            return None
        Image.MAX_IMAGE_PIXELS = None
        got = np.ascontiguousarray(np.asarray(Image.open(eqr))[:, :, ::-1])
        return got, {"value": 1.0 / sec, "unit": "frames/s", "cores": max(1, peak - 1), "kind": "reference",
                     "host_cores_available": host_cpus(), "host_hardware_threads": os.cpu_count() or 1, "seconds_per_frame": round(sec, 2),
                     "sample": "the reference's own TestRenderStereoPanorama program (its sources compiled over stand-ins for "
                               "OpenCV / Eigen / folly / gflags / glog: oracle/_ref, built with the optimisation flags of the reference's own CMakeLists.txt:34 (-O3 -mavx -funroll-loops), no FMA) rendering ONE full 8K frame of "
                               "the bench workload (eqr 8400x4096 -> 8192x8192, top+bottom, pixflow_low, sharpening as benched) as one process: 17 PNG "
                               "inputs decoded from disk, the program's own thread fan-out, equirect and per-frame state files "
                               "encoded to disk — what batch_process_video.py pays per frame; cores = peak threads it ran"}
    finally:
        shutil.rmtree(work, ignore_errors=True)


# compulsory bytes per 8K frame of the warp/blend kernel families (MB)
WARP_BLEND_MB = {"project_side": 176 + 180, "project_pole": 2 * 12.6 + 141, "novel_view": 840, "assemble_pano": 400,
                 "pole_warp": 1600, "flatten": 1650}
# compulsory bytes per pixel-level of the flow stencil families (SURVEY.md §8d pass list, reference dtypes)
FLOW_STENCIL_B_PER_PXLEVEL = {"flow_gradients": 24, "flow_blur15": 16, "flow_median": 32, "flow_diffusion": 48,
                              "flow_upscale": 16}


def streams_batched(R, rig, flags, device, frames, args, dry, g, slots=None, timed_steps=None, check=True, contexts=2, pipelined=False):
    """bench.py's `video_streams_batched` leg: 2 contexts x S frame slots, one stream per slot (stream s = the synthetic video
    entered at another frame, walked forwards and backwards over the distinct frames held), every step a render_batch(use_prev)
    per context on device-resident temporal state, the step's inputs sent in place from page-locked host memory on the upload
    stream while the previous step renders. Steady state = the steps behind 4 run-in steps. Then EVERY stream's last frame is
    byte-compared with the same stream rendered frame by frame in a context of its own (latency sweep kernel, s360_frame_render)."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    F = contexts
    run_in, timed = 4, max(2, timed_steps or args.stream_steps)
    n_steps = run_in + timed
    free_b, total_b = (0, 0) if dry else torch.cuda.mem_get_info(device)
    # page-locked ring of distinct frames (read-only: any number of slots may send the same buffers)
    n_ring = min(len(frames), 4 if dry else 24)
    ring = []
    for k in range(n_ring):
        side, top, bottom = frames[k]
        ps = [R.pinned_empty(a.shape) for a in side]
        for d, a in zip(ps, side):
            np.copyto(d, a)
        pt, pb = R.pinned_empty(top.shape), R.pinned_empty(bottom.shape)
        np.copyto(pt, top)
        np.copyto(pb, bottom)
        ring.append((ps, pt, pb))

    def frame_of(stream, k):  # stream s enters the video 3 s frames in (the period below is even and not a multiple of 3)
        m = (k + (3 if n_ring > 8 else 1) * stream) % max(2 * (n_ring - 1), 1)
        return ring[m if m < n_ring else 2 * (n_ring - 1) - m]

    S = slots or args.stream_slots
    if S <= 0:  # measured at 8K: 5.7 GB per slot + 0.75 GB for the second halves of the flow INPUTS' double buffers (the flows
        # themselves are updated in place) + 1.7 GB of previous-frame pyramids, 6 GB per context
        S = 2 if dry else max(1, int((0.90 * total_b / 1e9 / F - 6.5) / 7.6))
    rec = {}
    for attempt in range(4):
        ctxs = []
        try:
            ctxs = [R.Context(rig, R.make_params(**flags), device=device) for _ in range(F)]
            for c in ctxs:
                c.set_frame_slots(S)
                c.set_sweep_mode("throughput")
                if pipelined:  # step k's pole stage on the second stream beside step k+1's side stage
                    c.set_frame_pipelining(True)

            def step(ci, k):
                c = ctxs[ci]
                for j in range(S):
                    c.select_frame_slot(j)
                    c.upload_frame(*frame_of(ci * S + j, k))  # page-locked: sent in place, behind the previous step's projections
                c.render_batch(k > 0)

            def sync():
                for c in ctxs:
                    c.synchronize()
                if not dry:
                    torch.cuda.synchronize()
            with ThreadPoolExecutor(F) as pool:
                for k in range(run_in):
                    list(pool.map(lambda ci: step(ci, k), range(F)))
                sync()
                used = 0 if dry else (total_b - torch.cuda.mem_get_info(device)[0]) / 1e9
                t0 = time.perf_counter()
                for k in range(run_in, n_steps):
                    list(pool.map(lambda ci: step(ci, k), range(F)))
                sync()
                dt = time.perf_counter() - t0
            break
        except R.S360Error as e:  # did not fit: fewer slots
            for c in ctxs:
                try:
                    c.close()
                except Exception:  # noqa: BLE001
                    pass
            if not dry:
                torch.cuda.empty_cache()
            rec.setdefault("retries", []).append("%d slots: %s" % (S, str(e)[:120]))
            if S <= 1 or attempt == 3:
                raise
            S = max(1, S - 2)
    if not check:  # a row of `slots_table`: the figure and what it costs in HBM (the full leg checks every stream of its own run)
        for c in ctxs:
            c.close()
        if not dry:
            torch.cuda.empty_cache()
        return {"streams": F * S, "slots_per_context": S, "steps": timed, "frames_per_s": timed * F * S / dt, "hbm_used_GB": round(used, 1)}
    last = []
    for c in ctxs:
        for j in range(S):
            c.select_frame_slot(j)
            last.append(np.array(c.download_equirect()))
    # the A13 kernels' share: one more step with the per-family events (perturbs the launch stream: not part of the timing)
    prof = {}
    for c in ctxs:
        c.profile_enable(True)
    with ThreadPoolExecutor(F) as pool:
        list(pool.map(lambda ci: step(ci, n_steps), range(F)))
    for c in ctxs:
        c.synchronize()
        for k, v in c.profile_get().items():
            prof[k] = prof.get(k, 0.0) + v[0]
        c.profile_enable(False)
        c.close()
    if not dry:
        torch.cuda.empty_cache()
    tot = sum(prof.values()) or 1.0
    # ---- every stream alone: W contexts of one slot, one stream after the other in each, latency kernel ----
    W = 2 if dry else 8
    t1 = time.perf_counter()
    alone_ctx = [R.Context(rig, R.make_params(**flags), device=device) for _ in range(W)]

    def alone(w):
        c, res = alone_ctx[w], {}
        c.set_sweep_mode("latency")
        for s in range(w, F * S, W):
            for k in range(n_steps):
                c.upload_frame(*frame_of(s, k))
                c.render(k > 0)
            res[s] = np.array_equal(c.download_equirect(), last[s])
        return res
    ok = {}
    with ThreadPoolExecutor(W) as pool:
        for r in pool.map(alone, range(W)):
            ok.update(r)
    for c in alone_ctx:
        c.close()
    bad = sorted(s for s, v in ok.items() if not v)
    distinct = len({hash(a[::64, ::64].tobytes()) for a in last})
    rec.update({
        "mode": "%d streams = 2 contexts x %d frame slots, one stream per slot; every step one s360_frame_render_batch(use_prev=1) per "
                "context: frame k of every stream regularised toward its own device-resident frame k-1 (both halves of every slot's "
                "temporal double buffers resident); inputs sent in place from page-locked host memory on the upload stream while the "
                "previous step renders; sharpening %.2f; steady state = steps %d..%d" % (F * S, S, flags["sharpening"], run_in, n_steps - 1),
        "streams": F * S, "slots_per_context": S, "steps": timed, "frames": timed * F * S,
        "frames_per_s": timed * F * S / dt, "ms_per_step": 1e3 * dt / timed, "ms_per_frame": 1e3 * dt / (timed * F * S),
        "hbm_used_GB": round(used, 1), "distinct_frames_in_ring": n_ring,
        "temporal_kernels_share_in_flight": round(prof.get("flow_prev", 0.0) / tot, 4),
        "kernel_ms_per_frame_in_flight": {k: round(v / (F * S), 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1])},
        "checked": len(ok) == F * S and not bad, "checked_streams": len(ok), "mismatching_streams": bad,
        "distinct_last_frames": distinct,
        "check": "the last frame (frame %d) of EVERY stream byte-compared with the same stream rendered frame by frame in a context of "
                 "its own (one slot, latency sweep kernel, s360_frame_render(use_prev)); %d such contexts at a time" % (n_steps - 1, W),
        "check_seconds": round(time.perf_counter() - t1, 1)})
    return rec


def main():
    t_process = time.perf_counter()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="timed region + check + roofline only (profiling runs)")
    ap.add_argument("--sharded-child", action="store_true",
                    help="internal (N > 1): the configs[3] leg — one frame sharded over the ranks, the native RCCL exchanges — in "
                         "a process of its own per rank, so that nothing it does can cost the parent's bench line")
    ap.add_argument("--e2e-only", type=int, default=0, metavar="N",
                    help="only the end_to_end_files leg: N frames of the synthetic stream through the host program, PNG files "
                         "in and out (a short GPU call while tuning the host side; prints that leg's record, not a bench line)")
    ap.add_argument("--inflight", type=int, default=2,
                    help="contexts in flight per GPU (one HIP stream + one submitting host thread each)")
    ap.add_argument("--slots", type=int, default=22,
                    help="frame slots per context: S independent frames rendered by ONE launch sequence with their flows "
                         "in the same batched kernels (s360_frame_render_batch); a step is then one batch of S frames")
    ap.add_argument("--stream-slots", type=int, default=0,
                    help="video_streams_batched leg: streams (frame slots) per context; 0 = as many as fit with both halves of "
                         "every slot's temporal double buffers resident")
    ap.add_argument("--streams-only", action="store_true",
                    help="only the video_streams_batched leg (slots-against-frames/s probes: --stream-slots N); prints that leg's record")
    ap.add_argument("--stream-contexts", type=int, default=2, help="--streams-only: contexts of the leg")
    ap.add_argument("--stream-pipelined", action="store_true", help="--streams-only: the contexts' steps frame-pipelined (s360_set_frame_pipelining)")
    ap.add_argument("--stream-steps", type=int, default=8, help="video_streams_batched leg: timed steps (after 4 run-in steps)")
    ap.add_argument("--pipeline-batches", action="store_true",
                    help="the timed region's batches with frame pipelining (s360_set_frame_pipelining on a context of frame slots): "
                         "batch k's pole stage on a second stream beside batch k+1's side stage inside ONE context")
    ap.add_argument("--video-frames", type=int, default=190,
                    help="frames of the configs[4] stream leg (SURVEY 8d: 190, steady state over frames 10-189)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` started directly: become the launcher the contract names — one rank per GPU under
        # torch.distributed.run on 127.0.0.1 — with the same arguments; rank 0 prints the one JSON line on this process's stdout.
        import socket
        port = 29500
        for _ in range(8):  # a free port whose successor is free as well (the sharded-frame children rendezvous on port + 1)
            with socket.socket() as a, socket.socket() as b:
                a.bind(("127.0.0.1", 0))
                port = a.getsockname()[1]
                try:
                    b.bind(("127.0.0.1", port + 1))
                    break
                except OSError:
                    continue
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver shares device memory between processes through dmabuf only
        env.setdefault("OMP_NUM_THREADS", "1")
        sys.stdout.flush()
        os.execve(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:], env)

    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # hardware queues for the in-flight frames (read at HIP init)
    import torch
    from surround360_amd import parallel, render as R, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d under a launcher that made %d ranks" % (args.gpus, world))
    # Debugging aids for a box with fewer GPUs than ranks (NOT a bench configuration): S360_BENCH_BACKEND=gloo and
    # S360_BENCH_DEVICE=0 run the multi-process control flow with every rank on one device.
    backend = os.environ.get("S360_BENCH_BACKEND", "nccl")
    local_rank = int(os.environ.get("S360_BENCH_DEVICE", local_rank))
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    # Developer dry run, NOT a bench configuration: S360_TEST_EMULATED_LIB=1 walks this whole script on a machine without
    # a GPU — the library's sources compiled for the CPU (tools/libs360_emu.so), a rig scaled to 128x128 cameras, eqr
    # 252x126, a handful of steps — to check the script's control flow. Its JSON line says "dry_run".
    dry = os.environ.get("S360_TEST_EMULATED_LIB") == "1"
    rig_path, cam_size, world_h, pair_size = RIG, 2048, 8192, 2048
    flags = dict(FLAGS_8K)
    if dry:
        from surround360_amd import _capi
        _capi.LIB_PATH = os.path.join(ROOT, "tools", "libs360_emu.so")
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import rigutil
        cam_size, world_h, pair_size = 128, 256, 100
        rig_path = rigutil.scaled_rig_json(RIG, "/tmp/bench_dry_run_rig_%d.json" % rank, cam_size / 2048.0)
        flags.update(eqr_width=252, eqr_height=126, final_eqr_width=240, final_eqr_height=240)

        class _NoCuda:  # the handful of torch.cuda calls of this script
            set_device = synchronize = empty_cache = staticmethod(lambda *a, **k: None)
            mem_get_info = staticmethod(lambda *a, **k: (0, 0))
        torch.cuda = _NoCuda
    torch.cuda.set_device(local_rank)
    dev = torch.device("cpu") if dry else torch.device("cuda", local_rank)
    red_dev = dev if backend == "nccl" else torch.device("cpu")  # where the max-over-ranks timing tensors live

    rig = R.RigDescription(rig_path)
    P = rig.get_side_camera_count()
    F = max(1, args.inflight)
    S = max(1, args.slots)
    # ---- synthetic stream: one seeded equirect world (noise + near objects at 2 m / 5 m), rendered
    # through the 17 rig cameras on the GPU; frame k = world rotated by 0.2 deg * k, one disc moving 0.5 deg per frame.
    # (The world is the 16384x8192 texture SURVEY 8d names — 2.1 GB of texture + depth on the device while the frames are
    # rendered, freed before the contexts are made; rounds 1-3 used 8192x4096.)
    n_video = 0 if (args.no_extras or world > 1) else max(args.video_frames, 12)
    # distinct frames kept in host memory for the stream leg: all of them if the host has room (17 x 12.6 MB each),
    # otherwise a ring walked forwards and backwards (consecutive frames still differ by one step of motion)
    frame_bytes = 17 * cam_size * cam_size * 3
    n_distinct = n_video
    try:
        avail = [int(ln.split()[1]) * 1024 for ln in open("/proc/meminfo") if ln.startswith("MemAvailable:")][0]
        if 2.5 * n_video * frame_bytes + (60 << 30) > avail:
            n_distinct = min(n_video, 48)
    except Exception:  # noqa: BLE001
        n_distinct = min(n_video, 48)
    wtex = synth.World(world_h, seed=360, device=dev)  # the same stream on every rank (the sharded frame needs identical inputs)
    rr = synth.RigRenderer(rig_path, wtex, cam_size)
    if args.e2e_only > 0:
        fr = [rr.frame_numpy(yaw_deg=0.2 * k, disc_deg=10.0 + 0.5 * k) for k in range(args.e2e_only)]
        del rr, wtex
        torch.cuda.empty_cache()
        prog = os.path.join(ROOT, "tools", "emu", "TestRenderStereoPanorama") if dry else os.path.join(ROOT, "host", "TestRenderStereoPanorama")
        c1 = R.Context(rig, R.make_params(**flags), device=local_rank)
        n14 = min(14, len(fr))
        res = host_program_stream(fr[:n14], rig_path, flags, prog, device=local_rank,
                                  bins_leg=make_bins_leg(lambda: c1, rig_path, lambda k: fr[k % len(fr)], dry, local_rank))
        rec = res[0] if isinstance(res, tuple) else res
        if isinstance(res, tuple):  # the same chain through the C ABI in this process
            fr = fr[:n14]
            c1.set_frame_pipelining(True)
            for k, f in enumerate(fr):
                c1.upload_frame(*f)
                c1.render(k > 0)
            rec["last_frame_equals_in_process_stream"] = bool(np.array_equal(c1.download_equirect(), res[1]))
            c1.close()
        print(json.dumps({"end_to_end_files": rec}))
        return
    if args.sharded_child:
        if os.environ.get("S360_BENCH_CHILD_ABORT") == str(rank):  # developer check of the isolation: this rank's child dies
            os.abort()
        # ---- configs[3]: one frame at a time, the 14 pairs and the 4 pole units sharded over the ranks ----
        f0 = rr.frame_numpy(yaw_deg=0.0, disc_deg=10.0)  # the same frame on every rank (same seeds)
        del rr, wtex
        torch.cuda.empty_cache()
        c1 = R.Context(rig, R.make_params(**flags), device=local_rank)
        c1.set_sweep_mode("latency")
        c1.upload_frame(*f0)
        single0 = None
        if rank == 0:  # the unsharded frame, for the comparison
            c1.render(False)
            single0 = c1.download_equirect()
        bounds = parallel.partition_pairs(P, world)
        p0, p1 = bounds[rank], bounds[rank + 1]
        # the library's own RCCL communicator: rank 0 creates the id, torch.distributed only carries the 128 bytes
        ids = [R.Context.comm_get_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        c1.comm_init_rank(ids[0], rank, world)
        owner = parallel.pole_owners(world)  # pole unit u on rank u (2-3 ranks: the two poles on ranks 0 / 1)
        masks, needs = parallel.unit_masks(owner, world), parallel.strip_needs(owner, world)

        def sharded():
            c1.render_pairs(p0, p1, False)
            c1.exchange_strips(bounds, needs)  # grouped ncclSend/ncclRecv on the context stream (comm.cpp)
            if masks[rank] or rank == 0:
                c1.pole_units(masks[rank], False)
            c1.gather_pole_layers(owner, 0)    # second grouped exchange: the warped pole layers to the root
            if rank == 0:
                c1.composite(15)

        def csync():
            c1.synchronize()
            torch.cuda.synchronize()
            dist.barrier()
        sharded()
        csync()
        n_single = 3
        t1 = time.perf_counter()
        for _ in range(n_single):
            sharded()
        csync()
        dt1 = time.perf_counter() - t1
        t = torch.tensor([dt1], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt1 = float(t.item())
        ok = bool(np.array_equal(c1.download_equirect(), single0)) if rank == 0 else True
        # ---- the two exchanges on their own: every rank enters each one behind a barrier with an idle stream, HIP events on the
        # library's stream around ncclGroupStart .. ncclGroupEnd (s360_profile_get "exchange_strips" / "exchange_pole_layers");
        # the buffers hold the frame just rendered, so re-sending them changes nothing ----
        n_x = 5
        before = [c1.comm_stats(0), c1.comm_stats(1)]
        c1.profile_enable(True)
        for _ in range(n_x):
            c1.exchange_strips(bounds, needs)
            csync()
        for _ in range(n_x):
            c1.gather_pole_layers(owner, 0)
            csync()
        pr = c1.profile_get()
        c1.profile_enable(False)
        after = [c1.comm_stats(0), c1.comm_stats(1)]
        mine = []
        for i, fam in enumerate(("exchange_strips", "exchange_pole_layers")):
            calls = max(after[i]["calls"] - before[i]["calls"], 1)
            mine += [pr.get(fam, (0.0, 0))[0] / calls, (after[i]["bytes_sent"] - before[i]["bytes_sent"]) / calls,
                     (after[i]["bytes_received"] - before[i]["bytes_received"]) / calls]
        mine += [float(c1.comm_size()), float(c1.comm_rank())]
        tm = torch.tensor(mine, dtype=torch.float64, device=red_dev)
        allm = [torch.zeros_like(tm) for _ in range(world)]
        dist.all_gather(allm, tm)
        per = [[float(v) for v in t.tolist()] for t in allm]
        if rank == 0:
            XGMI_LINK_GBPS = 153.0  # MI355X: 7 links per GPU, ~153 GB/s each, point to point

            gq = c1.geometry
            per_strip = gq.cam_image_height * (flags["eqr_width"] // P) * 4  # one pair's strip of one eye (uchar4)
            # the busiest single link: the largest (sender -> receiver) transfer of the group — xGMI is point to point, one link per peer
            link = [max([bin(needs[r]).count("1") * (bounds[q + 1] - bounds[q]) * per_strip
                         for q in range(world) for r in range(world) if q != r] or [0]), 0]

            def exchange(i, what):
                ms = [p[3 * i] for p in per]
                sent = [p[3 * i + 1] for p in per]
                recv = [p[3 * i + 2] for p in per]
                worst = max(ms)
                link_bytes = link[0] if i == 0 else max(sent)  # pole layers: everything an owner sends goes to the root
                rate = (lambda b: b / worst / 1e6) if worst > 0 else (lambda b: None)  # (the CPU emulation's events read 0 ms)
                return {"what": what, "ms": worst, "ms_per_rank": [round(v, 4) for v in ms],
                        "bytes_moved": int(sum(sent)), "bytes_sent_per_rank": [int(v) for v in sent],
                        "bytes_received_per_rank": [int(v) for v in recv],
                        "GBps_aggregate": rate(sum(sent)), "GBps_busiest_receiver": rate(max(recv)),
                        "busiest_link_bytes": int(link_bytes), "GBps_busiest_link": rate(link_bytes),
                        "xgmi_link_peak_GBps": XGMI_LINK_GBPS,
                        "frac_of_link_peak": rate(link_bytes) / XGMI_LINK_GBPS if worst > 0 else None,
                        "timing": "HIP events on the library's stream around ncclGroupStart..ncclGroupEnd, %d calls, every rank behind a "
                                  "barrier with an idle stream; ms = the slowest rank" % n_x}
            sizes = sorted({int(p[6]) for p in per})
            print(json.dumps({"sharded_frame": {
                "mode": "configs[3]: one frame at a time, 14 pairs sharded over %d GPUs (%s), one native RCCL exchange (grouped "
                        "ncclSend/ncclRecv, s360_frame_exchange_strips) handing the strips to the ranks that assemble an eye, the "
                        "pole units on ranks %s, a second grouped exchange returning their warped layers to rank 0, which "
                        "composites; run in a process of its own per rank beside the bench's resident contexts" % (world, bounds, owner),
                "ms": 1e3 * dt1 / n_single, "frames_per_s": n_single / max(dt1, 1e-9),
                "rccl_ranks": sizes[0] if len(sizes) == 1 else sizes,
                "rccl_ranks_source": "ncclCommCount of every rank's communicator (s360_comm_size); user ranks %s" % sorted(int(p[7]) for p in per),
                "rccl_library": R.Context.comm_library_path(),
                "pair_blocks": bounds, "pole_unit_owner": owner,
                "exchange_strips": exchange(0, "strips of every pair block to the ranks that assemble an eye (SURVEY 8e, TRSP:320-385)"),
                "exchange_pole_layers": exchange(1, "warped pole layers of the units run off the root, to the root (TRSP:811-860)"),
                "checked": ok, "equals_single_gpu_frame": ok,
                "check": "the sharded frame's equirect byte-compared with the same inputs rendered by rank 0 alone"}}))
            sys.stdout.flush()
        dist.barrier()
        c1.close()
        dist.destroy_process_group()
        return
    frames = [rr.frame_numpy(yaw_deg=0.2 * k, disc_deg=10.0 + 0.5 * k) for k in range(24 if args.streams_only else max(F * S, n_distinct))]
    if args.streams_only:
        del rr, wtex
        torch.cuda.empty_cache()
        print(json.dumps({"video_streams_batched": streams_batched(R, rig, flags, local_rank, frames, args, dry, None,
                                                                   contexts=max(1, args.stream_contexts), pipelined=args.stream_pipelined)}))
        return

    def stream_frame(k):  # frame k of the stream: 0,1,..,n-1,n-2,..,1,0,1,.. over the distinct frames held
        if n_distinct >= n_video:
            return frames[k]
        period = 2 * (n_distinct - 1)
        m = k % period
        return frames[m if m < n_distinct else period - m]
    del rr, wtex
    torch.cuda.empty_cache()

    ctxs = [R.Context(rig, R.make_params(**flags), device=local_rank) for _ in range(F)]
    ctx = ctxs[0]
    for k, c in enumerate(ctxs):
        if S > 1:
            c.set_frame_slots(S)
        for j in range(S):
            if S > 1:
                c.select_frame_slot(j)
            c.upload_frame(*frames[k * S + j])  # inputs resident in HBM before the timed region
        if F * S > 1:
            c.set_sweep_mode("throughput")  # several frames in flight: the kernel with the fewest instructions per pixel
        if args.pipeline_batches:
            c.set_frame_pipelining(True)

    def sync(barrier=True):
        for c in ctxs:
            c.synchronize()
        torch.cuda.synchronize()
        if dist is not None and barrier:
            dist.barrier()

    # ---- timed region: every rank renders K whole frames, up to F in flight (independent frames: no collective) ----
    # One host thread per context: a frame is ~1000 kernel launches, and a single submitting thread caps the node
    # at ~27 frames/s whatever the GPU does (ctypes releases the GIL inside libs360, the HIP runtime locks per stream).
    from concurrent.futures import ThreadPoolExecutor
    pools = [ThreadPoolExecutor(max_workers=1) for _ in range(F)]
    counter = [0]
    futures = []
    enqueue_s = [0.0] * F  # host time spent inside s360_frame_render, per context

    def enqueue(k):
        t = time.perf_counter()
        if S > 1:
            ctxs[k].render_batch(False)  # S frames, one launch sequence
        else:
            ctxs[k].render(False)  # asynchronous enqueue on that context's stream
        enqueue_s[k] += time.perf_counter() - t

    def step():
        k = counter[0] % F
        futures.append(pools[k].submit(enqueue, k))
        counter[0] += 1

    def drain(barrier=True):
        for f in futures:
            f.result()
        del futures[:]
        sync(barrier)

    # Untimed set-up before the W warm-up steps: every context renders once (allocations, cached maps), then — only
    # if launches are being held up — frames are rendered until the host-side enqueue time per frame is back to a
    # small multiple of the uncontended one. Observed on shared boxes: for the first seconds after another GPU
    # process has exited, enqueueing a frame takes ~8x longer and the GPU starves. Bounded at 60 s; not part of the
    # warm-up or of the timed steps.
    settle = {"batches": 0, "seconds": 0.0}
    if F > 1:
        for _ in range(F):
            step()
        drain(barrier=False)
        t = time.perf_counter()
        enqueue(0)
        base_enqueue = time.perf_counter() - t  # one thread, idle GPU queues
        sync(barrier=False)
        t_settle = time.perf_counter()
        while time.perf_counter() - t_settle < 60.0:
            for k in range(F):
                enqueue_s[k] = 0.0
            for _ in range(F):
                step()
            drain(barrier=False)
            settle["batches"] += 1
            if sum(enqueue_s) / F < 6.0 * base_enqueue:
                break
        settle["seconds"] = time.perf_counter() - t_settle
        sync()  # all ranks settled
    for _ in range(args.warmup):
        step()
    drain()
    for k in range(F):
        enqueue_s[k] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):  # the timed region carries no event records: nothing but the frames' own launches
        step()
    drain()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    enqueue_ms_per_frame = 1e3 * sum(enqueue_s) / max(args.steps * S, 1)
    # ---- the same region once more without the sharpening pass (SURVEY 8d asks for both; FLAGS_sharpening is a run-time
    # flag): fewer steps, same contexts, same frames. Secondary figure, never `value`.
    steps0 = max(2 * F, min(args.steps, 8))
    for c in ctxs:
        c.set_sharpening(0.0)
    for _ in range(F):
        step()
    drain()
    t0s = time.perf_counter()
    for _ in range(steps0):
        step()
    drain()
    dt0 = time.perf_counter() - t0s
    if dist is not None:
        t = torch.tensor([dt0], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt0 = float(t.item())
    for c in ctxs:
        c.set_sharpening(flags["sharpening"])
    for _ in range(F):  # the frames checked below are the sharpened ones again
        step()
    drain()
    # Per-kernel-family breakdown: one more step per context with the same contexts in flight, untimed, with the
    # library's per-family HIP events switched on (two event records per family scope perturb the launch stream,
    # which is why the timed steps above run without them).
    prof, prof_steps = {}, F
    for c in ctxs:
        c.profile_enable(True)
    for _ in range(prof_steps):
        step()
    drain(barrier=False)
    for c in ctxs:
        for k, v in c.profile_get().items():
            a = prof.get(k, (0.0, 0))
            prof[k] = (a[0] + v[0], a[1] + v[1])
        c.profile_enable(False)
    free_b, total_b = torch.cuda.mem_get_info(dev)
    hbm_used_gb = round((total_b - free_b) / 1e9, 1)

    # ---- check of the timed region: every in-flight frame against ONE context rendering the same inputs alone ----
    inflight_out = []
    for c in ctxs:
        for j in range(S):
            if S > 1:
                c.select_frame_slot(j)
            inflight_out.append(c.download_equirect())
    for c in ctxs[1:]:
        c.close()
    del ctxs[1:]
    batched_alone = None
    if S > 1:  # one batched context alone on the GPU: isolated durations of launches that hold S frames' flows
        ctx.render_batch(False)
        sync(barrier=False)
        tb = time.perf_counter()
        for _ in range(2):
            ctx.render_batch(False)
        sync(barrier=False)
        ms_per_batch = 1e3 * (time.perf_counter() - tb) / 2
        ctx.profile_enable(True)
        for _ in range(2):
            ctx.render_batch(False)
        sync(barrier=False)
        batched_alone = {"ms_per_batch": ms_per_batch,
                         "prof": {k: (v[0] / 2, v[1] / 2) for k, v in ctx.profile_get().items()}}
        ctx.profile_enable(False)
        ctx.select_frame_slot(0)
        ctx.set_frame_slots(1)
    ctx.set_sweep_mode("latency")
    mism = []
    single0 = None
    for k in range(F * S):
        ctx.upload_frame(*frames[k])
        ctx.render(False)
        alone = ctx.download_equirect()
        if k == 0:
            single0 = alone
        if not np.array_equal(alone, inflight_out[k]):
            mism.append(k)
    n_mism_all = len(mism)
    if dist is not None:  # the line is rank 0's: a mismatch on any rank must show in it
        t = torch.tensor([float(len(mism))], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        n_mism_all = int(t.item())
    checked = {"checked": n_mism_all == 0, "checked_frames": world * F * S, "mismatching_frames": mism,
               "mismatching_frames_all_ranks": n_mism_all,
               "check": "equirect of every in-flight frame of every rank (throughput sweep kernel; %d context(s) x %d slot(s) per "
                        "GPU) byte-compared with one context rendering the same inputs alone (latency sweep kernel)" % (F, S)}
    del inflight_out
    ctx.upload_frame(*frames[0])

    g = ctx.geometry
    side_px, pole_px = flow_px_levels(g, flags["eqr_width"])
    bytes_per_frame = sweep_algorithmic_bytes(g, 2 * P, 4, flags["eqr_width"])

    def isolated(mode, n):
        """n frames one at a time on the otherwise idle GPU: wall ms per frame + per-family (ms, launches) per frame."""
        ctx.set_sweep_mode(mode)
        ctx.render(False)
        sync(barrier=False)
        t1 = time.perf_counter()
        enq = 0.0
        for _ in range(n):  # wall time without event records
            te = time.perf_counter()
            ctx.render(False)
            enq += time.perf_counter() - te
        sync(barrier=False)
        ms = 1e3 * (time.perf_counter() - t1) / n
        ctx.profile_enable(True)
        for _ in range(n):  # the same frames again with the per-family HIP events
            ctx.render(False)
        sync(barrier=False)
        pr = {k: (v[0] / n, v[1] / n) for k, v in ctx.profile_get().items()}
        ctx.profile_enable(False)
        return ms, pr, 1e3 * enq / n

    def sweep_roofline(pr, kernel, note):
        ms, launches = pr.get("flow_sweep", (0.0, 0))
        per_launch_bytes = bytes_per_frame / max(launches, 1)
        avg_ms = ms / max(launches, 1)
        ach = per_launch_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        return {"bound": "hbm", "kernel": kernel, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": None,
                "bound_note": "reported against the HBM roofline as the contract asks; what bounds this kernel is the latency of its "
                              "dependency chain (BASELINE.md section 3, SURVEY 8d: the raster-order recurrence runs as an anti-diagonal "
                              "wavefront, a step is ~280 dependent instructions) — judge it on us per diagonal step and flows in flight", "avg_launch_ms": avg_ms, "launches_per_frame": launches,
                "algorithmic_bytes_per_launch": per_launch_bytes, "note": note}

    # ---- isolated kernels: one context alone, throughput-mode kernel (the timed region's) and latency-mode kernel ----
    if S > 1:
        tp_ms, tp_prof = None, {"flow_sweep": (1.0, 1)}  # replaced below by the batched context's own launches
    else:
        tp_ms, tp_prof, _ = isolated("throughput", 2)
    roofline = sweep_roofline(
        tp_prof, "k_sweep_quad (PixFlow propagation sweeps, PixFlow.h:388-410)",
        "dominant kernel of the timed region, measured with ONE context alone on the GPU (HIP events on the library's "
        "stream, launches do not overlap): algorithmic bytes per launch (48 B per pixel-level-sweep x the pixel-levels "
        "of the launch's flows: the 28 side or the 4 pole flows of each of the context's %d frame slots) / average launch "
        "duration. A dependency-latency-bound wavefront kernel (DESIGN.md §5): one launch is a serial chain of w+h "
        "diagonal steps; `aggregate_frac` is the same bytes over the wall time of the timed region, where the launches "
        "of %d contexts overlap" % (S, F))
    roofline["aggregate_frac"] = bytes_per_frame * args.steps * S / dt / 1e9 / HBM_PEAK_GBS
    roofline["aggregate_GBps"] = bytes_per_frame * args.steps * S / dt / 1e9
    if batched_alone:
        bms, bl = batched_alone["prof"].get("flow_sweep", (0.0, 0))
        per_launch = S * bytes_per_frame / max(bl, 1)
        ach = per_launch / (bms / max(bl, 1) * 1e-3) / 1e9 if bms > 0 else 0.0
        roofline.update({"achieved": ach, "frac": ach / HBM_PEAK_GBS, "avg_launch_ms": bms / max(bl, 1),
                         "launches_per_frame": bl / S, "algorithmic_bytes_per_launch": per_launch,
                         "frames_per_launch": S, "batch_alone_ms": batched_alone["ms_per_batch"],
                         "batch_alone_ms_per_frame": batched_alone["ms_per_batch"] / S,
                         "batch_alone_kernel_ms_per_frame": {k: round(v[0] / S, 3) for k, v in sorted(
                             batched_alone["prof"].items(), key=lambda kv: -kv[1][0])}})
    traffic_file = os.path.join(ROOT, "profiles", "sweep_traffic.json")
    if os.path.exists(traffic_file):  # HBM bytes per launch from the separate rocprofv3 --pmc passes (profiles/)
        try:
            tj = json.load(open(traffic_file))
            t = tj.get("kernels", {}).get("k_sweep_quad", {}).get("hbm_bytes_per_launch")
            fpl = tj.get("frames_per_launch", 1)
            if t:  # measured with `fpl` frames per launch; a launch's traffic is proportional to the frames it holds
                roofline["traffic"] = t * S / fpl
                roofline["traffic_source"] = "profiles/sweep_traffic.json (%d frame(s) per launch when profiled)" % fpl
        except Exception:
            pass

    # ---- what actually bounds the dominant kernel: its VALU issue (profiles/valu_busy.json, the committed SQ counter passes —
    # not measured by this run), per instantiation: <true, 3> holds the side flows of a launch (3 lanes per pixel), <true, 4> the
    # pole flows. valu_roof_frac = executed VALU wave-instructions per frame / (SIMDs x measured FMA issue rate x the sweeps' time).
    try:
        with open(os.path.join(ROOT, "profiles", "valu_busy.json")) as f:
            vbj = json.load(f)
        fs_ = vbj["families"]["flow_sweep"]
        ir = vbj["issue_rate"]
        roof = ir["simds"] * ir["valu_inst_per_cycle_per_simd"][fs_.get("rate_class", "fma")] * ir["clock_ghz"] * 1e9
        sweep_ms_frame = (batched_alone["prof"].get("flow_sweep", (0.0, 0))[0] / S) if batched_alone else tp_prof.get("flow_sweep", (0.0, 0))[0]
        if fs_.get("valu_insts_per_frame") and sweep_ms_frame > 0:
            roofline["valu_roof_frac"] = round(fs_["valu_insts_per_frame"] / roof / (sweep_ms_frame * 1e-3), 4)
            roofline["valu_rate_class"] = fs_.get("rate_class", "fma")
        roofline["valu_busy"] = fs_["valu_busy"]
        per = {}
        tot_ms = sum(r.get("ms", 0.0) for r in fs_.get("per_kernel", {}).values()) or 1.0
        for name, r in fs_.get("per_kernel", {}).items():
            key = "side_flows<true,3>" if "<true, 3>" in name else ("pole_flows<true,4>" if "<true, 4>" in name else name[:40])
            per[key] = {"valu_busy": r.get("valu_busy"), "issuing": r.get("wave_time_issuing"), "waiting": r.get("wave_time_waiting"),
                        "share_of_sweep_time": round(r.get("ms", 0.0) / tot_ms, 3)}
            vi = fs_.get("valu_insts_per_launch", {}).get(name)
            if vi and r.get("ms") and r.get("launches"):
                per[key]["valu_roof_frac"] = round(vi / roof / (r["ms"] / r["launches"] * 1e-3), 4)
        if per:
            roofline["per_instantiation"] = per
        roofline["valu_source"] = vbj["source"]
    except Exception:  # noqa: BLE001 - an annotation only
        pass
    # ---- the whole path against HBM: SURVEY 8d's compulsory bytes per frame (232 B per flow pixel-level + the warp/blend rows)
    path_bytes = 232 * (2 * P * side_px + 4 * pole_px) + sum(WARP_BLEND_MB.values()) * 1e6
    roofline["path_algorithmic_bytes_per_frame"] = path_bytes
    roofline["path_hbm_frac"] = path_bytes * args.steps * S / dt / 1e9 / HBM_PEAK_GBS
    if batched_alone:
        roofline["path_hbm_frac_batch_alone"] = path_bytes * S / (batched_alone["ms_per_batch"] * 1e-3) / 1e9 / HBM_PEAK_GBS

    out = {
        "metric": "stereo-equirect frames/sec at 8K, 17-cam rig",
        "value": world * args.steps * S / dt,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True,
        "value_note": "aggregate over INDEPENDENT frames (no predecessor: BASELINE configs[2]); temporally chained streams — every preset of "
                      "the reference — are the `video_streams_batched` leg, one stream the `video_stream` leg, one frame alone `single_frame` "
                      "(N GPUs: `sharded_frame`, configs[3], the one leg with RCCL exchanges)",
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (seeded %dx%d equirect world through the 17-camera rig, 2048x2048 inputs; every in-flight "
                "context holds a different frame of the stream)" % (2 * world_h, world_h),
        "config": {"workload": "BASELINE configs[2]: full 17-cam synthetic frame (2048x2048 inputs rendered from the seeded "
                               "%dx%d equirect world of SURVEY 8d), eqr 8400x4096 -> stereo 8192x8192, "
                               "top+bottom poles, pixflow_low, sharpening 0.25 (the reference's 8k preset, "
                               "batch_process_video.py:194-199)" % (2 * world_h, world_h),
                   "parallelism": "frames: %d GPU(s) x %d contexts x %d slots, no collective" % (world, F, S),
                   "parallelism_note": "independent frames: each GPU renders whole frames, %d contexts in flight per GPU (one HIP "
                                       "stream each) of %d frame slots, no data-path collective" % (F, S),
                   "frames_in_flight": F * S, "contexts": F, "slots_per_context": S, "frames_per_step": S},
        "roofline": roofline,
        "without_sharpening": {"value": world * steps0 * S / dt0, "unit": "frames/s", "steps": steps0,
                               "ms_per_step": 1e3 * dt0 / steps0,
                               "note": "the same contexts and frames with --sharpening 0 (SURVEY 8d: both figures)"},
        "kernel_ms_per_frame_in_flight": {k: round(v[0] / (prof_steps * S), 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])},
        "host": {"submit_threads": F, "enqueue_ms_per_frame": enqueue_ms_per_frame,
                 "settle_batches_before_warmup": settle["batches"], "settle_seconds": round(settle["seconds"], 2)},
    }
    out.update(checked)
    out["hbm_used_GB_in_timed_region"] = hbm_used_gb
    if dry:
        out["dry_run"] = "NOT A MEASUREMENT: the library emulated on the CPU at toy sizes (S360_TEST_EMULATED_LIB=1)"

    def emit():
        if rank == 0:
            print(json.dumps(out))
            sys.stdout.flush()

    # A hang or failure past this point must not cost the bench line: the timed region is complete, so a watchdog
    # prints what has been measured so far.
    import threading
    state = {"emitted": False}
    lock = threading.Lock()

    def bail(reason):
        with lock:
            if state["emitted"]:
                return
            state["emitted"] = True
            out.setdefault("errors", []).append(reason)
            emit()
        os._exit(0)

    # (N > 1: only the sharded frame follows — seconds; a stuck RCCL exchange must not outlast the driver's patience)
    watchdog = threading.Timer(900.0 if world == 1 else 420.0, bail, args=("post-timed-region phases timed out",))
    watchdog.daemon = True

    def stream(pipelined, n_frames, fetch, frame_of=None, barrier_at_lead=False):
        """One stream the way a streaming host runs it (host/TestRenderStereoPanorama --num_frames): the next frame's
        images are put into page-locked buffers by a second host thread (there: the PNG decoders write into them),
        uploaded in place, and the finished equirect of frame k-1 comes back into a page-locked buffer while frame k
        renders (the library releases the context while that call waits)."""
        ctx.set_frame_pipelining(pipelined)
        eq = R.pinned_empty((g.out_height, g.out_width, 3)) if fetch else None
        lead = min(10, n_frames - 2)
        up = 0.0
        t2 = None
        frame_of = frame_of or stream_frame
        f0 = frame_of(0)
        ring = [([R.pinned_empty(a.shape) for a in f0[0]], R.pinned_empty(f0[1].shape), R.pinned_empty(f0[2].shape))
                for _ in range(3)]

        def stage(k):
            side, top, bottom = frame_of(k)
            dst = ring[k % 3]
            for d, a in zip(dst[0], side):
                np.copyto(d, a)
            np.copyto(dst[1], top)
            np.copyto(dst[2], bottom)
        with ThreadPoolExecutor(1) as feeder:
            fut = feeder.submit(stage, 0)
            for k in range(n_frames):
                if k == lead:  # frames 0..lead-1 are the run-in (the first has no temporal state, buffers are sized)
                    sync(barrier=barrier_at_lead)
                    up = 0.0
                    t2 = time.perf_counter()
                tu = time.perf_counter()
                fut.result()
                if k + 1 < n_frames:
                    ctx.uploads_complete()  # (the buffer of frame k+1 was last read by frame k-2's uploads)
                    fut = feeder.submit(stage, k + 1)
                ctx.upload_frame(*ring[k % 3])  # page-locked: sent in place on the upload stream; overlaps frame k-1
                up += time.perf_counter() - tu
                ctx.render(k > 0)
                if fetch and k > 0:  # frame k-1 comes back while k renders
                    ctx.download_equirect_of(1, eq)
            if fetch:
                eq = np.array(ctx.download_equirect())
            sync(barrier=False)
        n = n_frames - lead
        ms = 1e3 * (time.perf_counter() - t2) / n
        ctx.set_frame_pipelining(False)
        ctx.uploads_complete()
        return {"frames": n_frames, "steady_state_frames": n, "ms_per_frame": ms, "frames_per_s": 1e3 / ms,
                "host_upload_ms_per_frame": 1e3 * up / n, "equirect_fetched_per_frame": bool(fetch)}, eq

    watchdog.start()
    try:
        # ---- one frame at a time (configs[2] on 1 GPU; configs[3] = pairs sharded + strip gather on N GPUs) ----
        bounds = parallel.partition_pairs(P, world)
        p0, p1 = bounds[rank], bounds[rank + 1]
        if world == 1:
            lat_ms, lat_prof, lat_enq = isolated("latency", 3)
            n_diag = 2 * (sum(w + h - 1 for w, h in pyramid_levels(g.overlap_image_width, g.cam_image_height)) +
                          sum(w + h - 1 for w, h in pyramid_levels(int(np.float32(flags["eqr_width"]) * np.float32(1.2)), g.top_rows)))
            lock_roof = sweep_roofline(lat_prof, "k_sweep_lock", "latency-mode sweep kernel, one frame alone")
            wb = {}
            for k, mb in WARP_BLEND_MB.items():
                ms = lat_prof.get(k, (0.0, 0))[0]
                if ms > 0:
                    gbs = mb * 1e6 / (ms * 1e-3) / 1e9
                    wb[k] = {"ms_per_frame": round(ms, 3), "achieved_GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
            fs = {}
            pxl = 28 * side_px + 4 * pole_px
            for k, b in FLOW_STENCIL_B_PER_PXLEVEL.items():
                ms = lat_prof.get(k, (0.0, 0))[0]
                if ms > 0:
                    # gradients are per image (28 side images, 6 pole images: 4 B read + 8 B written each)
                    nbytes = 12 * (28 * side_px + 6 * pole_px) if k == "flow_gradients" else b * pxl
                    gbs = nbytes / (ms * 1e-3) / 1e9
                    fs[k] = {"ms_per_frame": round(ms, 3), "achieved_GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}
            # counter bytes of the stencil families beside their algorithmic fractions (profiles/stencil_traffic.json: the
            # committed FETCH_SIZE / WRITE_SIZE passes, gfx950 correction applied; not measured by this run)
            try:
                with open(os.path.join(ROOT, "profiles", "stencil_traffic.json")) as f:
                    stj = json.load(f)
                for k, rec in fs.items():
                    tr = stj["families"].get(k)
                    if tr:
                        rec["traffic_bytes_per_frame"] = round(tr["hbm_bytes_per_frame"])
                        rec["traffic_GBps"] = round(tr["hbm_bytes_per_frame"] / (rec["ms_per_frame"] * 1e-3) / 1e9, 1)
                        rec["traffic_frac"] = round(rec["traffic_GBps"] / HBM_PEAK_GBS, 4)
            except Exception:  # noqa: BLE001 - an annotation only
                pass
            # Beside each HBM fraction the kernel family's SQ "VALU busy" figure from the committed counter pass (profiles/valu_busy.json;
            # not measured by this run): a VALU-bound kernel is then shown at its bound rather than asserted to be there.
            vb_note = None
            try:
                with open(os.path.join(ROOT, "profiles", "valu_busy.json")) as f:
                    vbj = json.load(f)
                for group in (wb, fs):
                    for k, rec in group.items():
                        if k in vbj["families"]:
                            rec["valu_busy_profiled"] = vbj["families"][k]["valu_busy"]
                            ipf, ir = vbj["families"][k].get("valu_insts_per_frame"), vbj.get("issue_rate")
                            if ipf and ir:  # executed VALU wave-instructions per frame against what the chip issues in the family's time
                                cls = vbj["families"][k].get("rate_class", "fma")  # perm / dot2 / min / max / integer ops issue at half the FMA rate
                                roof = ir["simds"] * ir["valu_inst_per_cycle_per_simd"][cls] * ir["clock_ghz"] * 1e9
                                rec["valu_rate_class"] = cls
                                rec["valu_roof_frac"] = round(ipf / roof / (rec["ms_per_frame"] * 1e-3), 4)
                vb_note = vbj["source"]
            except Exception:  # noqa: BLE001 - an annotation only
                pass
            non_sweep = sum(v[0] for k, v in lat_prof.items() if k != "flow_sweep")
            out["single_frame"] = {
                "mode": "configs[2]: one frame at a time, all pairs on 1 GPU, latency sweep kernel",
                "ms": lat_ms, "frames_per_s": 1e3 / lat_ms, "enqueue_ms": lat_enq,
                "sweep": {"kernel": "k_sweep_lock", "ms_per_frame": lat_prof.get("flow_sweep", (0, 0))[0],
                          "roofline_frac": lock_roof["frac"], "avg_launch_ms": lock_roof["avg_launch_ms"],
                          "serial_diagonal_steps_per_frame": n_diag,
                          "us_per_diagonal_step": 1e3 * lat_prof.get("flow_sweep", (0, 0))[0] / n_diag},
                "kernel_ms_non_sweep": round(non_sweep, 3),
                "kernel_ms_per_frame": {k: round(v[0], 3) for k, v in sorted(lat_prof.items(), key=lambda kv: -kv[1][0])},
                "warp_blend_roofline": wb, "flow_stencil_roofline": fs, "valu_busy_profiled_source": vb_note,
                "throughput_kernel_alone_ms": tp_ms}
        else:
            # configs[3] (one frame sharded over the ranks, the native RCCL exchanges) runs in a CHILD process per rank, on a
            # rendezvous of its own: that path has never run on more than one real GPU, and a crash or a stuck exchange in
            # it must not cost the line measured above. Every rank starts its child here, right behind the timed region.
            import subprocess
            env = dict(os.environ, MASTER_PORT=str(int(os.environ.get("MASTER_PORT", "29500")) + 1))
            # (torchrun's workers rendezvous through the agent's store; the children make a store of their own: rank 0's
            # child serves it on the new port)
            env["TORCHELASTIC_USE_AGENT_STORE"] = "False"
            for k in ("TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS", "TORCHELASTIC_ERROR_FILE"):
                env.pop(k, None)
            cmd = [sys.executable, os.path.abspath(__file__), "--sharded-child", "--gpus", str(world)]
            try:
                r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
                lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                rec = json.loads(lines[-1])["sharded_frame"] if lines else {"error": "rc %d: %s" % (r.returncode, r.stderr[-400:])}
            except subprocess.TimeoutExpired:
                rec = {"error": "the sharded-frame processes did not finish in 240 s"}
            if rank == 0:
                out["sharded_frame"] = rec
                # the only communicator of this run is the sharded frame's: its size as RCCL reports it (absent if that leg failed)
                if "rccl_ranks" in rec:
                    out["config"]["rccl_ranks"] = rec["rccl_ranks"]
                    out["config"]["rccl_ranks_note"] = "communicator of the `sharded_frame` leg (ncclCommCount); the timed region has none"
            # ---- configs[4] on N GPUs, the honest form: ONE stream cannot use more than one GPU (its pole flows are one serial
            # chain per frame and every frame needs its predecessor's flows: DESIGN.md sections 5 and 7), N streams use N —
            # every rank runs a stream of its own on its GPU at the same time (what host/TestRenderStereoPanorama
            # --num_streams N does in one process); no collective, the ranks only meet for the barrier and the figures ----
            if not args.no_extras:
                try:
                    n_multi = max(14, min(args.video_frames, 40))
                    held = len(frames)

                    def ring_frame(k):  # the frames held in host memory walked forwards and backwards
                        if held < 2:
                            return frames[0]
                        m = k % (2 * (held - 1))
                        return frames[m if m < held else 2 * (held - 1) - m]
                    ctx.set_sweep_mode("latency")
                    lead = min(10, n_multi - 2)
                    rec_r, _ = stream(True, n_multi, True, frame_of=lambda k: ring_frame(k + rank), barrier_at_lead=True)
                    ms_r = rec_r["ms_per_frame"]
                    tms = torch.tensor([ms_r], dtype=torch.float64, device=red_dev)
                    allms = [torch.zeros_like(tms) for _ in range(world)]
                    dist.all_gather(allms, tms)
                    per = [float(t.item()) for t in allms]
                    if rank == 0:
                        out["video_stream"] = {
                            "mode": "%d streams, one per GPU, at the same time: each %d frames with temporal regularisation, frame "
                                    "pipelining, inputs uploaded from host memory and the finished equirect fetched while the next "
                                    "frame renders; steady state = frames %d..%d; no collective" % (world, n_multi, lead, n_multi - 1),
                            "streams": world, "ms_per_frame_of_each_stream": [round(v, 2) for v in per],
                            "frames_per_s": sum(1e3 / v for v in per), "frames_per_s_per_stream": [round(1e3 / v, 3) for v in per]}
                except Exception as e:  # noqa: BLE001
                    if rank == 0:
                        out["video_stream"] = {"error": repr(e)}

        if world == 1 and not args.no_extras:
            # ---- the same single frame without the sharpening pass (secondary; the presets all sharpen) ----
            ctx.set_sharpening(0.0)
            ctx.render(False)
            sync(barrier=False)
            t1 = time.perf_counter()
            for _ in range(3):
                ctx.render(False)
            sync(barrier=False)
            ms = 1e3 * (time.perf_counter() - t1) / 3
            ctx.profile_enable(True)
            for _ in range(3):
                ctx.render(False)
            sync(barrier=False)
            pr = ctx.profile_get()
            ctx.profile_enable(False)
            ctx.set_sharpening(flags["sharpening"])
            out["single_frame"]["without_sharpening"] = {"ms": ms, "finish_ms": pr.get("finish", (0, 0))[0] / 3,
                                                         "finish_ms_with": out["single_frame"]["kernel_ms_per_frame"].get("finish")}

            # ---- BASELINE configs[1]: one 2048x2048 pair, both directions (TestOpticalFlow.cpp:50-143) ----
            i0, i1 = synth.flow_pair(pair_size, pair_size, seed=360)
            cf = R.Context(rig, R.make_params(), device=local_rank)
            cf.compute_optical_flow(i0, i1, "pixflow_low", "LEFT")
            t1 = time.perf_counter()
            gl = cf.compute_optical_flow(i0, i1, "pixflow_low", "LEFT")
            gr = cf.compute_optical_flow(i1, i0, "pixflow_low", "RIGHT")
            gpu_s = time.perf_counter() - t1
            cf.close()
            c2 = {"workload": "BASELINE configs[1]: one 2048x2048 BGRA pair, pixflow_low, flowLtoR + flowRtoL",
                  "gpu_runtime_sec": gpu_s, "gpu_note": "host buffers in and out (PCIe both ways), latency sweep kernel"}
            if not args.no_cpu_baseline:
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import oracle_lib as O
                with ThreadPoolExecutor(2) as ex:
                    t1 = time.perf_counter()
                    fl = ex.submit(O.compute_optical_flow, i0, i1, "pixflow_low", "LEFT")
                    fr = ex.submit(O.compute_optical_flow, i1, i0, "pixflow_low", "RIGHT")
                    wl, wr = fl.result(), fr.result()
                    c2["cpu_runtime_sec"] = time.perf_counter() - t1
                c2["cpu_note"] = "oracle, the two directions on two threads (NovelView.cpp:282-297 runs them back to back)"
                c2["checked"] = bool(np.array_equal(gl.view(np.uint32), wl.view(np.uint32)) and
                                     np.array_equal(gr.view(np.uint32), wr.view(np.uint32)))
            out["config2_flow_pair"] = c2

            # ---- BASELINE configs[4] on one GPU: one video stream of 190 frames (Building-20 shape),
            # temporal regularisation, steady state over frames 10..N-1, the finished frame fetched while the next renders ----
            ctx.set_sweep_mode("latency")
            video = {"mode": "one stream of %d frames (%d distinct in host memory%s; world rotating 0.2 deg/frame, one moving disc); "
                             "frame k regularised toward frame k-1's device-resident flows and images; inputs uploaded from host "
                             "memory while the previous frame renders; sharpening 0.25; steady state = frames 10..%d" % (
                                 n_video, n_distinct, "" if n_distinct >= n_video else ", walked forwards and backwards",
                                 n_video - 1)}
            unp, _ = stream(False, min(n_video, 40), False)
            video["unpipelined"] = unp
            pip, last_eq = stream(True, n_video, True)
            video.update(pip)
            # the per-frame state spill (.bin flows + flow_images of TRSP:201-255, 413-452): a stream keeps that state on the
            # device; writing it for every frame like the reference costs one download + write of 32 flows and 36 images
            try:
                import shutil
                import tempfile
                tmpd = tempfile.mkdtemp(prefix="s360_spill_")
                t1 = time.perf_counter()
                fl = [ctx.get_f32(nm, i) for nm in ("flow_l_to_r", "flow_r_to_l") for i in range(P)]
                fl += [ctx.get_f32("flow_pole", u) for u in range(4)]
                im = [ctx.get_u8(nm, i) for nm in ("overlap_l", "overlap_r") for i in range(P)]
                im += [ctx.get_u8(nm, u) for nm in ("extended_side", "extended_fisheye") for u in range(4)]
                t_dl = time.perf_counter() - t1
                t1 = time.perf_counter()
                for i, f in enumerate(fl):
                    R.save_flow_to_file(f, os.path.join(tmpd, "flow_%d.bin" % i))
                t_wr = time.perf_counter() - t1
                nbytes = sum(f.nbytes for f in fl)
                shutil.rmtree(tmpd, ignore_errors=True)
                video["spill_ms_per_frame"] = round(1e3 * (t_dl + t_wr), 1)
                video["spill"] = {"download_ms": round(1e3 * t_dl, 1), "bin_write_ms": round(1e3 * t_wr, 1),
                                  "flow_bytes": nbytes, "image_bytes": sum(i.nbytes for i in im),
                                  "note": "one frame's temporal state fetched to host memory (28 side + 4 pole flows, 28 overlap + 8 "
                                          "extended pole images) and the 32 flows written as the reference's .bin files to a temporary "
                                          "directory; PNG encoding of the state images not included. The stream above does not pay "
                                          "this: its state stays on the device (host/TestRenderStereoPanorama --num_frames writes it "
                                          "after the last frame only)"}
                del fl, im
            except Exception as e:  # noqa: BLE001
                video["spill"] = {"error": repr(e)}
            out["video_stream"] = video

            # ---- the same stream END TO END through the drop-in host program: PNG files in, PNG files out ----
            try:
                n_e2e = min(14, n_distinct)
                prog = os.path.join(ROOT, "tools", "emu", "TestRenderStereoPanorama") if dry else \
                    os.path.join(ROOT, "host", "TestRenderStereoPanorama")

                bins_leg = make_bins_leg(lambda: ctx, rig_path, stream_frame, dry, local_rank)
                res = host_program_stream([stream_frame(k) for k in range(n_e2e)], rig_path, flags, prog, device=local_rank, bins_leg=bins_leg)
                if isinstance(res, tuple):
                    rec, last_png = res
                    _, eq_chain = stream(True, n_e2e, True)  # the same chain through the C ABI in this process
                    rec["last_frame_equals_in_process_stream"] = bool(eq_chain.shape == last_png.shape and
                                                                      np.array_equal(eq_chain, last_png))
                    # the same with every file in memory (/dev/shm): what the program does when the scratch file system is not
                    # the limit (tools/host_io_time measures that file system alone)
                    try:
                        import shutil as _sh
                        if os.path.isdir("/dev/shm") and _sh.disk_usage("/dev/shm").free > (6 << 30):
                            res2 = host_program_stream([stream_frame(k) for k in range(n_e2e)], rig_path, flags, prog,
                                                       device=local_rank, scratch="/dev/shm")
                            r2 = res2[0] if isinstance(res2, tuple) else res2
                            rec["files_in_memory"] = {k: r2[k] for k in ("ms_per_frame_steady", "frames_per_s_steady", "ms_per_frame_stream",
                                                                         "host_thread_ms_per_frame", "process_wall_s", "error") if k in r2}
                            if isinstance(res2, tuple):
                                rec["files_in_memory"]["same_last_frame"] = bool(np.array_equal(res2[1], last_png))
                    except Exception as e:  # noqa: BLE001
                        rec["files_in_memory"] = {"error": repr(e)}
                    out["end_to_end_files"] = rec
                else:
                    out["end_to_end_files"] = res
            except Exception as e:  # noqa: BLE001
                out["end_to_end_files"] = {"error": repr(e)}

            if not args.no_cpu_baseline:
                ref = None
                try:
                    ref = cpu_baseline_reference(*frames[0], rig_path=rig_path, flags=flags)
                except Exception as e:  # noqa: BLE001
                    out.setdefault("errors", []).append("reference program baseline failed: %r" % (e,))
                if ref is not None:
                    want, cb = ref
                else:  # no oracle/_ref on this machine: the oracle port of the same path
                    want, cb = cpu_baseline_8k(*frames[0], rig_path=rig_path, flags=flags)
                cb["checked_against_gpu"] = bool(want.shape == single0.shape and np.array_equal(want, single0))
                out["cpu_baseline"] = cb

            # ---- ISP front end: raw 2048x2048 Bayer frames -> BGR, alone and straight into a frame ----
            # In a process of its own (tools/isp_time.py): informative, and nothing it does can cost the lines above.
            try:
                import subprocess
                cmd = [sys.executable, os.path.join(ROOT, "tools", "isp_time.py"), "--json", "--device", str(local_rank)]
                if args.no_cpu_baseline:
                    cmd.append("--no-cpu")
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
                lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                out["isp"] = json.loads(lines[-1]) if lines else {"error": "rc %d: %s" % (r.returncode, r.stderr[-300:])}
            except Exception as e:  # noqa: BLE001
                out["isp"] = {"error": repr(e)}

            # ---- the reference's REAL workload as a batch (round 5): every preset renders frame k with --prev_frame_data_dir
            # (batch_process_video.py:157-158, TRSP:215-235, 421-436, PixFlow.h:101-118, 185-193). S streams in the frame slots of
            # two contexts, every step ONE s360_frame_render_batch(use_prev=1) per context: frame k of every stream regularised
            # toward its own stream's device-resident previous flows and images (the A13 kernels at every pyramid level, both
            # halves of every slot's temporal double buffers resident). The headline's frames have no predecessor; these do. Last
            # leg of the run, in fresh contexts (the others are closed first): nothing it does can cost the lines above.
            try:
                ctx.close()
                torch.cuda.empty_cache()
                out["video_streams_batched"] = streams_batched(R, rig, flags, local_rank, frames, args, dry, g)
            except Exception as e:  # noqa: BLE001
                import traceback
                out["video_streams_batched"] = {"error": repr(e), "trace": traceback.format_exc()[-500:]}

            # ---- what a frame costs in a third of the memory: `value` and the chained figure hold 250-280 GB of the 288; the same
            # two workloads with fewer frame slots, measured here in fresh contexts (short legs: the rows are throughput against HBM
            # held, the headline's and the full chained leg's own checks cover the kernels) ----
            try:
                table = [{"workload": "independent frames (the headline)", "slots_per_context": S, "contexts": F,
                          "frames_per_s": out["value"], "hbm_used_GB": hbm_used_gb}]

                def independent_row(S2, nctx=F, pipelined=False):
                    cs = [R.Context(rig, R.make_params(**flags), device=local_rank) for _ in range(nctx)]
                    try:
                        for k, c in enumerate(cs):
                            c.set_frame_slots(S2)
                            for j in range(S2):
                                c.select_frame_slot(j)
                                c.upload_frame(*frames[k * S2 + j])
                            c.set_sweep_mode("throughput")
                            if pipelined:  # batch k's pole stage on a second stream beside batch k+1's side stage
                                c.set_frame_pipelining(True)
                        with ThreadPoolExecutor(nctx) as pool:
                            def batch(n):
                                for _ in range(n):
                                    list(pool.map(lambda c: c.render_batch(False), cs))
                                for c in cs:
                                    c.synchronize()
                            batch(2)
                            used = 0.0 if dry else (torch.cuda.mem_get_info(dev)[1] - torch.cuda.mem_get_info(dev)[0]) / 1e9
                            t = time.perf_counter()
                            nb = 2 if dry else 6
                            batch(nb)
                            dts = time.perf_counter() - t
                        cs[0].select_frame_slot(0)
                        same = bool(np.array_equal(cs[0].download_equirect(), single0))
                        return {"workload": "independent frames" + (", batches pipelined inside the context (s360_set_frame_pipelining)" if pipelined else ""),
                                "slots_per_context": S2, "contexts": nctx, "frames_per_s": nb * nctx * S2 / dts,
                                "hbm_used_GB": round(used, 1), "frame_0_equals_single": same}
                    finally:
                        for c in cs:
                            c.close()
                        if not dry:
                            torch.cuda.empty_cache()
                for S2 in ([1] if dry else [8]):
                    if S2 < S or dry:
                        table.append(independent_row(S2))
                table.append(independent_row(2 if dry else S, nctx=1, pipelined=True))  # ONE context: what the second one buys, for less HBM
                vb = out.get("video_streams_batched", {})
                if "frames_per_s" in vb:
                    table.append({"workload": "temporally chained streams (video_streams_batched)", "slots_per_context": vb["slots_per_context"],
                                  "contexts": 2, "frames_per_s": vb["frames_per_s"], "hbm_used_GB": vb["hbm_used_GB"]})
                for S2 in ([1] if dry else [10, 6]):
                    if dry or S2 < vb.get("slots_per_context", 0):
                        r = streams_batched(R, rig, flags, local_rank, frames, args, dry, g, slots=S2, timed_steps=4, check=False)
                        table.append({"workload": "temporally chained streams", "slots_per_context": r["slots_per_context"], "contexts": 2,
                                      "frames_per_s": r["frames_per_s"], "hbm_used_GB": r["hbm_used_GB"]})
                for pl in (False, True):  # ONE context, plain and with its steps frame-pipelined
                    r = streams_batched(R, rig, flags, local_rank, frames, args, dry, g, slots=2 if dry else 22, timed_steps=4, check=False,
                                        contexts=1, pipelined=pl)
                    table.append({"workload": "temporally chained streams" + (", steps pipelined inside the context (s360_set_frame_pipelining)" if pl else ""),
                                  "slots_per_context": r["slots_per_context"], "contexts": 1, "frames_per_s": r["frames_per_s"],
                                  "hbm_used_GB": r["hbm_used_GB"]})
                out["slots_table"] = table
            except Exception as e:  # noqa: BLE001
                import traceback
                out["slots_table"] = {"error": repr(e), "trace": traceback.format_exc()[-400:]}

    except Exception as e:  # noqa: BLE001 - reported in the JSON line
        import traceback
        bail("post-timed-region phase failed: %r %s" % (e, traceback.format_exc()[-600:]))
    watchdog.cancel()
    with lock:
        if state["emitted"]:
            return
        state["emitted"] = True
    emit()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
