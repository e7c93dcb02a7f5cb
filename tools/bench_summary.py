#!/usr/bin/env python
"""The figures of a bench.py JSON line a builder looks at first (tools/gpu_job.sh prints them into the GPU call's tail)."""
import json
import sys

d = json.loads([ln for ln in open(sys.argv[1]) if ln.startswith("{")][-1])
r = d.get("roofline", {})
print("value %.2f %s  ms/step %.1f  checked %s  roofline %.4f (%.3f ms/launch)  hbm %s GB" % (
    d["value"], d["unit"], d["ms_per_step"], d.get("checked"), r.get("frac", 0), r.get("avg_launch_ms", 0), d.get("hbm_used_GB_in_timed_region")))
sf = d.get("single_frame", {})
if "ms" in sf:
    print("single frame %.1f ms, sweep %.3f us/step; warp/blend: %s" % (
        sf["ms"], sf.get("sweep", {}).get("us_per_diagonal_step", 0),
        {k: (v["ms_per_frame"], v["frac"], v.get("valu_roof_frac")) for k, v in sf.get("warp_blend_roofline", {}).items()}))
for key in ("video_stream", "video_streams_batched", "end_to_end_files", "cpu_baseline", "isp"):
    v = d.get(key)
    if isinstance(v, dict):
        keep = {k: v[k] for k in ("frames_per_s", "ms_per_frame", "streams", "slots_per_context", "hbm_used_GB", "checked",
                                  "temporal_kernels_share_in_flight", "mismatching_streams", "check_seconds", "retries", "error",
                                  "ms_per_frame_steady", "process_wall_s", "single_invocation_s", "single_invocation", "value",
                                  "checked_against_gpu", "last_frame_equals_in_process_stream") if k in v}
        print(key, json.dumps(keep))
if "errors" in d:
    print("ERRORS", d["errors"])
