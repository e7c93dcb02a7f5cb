#!/bin/bash
# Regenerates the rocprofv3 summaries under profiles/ on a GPU box (run from the repo root, e.g. through gpurun;
# gpurun_out/ is scratch). Usage: tools/make_profiles.sh <tag> [slots]     e.g. tools/make_profiles.sh r02_v1 12
#   1. <tag>_kernel_stats.txt          kernel-trace summary of the DEFAULT bench command (contexts overlap: durations
#                                      of launches that share the GPU are longer than isolated ones)
#   2. <tag>_isolated_kernel_stats.txt the same workload with ONE context alone (--inflight 1): launches do not overlap,
#                                      so the average duration of k_sweep_quad here is the bench line's
#                                      roofline.avg_launch_ms (HIP events) measured a second way
#   3. <tag>_pmc_fetch_write.txt       FETCH_SIZE and WRITE_SIZE in separate --pmc passes of the isolated command
#                                      (counters are never combined with other trace domains)
#   4. profiles/sweep_traffic.json     HBM bytes per sweep launch (gfx950: FETCH_SIZE x2 on the read side)
#   5. profiles/stencil_traffic.json   HBM bytes per frame of the flow-stencil kernel families, same passes, same correction
set -e
TAG=${1:?tag}
SLOTS=${2:-22}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=${S360_PROF_DIR:-/tmp/s360_prof}/prof_$TAG  # (rocprofv3 databases: hundreds of MB — outside gpurun_out/, which is merged back and capped)
mkdir -p $OUT profiles
ISO="python bench.py --inflight 1 --slots $SLOTS --steps 2 --warmup 1 --no-extras --no-cpu-baseline"
# (--no-extras since round 5: the timed region + its check + the single-frame figures; the stream legs would add minutes of tracing)
rocprofv3 --kernel-trace --stats -d $OUT/ks -o ks -- python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline > $OUT/ks.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/iso -o iso -- $ISO > $OUT/iso.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/f -o f -- $ISO > $OUT/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/w -o w -- $ISO > $OUT/w.log 2>&1
python tools/rocpd_kernel_stats.py $OUT/ks/ks_results.db \
  "rocprofv3 --kernel-trace --stats summary ($TAG): python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline (default: 2 contexts x 22 frame slots)" \
  "durations from the rocpd kernel dispatch table; launches of the two contexts overlap, so these are NOT isolated durations" \
  > profiles/${TAG}_kernel_stats.txt
python tools/rocpd_kernel_stats.py $OUT/iso/iso_results.db \
  "rocprofv3 --kernel-trace --stats summary ($TAG): $ISO" \
  "ONE context alone: launches do not overlap. k_sweep_quad avg_us here = roofline.avg_launch_ms of the bench line; every k_sweep_quad launch holds the flows of $SLOTS frames" \
  > profiles/${TAG}_isolated_kernel_stats.txt
grep '^{"metric"' $OUT/iso.log | tail -1 > profiles/${TAG}_isolated_bench.json || true
{
  echo "# rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) and --pmc WRITE_SIZE (pass 2) on: $ISO ($TAG)"
  echo "# Units: KB as reported by rocprofv3. gfx950 (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts wide coalesced reads at 1/2 of their bytes."
  python tools/rocpd_pmc.py $OUT/f/f_results.db FETCH_SIZE | head -24
  echo
  python tools/rocpd_pmc.py $OUT/w/w_results.db WRITE_SIZE | head -24
} > profiles/${TAG}_pmc_fetch_write.txt
python tools/rocpd_pmc.py $OUT/f/f_results.db FETCH_SIZE > $OUT/fetch_full.txt
python tools/rocpd_pmc.py $OUT/w/w_results.db WRITE_SIZE > $OUT/write_full.txt
python - "$TAG" "$SLOTS" "$OUT" <<'PY'
import json, sys
tag, slots, outdir = sys.argv[1], int(sys.argv[2]), sys.argv[3]
txt = open("profiles/%s_pmc_fetch_write.txt" % tag).read()
fetch, write = txt.split("\n\n", 1)
def per_launch(block, kernel):
    """launches and KB per launch over ALL instantiations of a kernel template (k_sweep_quad<true, 3> and <true, 4> are
    the same kernel with three / four lanes per pixel: FlowEngine picks one per launch shape)"""
    n, tot = 0, 0.0
    for line in block.splitlines():
        if kernel in line:
            f = line.split()
            n += int(f[-3])
            tot += float(f[-2])
    return n, (tot / n if n else 0.0)
d = {"source": "profiles/%s_pmc_fetch_write.txt (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --inflight 1 --slots %d --steps 2 --warmup 1 --no-extras)" % (tag, slots),
     "correction": "gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes (MI355X_MICROARCH.md, HBM) -> read side doubled; WRITE_SIZE as reported",
     "frames_per_launch": slots, "kernels": {}}
for key, kname in (("k_sweep_lock", "k_sweep_lock<"), ("k_sweep_quad", "k_sweep_quad<")):
    n, f = per_launch(fetch, kname)
    _, w = per_launch(write, kname)
    if n:
        d["kernels"][key] = {"launches_profiled": n, "fetch_size_kb_per_launch": f, "write_size_kb_per_launch": w,
                             "hbm_bytes_per_launch_raw": (f + w) * 1024, "hbm_bytes_per_launch": (2 * f + w) * 1024}
if "k_sweep_quad" in d["kernels"]:
    d["hbm_bytes_per_launch"] = d["kernels"]["k_sweep_quad"]["hbm_bytes_per_launch"]
json.dump(d, open("profiles/sweep_traffic.json", "w"), indent=1)
# counter bytes of the flow-stencil families (bench.py puts them beside the algorithmic fractions): all launches of the profiled
# run / the frames it rendered (one k_novel_view launch per frame)
FAMILIES = {"flow_median": ["k_median5"], "flow_diffusion": ["k_sepblur<7, 2, 1,"], "flow_blur15": ["k_sepblur<7, 2, 2,", "k_sepblur<7, 2, 3,"],
            "flow_upscale": ["k_resize_cubic_f32c2"], "flow_gradients": ["k_sepblur<1, 2, 0, 1", "k_sepblur<1, 2, 0, 2"],
            "flow_pyramid": ["k_resize_linear_f32c1"], "flow_sweep": ["k_sweep_quad<"]}
def total(block, pats):
    n, tot = 0, 0.0
    for line in block.splitlines():
        if any(p in line for p in pats):
            f = line.split()
            n += int(f[-3])
            tot += float(f[-2])
    return n, tot
fetch_all, write_all = open(outdir + "/fetch_full.txt").read(), open(outdir + "/write_full.txt").read()  # (every kernel, not the top 24)
frames, _ = total(fetch_all, ["k_novel_view"])
fam = {}
for name, pats in FAMILIES.items():
    n, fk = total(fetch_all, pats)
    _, wk = total(write_all, pats)
    if n and frames:
        fam[name] = {"launches_profiled": n, "hbm_bytes_per_frame": (2 * fk + wk) * 1024 / frames,
                     "fetch_bytes_per_frame_corrected": 2 * fk * 1024 / frames, "write_bytes_per_frame": wk * 1024 / frames}
json.dump({"source": d["source"], "correction": d["correction"], "frames_profiled": frames, "families": fam},
          open("profiles/stencil_traffic.json", "w"), indent=1)
print("profiles written for", tag)
PY
