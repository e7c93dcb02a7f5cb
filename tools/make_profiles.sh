#!/bin/bash
# Regenerates the rocprofv3 summaries under profiles/ on a GPU box (run from the repo root, e.g. through gpurun;
# gpurun_out/ is scratch). Usage: tools/make_profiles.sh <tag>     e.g. tools/make_profiles.sh r02_v1
#   1. kernel-trace summary of the default bench configuration (16 frames in flight + single-frame phases)
#   2. FETCH_SIZE and WRITE_SIZE in separate --pmc passes (counters are never combined with other trace domains)
#   3. profiles/sweep_traffic.json (HBM bytes per sweep launch, gfx950 FETCH_SIZE correction x2 on the read side)
set -e
TAG=${1:?tag}
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT profiles
rocprofv3 --kernel-trace --stats -d $OUT/ks -o ks -- python bench.py --steps 16 --warmup 16 --no-cpu-baseline > $OUT/ks.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/f -o f -- python bench.py --steps 1 --warmup 1 --inflight 2 --no-cpu-baseline > $OUT/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/w -o w -- python bench.py --steps 1 --warmup 1 --inflight 2 --no-cpu-baseline > $OUT/w.log 2>&1
{
  echo "# rocprofv3 --kernel-trace --stats summary ($TAG): python bench.py --steps 16 --warmup 16 --no-cpu-baseline"
  echo "# durations from the rocpd kernel dispatch table (launches of different frames overlap, so the sum exceeds wall time)"
  python tools/rocpd_kernel_stats.py $OUT/ks/ks_results.db
} > profiles/${TAG}_kernel_stats.txt
{
  echo "# rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) and --pmc WRITE_SIZE (pass 2) on: python bench.py --steps 1 --warmup 1 --inflight 2 --no-cpu-baseline ($TAG)"
  echo "# Units: KB as reported by rocprofv3. gfx950 (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts wide coalesced reads at 1/2 of their bytes."
  python tools/rocpd_pmc.py $OUT/f/f_results.db FETCH_SIZE | head -24
  echo
  python tools/rocpd_pmc.py $OUT/w/w_results.db WRITE_SIZE | head -24
} > profiles/${TAG}_pmc_fetch_write.txt
python - "$TAG" <<'PY'
import json, re, sys
tag = sys.argv[1]
txt = open("profiles/%s_pmc_fetch_write.txt" % tag).read()
fetch, write = txt.split("\n\n", 1)
def per_launch(block, kernel):
    for line in block.splitlines():
        if kernel in line:
            f = line.split()
            return int(f[-3]), float(f[-1])
    return 0, 0.0
d = json.load(open("profiles/sweep_traffic.json"))
d["source"] = "profiles/%s_pmc_fetch_write.txt (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --steps 1 --warmup 1 --inflight 2)" % tag
for key, kname in (("k_sweep_lock", "k_sweep_lock<"), ("k_sweep_quad", "k_sweep_quad<")):
    n, f = per_launch(fetch, kname)
    _, w = per_launch(write, kname)
    if n:
        d["kernels"][key].update({"launches_profiled": n, "fetch_size_kb_per_launch": f, "write_size_kb_per_launch": w,
                                  "hbm_bytes_per_launch_raw": (f + w) * 1024, "hbm_bytes_per_launch": (2 * f + w) * 1024})
d["hbm_bytes_per_launch"] = d["kernels"]["k_sweep_quad"]["hbm_bytes_per_launch"]
json.dump(d, open("profiles/sweep_traffic.json", "w"), indent=1)
print("profiles written for", tag)
PY
