#!/bin/bash
# A/B of two builds of the library on ONE box, alternating: surround360_amd/libs360_old.so against libs360_new.so (both
# built beforehand, untracked). Per build and round: the batch-alone kernel table (one context, 22 slots) and the headline.
#   usage (through gpurun, from the repo root): bash tools/ab_libs.sh <tag> [rounds]
cd "$(dirname "$0")/.."
TAG=${1:?tag}; R=${2:-2}
O=gpurun_out/$TAG; mkdir -p $O
cp surround360_amd/libs360.so $O/libs360_keep.so
for r in $(seq 1 $R); do
  for v in ${VARIANTS:-old new}; do   # a variant "<lib>:<ENV=value>" runs libs360_<lib>.so with that environment variable
    lib=${v%%:*}; envs=""; [ "$lib" != "$v" ] && envs=${v#*:}
    cp surround360_amd/libs360_$lib.so surround360_amd/libs360.so
    env $envs timeout 600 python bench.py --steps 12 --warmup 6 --no-extras --no-cpu-baseline > $O/${v//[:=]/_}_$r.json 2> $O/${v//[:=]/_}_$r.err
    python - $O/${v//[:=]/_}_$r.json $v $r <<'P'
import json, sys
d = json.load(open(sys.argv[1]))
k = d["roofline"]["batch_alone_kernel_ms_per_frame"]
print("%s round %s: value %.2f checked %s  batch alone %.3f ms/frame  median %.3f (in flight %.3f)  sweep %.3f" % (
    sys.argv[2], sys.argv[3], d["value"], d["checked"], d["roofline"]["batch_alone_ms_per_frame"], k["flow_median"],
    d["kernel_ms_per_frame_in_flight"]["flow_median"], k["flow_sweep"]))
P
  done
done
cp $O/libs360_keep.so surround360_amd/libs360.so
