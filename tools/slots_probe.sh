#!/bin/bash
# Frames/s against frames in flight (DESIGN.md section 6's table), on a GPU box from the repo root: the headline's independent frames
# at 2 x {8, 12, 16, 22} slots and temporally chained streams at 2 x {6, 10, 15} slots.   usage: bash tools/slots_probe.sh <tag>
TAG=${1:?tag}; O=gpurun_out/$TAG; mkdir -p $O
for s in 8 12 16; do
  python bench.py --slots $s --steps 16 --warmup 8 --no-extras --no-cpu-baseline 2>/dev/null | grep '^{"metric"' > $O/independent_$s.json
  python - $O/independent_$s.json $s <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print("independent 2 x %s slots: %.2f frames/s, %s GB, checked %s" % (sys.argv[2], d["value"], d["hbm_used_GB_in_timed_region"], d["checked"]))
PY
done
for s in 6 10; do
  python bench.py --streams-only --stream-slots $s 2>/dev/null | grep '^{"video' > $O/chained_$s.json
  python - $O/chained_$s.json $s <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))["video_streams_batched"]; print("chained 2 x %s slots: %.2f frames/s, %s GB, checked %s" % (sys.argv[2], d["frames_per_s"], d["hbm_used_GB"], d["checked"]))
PY
done
