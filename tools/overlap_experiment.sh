#!/bin/bash
# VERDICT r05 item 3 on a GPU box (from the repo root): (1) one context alone, 16 slots, batches back to back with and without frame
# pipelining of the batches (batch k's pole stage on a second stream beside batch k+1's side stage);
# (2) the default 2-context region the same two ways; (3) a kernel trace of the default command and how much of the pole sweeps' time
# already runs beside another context's kernels (tools/sweep_overlap.py).   usage: bash tools/overlap_experiment.sh <tag>
TAG=${1:?tag}; O=gpurun_out/$TAG; mkdir -p $O
cd "$(dirname "$0")/.."
B="python bench.py --steps 12 --warmup 4 --no-extras --no-cpu-baseline"
for inflight in 1 2; do
  $B --inflight $inflight --slots 16 > $O/plain_$inflight.json 2> $O/plain_$inflight.err
  $B --inflight $inflight --slots 16 --pipeline-batches > $O/piped_$inflight.json 2> $O/piped_$inflight.err
done
python - $O <<'PY'
import json, sys
o = sys.argv[1]
for name in ("plain_1", "piped_1", "plain_2", "piped_2"):
    try:
        d = json.loads([ln for ln in open("%s/%s.json" % (o, name)) if ln.startswith("{")][-1])
        print("%-8s %6.2f frames/s  %7.2f ms per frame  checked %s  hbm %s GB  errors %s" % (
            name, d["value"], d["ms_per_step"] / d["config"]["slots_per_context"], d["checked"], d["hbm_used_GB_in_timed_region"], d.get("errors")))
    except Exception as e:
        print(name, "FAILED", e, open("%s/%s.err" % (o, name)).read()[-300:])
PY
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
P=/tmp/s360_prof/$TAG; mkdir -p $P
rocprofv3 --kernel-trace -d $P/kt -o kt -- python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline > $O/kt.log 2>&1
{
  echo "# rocprofv3 --kernel-trace of: python bench.py --steps 6 --warmup 2 --no-extras --no-cpu-baseline (2 contexts x 22 slots)"
  python tools/sweep_overlap.py $P/kt/kt_results.db "k_sweep_quad<true, 4>" 0.6
  echo
  python tools/sweep_overlap.py $P/kt/kt_results.db "k_sweep_quad<true, 3>" 0.6
} > $O/sweep_overlap.txt 2>&1
cat $O/sweep_overlap.txt
