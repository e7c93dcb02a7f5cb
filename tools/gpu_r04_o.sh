#!/bin/bash
# frames in flight: one context x 44 slots, three x 15, two x 24 against the default two x 22
cd "$(dirname "$0")/.."
O=gpurun_out/r04_o; mkdir -p $O
for cfg in "1 44" "3 15" "2 24"; do
  set -- $cfg
  timeout 500 python bench.py --no-extras --no-cpu-baseline --steps 16 --warmup 6 --inflight $1 --slots $2 > $O/bench_$1x$2.json 2> $O/bench_$1x$2.err
  python - "$O/bench_$1x$2.json" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1], d['value'], d['roofline'].get('frac'), d.get('checked'), d.get('hbm_used_GB_in_timed_region'), d.get('errors'))
except Exception as e: print(sys.argv[1], 'failed', e)
PY
  tail -2 $O/bench_$1x$2.err
done
