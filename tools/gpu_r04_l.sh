#!/bin/bash
# frames in flight: 2 x 12 (default) against 2 x 13, 2 x 14, 3 x 8
cd "$(dirname "$0")/.."
O=gpurun_out/r04_l; mkdir -p $O
for cfg in "2 13" "2 14" "3 8"; do
  set -- $cfg
  timeout 400 python bench.py --no-extras --no-cpu-baseline --inflight $1 --slots $2 > $O/bench_$1x$2.json 2> $O/bench_$1x$2.err
  python - "$O/bench_$1x$2.json" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1], d['value'], d['roofline'].get('frac'), d.get('checked'), d.get('errors'))
except Exception as e: print(sys.argv[1], 'failed', e)
PY
  tail -2 $O/bench_$1x$2.err
done
rocm-smi --showmeminfo vram 2>/dev/null | head -5
