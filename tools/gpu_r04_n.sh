#!/bin/bash
# flow engines sharing one buffer set, sharpen scratch per group: parity (slots, pipelined streams, sharded ops), then 2 x 20 / 2 x 22 slots
cd "$(dirname "$0")/.."
O=gpurun_out/r04_n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_frame.py tests/test_gpu_host.py -m gpu -x -q > $O/pytest.log 2>&1
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "config3 or config5 or flags or 8k" >> $O/pytest.log 2>&1
grep -E "passed|failed|error" $O/pytest.log | tail -3
for cfg in "2 20" "2 22"; do
  set -- $cfg
  timeout 500 python bench.py --no-extras --no-cpu-baseline --inflight $1 --slots $2 > $O/bench_$1x$2.json 2> $O/bench_$1x$2.err
  python - "$O/bench_$1x$2.json" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1], d['value'], d['roofline'].get('frac'), d.get('checked'), d.get('hbm_used_GB_in_timed_region'), d.get('errors'))
except Exception as e: print(sys.argv[1], 'failed', e)
PY
  tail -2 $O/bench_$1x$2.err
done
