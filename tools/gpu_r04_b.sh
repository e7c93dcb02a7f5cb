#!/bin/bash
# round 4, second call: the lock kernel's LDS window (ds reads, taps in flight across the barrier) A/B on the microbench,
# the quick bench line, the end-to-end stream through files, flow / variant / frame parity
cd "$(dirname "$0")/.."
O=gpurun_out/r04_b; mkdir -p $O
{
echo "## window on"; timeout 120 tools/sweep_microbench
echo "## window off (S360_SWEEP_DBG=256)"; S360_SWEEP_DBG=256 timeout 120 tools/sweep_microbench
echo "## ts window on"; timeout 60 tools/sweep_microbench_ts ts 5040 1052 4
} > $O/microbench.txt 2>&1
timeout 600 python bench.py --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 400 python bench.py --e2e-only 20 > $O/e2e.json 2> $O/e2e.err
timeout 900 python -m pytest tests/test_gpu_flow.py tests/test_gpu_zz_variants.py tests/test_gpu_frame.py tests/test_gpu_host.py -m gpu -x -q > $O/pytest.log 2>&1
timeout 600 python -m pytest tests/test_gpu_fullsize.py -k "config3 or config5" -x -q >> $O/pytest.log 2>&1
grep -E "passed|failed|error" $O/pytest.log | tail -3
cat $O/microbench.txt
tail -c 1500 $O/e2e.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_b/bench.json'))
print(d['value'], d.get('single_frame',{}).get('ms'), d.get('single_frame',{}).get('sweep'))
PY
