#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r04_k; mkdir -p $O
timeout 300 python tools/frame_time.py surround360_amd/libs360.so 24 > $O/frame_time.txt 2>&1
timeout 600 python bench.py --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 900 python -m pytest tests/test_gpu_flow.py tests/test_gpu_zz_variants.py tests/test_gpu_ops.py tests/test_gpu_frame.py -m gpu -x -q > $O/pytest.log 2>&1
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "config2 or config3 or preset" >> $O/pytest.log 2>&1
grep -E "passed|failed|error" $O/pytest.log | tail -3
grep -v Warn $O/frame_time.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_k/bench.json'))
print(d['value'], d['roofline'].get('avg_launch_ms'), d['roofline'].get('frac'), d['roofline'].get('batch_alone_kernel_ms_per_frame'))
PY
