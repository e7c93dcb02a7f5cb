#!/bin/bash
# round 4, sixth call: novel view (integer-row flow sample, 64x4 XCD-ordered tiles) timing + parity, the end-to-end stream with
# the Sub + Z_RLE PNG writer
cd "$(dirname "$0")/.."
O=gpurun_out/r04_f; mkdir -p $O
timeout 300 python tools/frame_time.py surround360_amd/libs360.so 24 > $O/frame_time.txt 2>&1
timeout 300 python bench.py --e2e-only 20 > $O/e2e.json 2> $O/e2e.err
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_frame.py tests/test_gpu_host.py -m gpu -x -q > $O/pytest.log 2>&1
timeout 600 python -m pytest tests/test_gpu_fullsize.py -k "config3" -x -q >> $O/pytest.log 2>&1
grep -E "passed|failed|error" $O/pytest.log | tail -3
grep -v Warn $O/frame_time.txt
python - <<'PY'
import json
e=json.load(open('gpurun_out/r04_f/e2e.json'))['end_to_end_files']
print({k:e.get(k) for k in ('ms_per_frame_stream','ms_per_frame_steady','host_thread_ms_per_frame','last_frame_equals_in_process_stream','output_png_bytes_per_frame')})
PY
