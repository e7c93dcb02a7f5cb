#!/bin/bash
# round 4, ninth call: half-records (15x15 blur writes 8 bytes, the sweeps complete the record from I0's gradient plane), ISP
# column pass; parity at 8K and timing
cd "$(dirname "$0")/.."
O=gpurun_out/r04_i; mkdir -p $O
{
for b in tools/mb_lock_r04 tools/mb_quad_halfrec; do
  echo "## $b"
  printf "side 607x884 B=168 x2 streams : "; timeout 100 $b tp1 607 884 168 2 3
  printf "pole 5040x1052 B=48 x1 (mask .55): "; S360_MB_MASKROWS=0.55 timeout 100 $b tp1 5040 1052 48 1 3
done
} > $O/microbench.txt 2>&1
timeout 300 python tools/frame_time.py surround360_amd/libs360.so 24 > $O/frame_time.txt 2>&1
timeout 600 python bench.py --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 300 python tools/isp_time.py --no-cpu > $O/isp_time.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_flow.py tests/test_gpu_zz_variants.py tests/test_gpu_frame.py tests/test_gpu_isp.py -m gpu -x -q > $O/pytest.log 2>&1
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q >> $O/pytest.log 2>&1
grep -E "passed|failed|error" $O/pytest.log | tail -3
cat $O/microbench.txt; grep -v Warn $O/frame_time.txt; grep -v "Warn\|^W2026" $O/isp_time.txt | tail -5
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_i/bench.json'))
print(d['value'], d['roofline'].get('avg_launch_ms'), d['roofline'].get('frac'), d['roofline'].get('batch_alone_kernel_ms_per_frame'))
PY
