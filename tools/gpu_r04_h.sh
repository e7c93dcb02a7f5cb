#!/bin/bash
# round 4, eighth call: tiled ISP IIR passes (trace + parity), e2e with cgroup-aware encoder threads
cd "$(dirname "$0")/.."
O=gpurun_out/r04_h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
P=/tmp/s360_prof_isp; mkdir -p $P
rocprofv3 --kernel-trace --stats -d $P/ks -o ks -- python tools/isp_time.py --no-cpu > $O/isp_time_prof.txt 2>&1
python tools/rocpd_kernel_stats.py $P/ks/ks_results.db "rocprofv3 --kernel-trace --stats: python tools/isp_time.py --no-cpu" "ISP + 8K frames from raw" > $O/isp_kernel_stats.txt 2>&1
timeout 300 python tools/isp_time.py > $O/isp_time.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_isp.py tests/test_gpu_zz_unpacker.py tests/test_gpu_host.py -m gpu -x -q > $O/pytest.log 2>&1
timeout 300 python bench.py --e2e-only 20 > $O/e2e.json 2> $O/e2e.err
grep -E "passed|failed|error" $O/pytest.log | tail -3
grep -v "Warn\|^W2026" $O/isp_time.txt | tail -8; grep "k_isp" $O/isp_kernel_stats.txt
python - <<'PY'
import json
e=json.load(open('gpurun_out/r04_h/e2e.json'))['end_to_end_files']
print({k:e.get(k) for k in ('ms_per_frame_stream','ms_per_frame_steady','host_thread_ms_per_frame','last_frame_equals_in_process_stream')})
PY
