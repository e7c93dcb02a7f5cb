#!/bin/bash
# round 4, fourth call: round 3's latency sweep kernel against this round's on whole 8K frames (same box), the host-side costs
# (page-locked against heap decode), the end-to-end leg both ways
cd "$(dirname "$0")/.."
O=gpurun_out/r04_d; mkdir -p $O
{
timeout 300 python tools/frame_time.py tools/libs360_r03lock.so 24
timeout 300 python tools/frame_time.py surround360_amd/libs360.so 24
timeout 300 python tools/frame_time.py tools/libs360_r03lock.so 24
timeout 300 python tools/frame_time.py surround360_amd/libs360.so 24
} > $O/frame_time.txt 2>&1
timeout 120 tools/host_io_time /tmp surround360_amd/libs360.so > $O/host_io_time.txt 2>&1
timeout 300 python bench.py --e2e-only 20 > $O/e2e_pinned.json 2> $O/e2e.err
S360_HOST_PINNED=0 timeout 300 python bench.py --e2e-only 20 > $O/e2e_heap.json 2>> $O/e2e.err
cat $O/frame_time.txt | grep -v Warning; cat $O/host_io_time.txt
python - <<'PY'
import json
for n in ("pinned","heap"):
    try:
        e=json.load(open('gpurun_out/r04_d/e2e_%s.json'%n))['end_to_end_files']
        print(n,{k:e.get(k) for k in ('ms_per_frame_stream','ms_per_frame_steady','host_thread_ms_per_frame','last_frame_equals_in_process_stream')})
    except Exception as ex: print(n,ex)
PY
