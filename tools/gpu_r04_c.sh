#!/bin/bash
# round 4, third call: latency sweep kernel of round 3 against this round's on the same box, the full bench line with a
# shorter video stream, the end-to-end leg
cd "$(dirname "$0")/.."
O=gpurun_out/r04_c; mkdir -p $O
{
echo "## r03 lock kernel"; timeout 120 tools/mb_lock_r03
echo "## r04 lock kernel"; timeout 120 tools/mb_lock_r04
echo "## r03 again"; timeout 120 tools/mb_lock_r03
echo "## r04 again"; timeout 120 tools/mb_lock_r04
} > $O/microbench.txt 2>&1
timeout 900 python bench.py --no-cpu-baseline --video-frames 60 > $O/bench.json 2> $O/bench.err
cat $O/microbench.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_c/bench.json'))
print(d['value'], d.get('single_frame',{}).get('ms'), d.get('single_frame',{}).get('sweep'))
v=d.get('video_stream',{}); print({k:v.get(k) for k in ('ms_per_frame','frames_per_s','host_upload_ms_per_frame','unpipelined')})
e=d.get('end_to_end_files',{}); print({k:e.get(k) for k in ('ms_per_frame_stream','ms_per_frame_steady','host_thread_ms_per_frame','last_frame_equals_in_process_stream','files_in_memory')})
print(d.get('errors'))
PY
tail -5 $O/bench.err
