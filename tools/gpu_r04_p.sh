#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r04_p; mkdir -p $O
timeout 600 python bench.py --no-cpu-baseline --video-frames 40 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_p/bench.json'))
print(d['value'], d['checked'], d.get('errors'), d['data'][:60], d['single_frame']['ms'], d['video_stream']['ms_per_frame'], d['end_to_end_files'].get('ms_per_frame_steady'), d['end_to_end_files'].get('last_frame_equals_in_process_stream'))
PY
tail -3 $O/bench.err
