#!/bin/bash
# closing check of the round: the default bench line with every leg but the reference program's 31 s CPU run, a short stream
cd "$(dirname "$0")/.."
O=gpurun_out/r04_w; mkdir -p $O
timeout 135 python bench.py --no-cpu-baseline --video-frames 24 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/r04_w/bench.json').read().strip().splitlines()[-1])
    print(d['value'], d['roofline']['frac'], d['checked'], d.get('errors'), d['data'][:52], d['single_frame']['ms'], d['video_stream']['ms_per_frame'],
          d['end_to_end_files'].get('ms_per_frame_steady'), d['end_to_end_files'].get('last_frame_equals_in_process_stream'), d.get('isp'))
except Exception as e:
    print('failed', e)
PY
tail -3 $O/bench.err
