#!/bin/bash
# round 4, fifth call: what bounds the end-to-end stream through files — encoder threads / encoder count / page-locked buffers
cd "$(dirname "$0")/.."
O=gpurun_out/r04_e; mkdir -p $O
run() { # name, env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --e2e-only 20 > $O/e2e_$n.json 2>> $O/e2e.err
}
run default
run t16 S360_PNG_THREADS=16
run t64 S360_PNG_THREADS=64
run t0 S360_PNG_THREADS=0
run enc2 S360_ENCODERS=2
run heap S360_HOST_PINNED=0
python - <<'PY'
import json
for n in ("default","t16","t64","t0","enc2","heap"):
    try:
        e=json.load(open('gpurun_out/r04_e/e2e_%s.json'%n))['end_to_end_files']
        print(n,{k:e.get(k) for k in ('ms_per_frame_stream','ms_per_frame_steady','host_thread_ms_per_frame','last_frame_equals_in_process_stream')})
    except Exception as ex: print(n,ex)
PY
