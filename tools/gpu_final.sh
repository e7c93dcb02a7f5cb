#!/bin/bash
# A short closing call when few GPU minutes are left: the operator / flow / frame tests, the full default bench line, then
# (time permitting) the host-program tests.   usage: bash tools/gpu_final.sh <tag>
cd "$(dirname "$0")/.."
O=gpurun_out/${1:?tag}; mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_flow.py tests/test_gpu_frame.py -m gpu -x -q > $O/pytest.log 2>&1
grep -E "passed|failed|error" $O/pytest.log | tail -2
timeout 420 python bench.py > $O/bench.json 2> $O/bench.err
python - <<'P' $O/bench.json
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("bench", d["value"], d["roofline"]["frac"], d.get("end_to_end_files"), d.get("errors"))
P
timeout 150 python -m pytest tests/test_gpu_zz_refprogram.py tests/test_gpu_host.py -m gpu -x -q > $O/pytest_host.log 2>&1
grep -E "passed|failed|error" $O/pytest_host.log | tail -2
