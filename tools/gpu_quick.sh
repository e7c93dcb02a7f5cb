#!/bin/bash
# A short GPU call between two kernel changes: the bench line without its extra legs + the operator / flow / frame tests.
#   usage: bash tools/gpu_quick.sh <tag>
cd "$(dirname "$0")/.."
O=gpurun_out/${1:?tag}; mkdir -p $O
timeout 600 python bench.py --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 900 python -m pytest tests/test_gpu_flow.py tests/test_gpu_ops.py tests/test_gpu_frame.py -m gpu -x -q > $O/pytest.log 2>&1
grep -E "passed|failed|error" $O/pytest.log | tail -2
