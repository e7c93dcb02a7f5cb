#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3h; mkdir -p $O
timeout 600 python bench.py --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 900 python -m pytest tests/test_gpu_flow.py tests/test_gpu_ops.py tests/test_gpu_frame.py -m gpu -x -q > $O/pytest.log 2>&1
tail -2 $O/pytest.log
