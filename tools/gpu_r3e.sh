#!/bin/bash
# round 3, GPU call E: per-launch lanes per pixel + 12-byte IIR intermediates; one frame alone with either sweep kernel.
cd "$(dirname "$0")/.."
O=gpurun_out/r3e; mkdir -p $O
timeout 600 python bench.py --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py --slots 1 --inflight 1 --steps 4 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_1x1.json 2> $O/bench_1x1.err
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_isp.py tests/test_gpu_flow.py tests/test_gpu_frame.py -m gpu -x -q > $O/pytest.log 2>&1
tail -3 $O/pytest.log
