#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3g; mkdir -p $O
timeout 600 python bench.py --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
ISO="python bench.py --inflight 1 --slots 12 --steps 2 --warmup 1 --no-extras --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/s360_prof/f -o f -- $ISO > $O/f.log 2>&1
python tools/rocpd_pmc.py /tmp/s360_prof/f/f_results.db FETCH_SIZE | head -16 > $O/fetch.txt
timeout 600 python -m pytest tests/test_gpu_flow.py tests/test_gpu_ops.py -m gpu -x -q > $O/pytest.log 2>&1
tail -2 $O/pytest.log
