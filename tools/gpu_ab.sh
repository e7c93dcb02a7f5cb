#!/bin/bash
# One GPU call between two kernel changes: sweep micro-benchmarks of several builds (tools/mb_*: built beforehand from the
# sources to compare), short bench lines under both latency kernels, the flow / operator / frame tests.
#   usage: bash tools/gpu_ab.sh <tag>
cd "$(dirname "$0")/.."
O=gpurun_out/${1:?tag}; mkdir -p $O
{
  for b in tools/mb_lat*; do
    [ -x $b ] || continue
    echo "## $b (latency mapping: checks of the band above every N steps, publishes every P: mb_latNP)"
    timeout 300 $b
  done
} > $O/microbench.txt 2>&1
for m in lock wave; do
  S360_LATENCY_SWEEP=$m timeout 900 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --video-frames 40 > $O/bench_$m.json 2> $O/bench_$m.err
done
S360_LATENCY_SWEEP=wave timeout 900 python -m pytest tests/test_gpu_flow.py tests/test_gpu_ops.py tests/test_gpu_frame.py -m gpu -x -q > $O/pytest.log 2>&1
grep -E "passed|failed|error" $O/pytest.log | tail -2
cat $O/microbench.txt
