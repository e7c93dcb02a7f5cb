#!/bin/bash
# One GPU call between two kernel changes: sweep micro-benchmarks (tools/mb_*: built beforehand), the quick bench line under
# the settings to compare, the flow / operator / frame tests.   usage: bash tools/gpu_ab.sh <tag>
cd "$(dirname "$0")/.."
O=gpurun_out/${1:?tag}; mkdir -p $O
{
  for l in 2 3 4; do
    echo "## tools/mb_new, S360_QUAD_LPP=$l"
    printf "side 607x884 B=168 x2 streams : "; S360_QUAD_LPP=$l timeout 100 tools/mb_new tp1 607 884 168 2 3
    printf "side 607x884 B=336 x1 stream  : "; S360_QUAD_LPP=$l timeout 100 tools/mb_new tp1 607 884 336 1 3
    printf "side 304x442 B=336 x1 stream  : "; S360_QUAD_LPP=$l timeout 100 tools/mb_new tp1 304 442 336 1 3
    printf "pole 5040x1052 B=48 x1 (mask .55): "; S360_QUAD_LPP=$l S360_MB_MASKROWS=0.55 timeout 100 tools/mb_new tp1 5040 1052 48 1 3
  done
} > $O/microbench.txt 2>&1
for l in 2 3; do
  S360_QUAD_LPP_SAT=$l timeout 600 python bench.py --no-extras --no-cpu-baseline > $O/bench_sat$l.json 2> $O/bench_sat$l.err
done
S360_QUAD_LPP_SAT=2 timeout 900 python -m pytest tests/test_gpu_flow.py tests/test_gpu_ops.py tests/test_gpu_frame.py -m gpu -x -q > $O/pytest.log 2>&1
grep -E "passed|failed|error" $O/pytest.log | tail -2
cat $O/microbench.txt
