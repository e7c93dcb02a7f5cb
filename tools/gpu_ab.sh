#!/bin/bash
# One short GPU call between two kernel changes: sweep micro-benchmarks of the builds to compare (tools/mb_*: built
# beforehand, each from its own sources), the quick bench line, the flow / operator / frame tests.
#   usage: bash tools/gpu_ab.sh <tag>        (the closing call of a round is tools/gpu_round.sh)
cd "$(dirname "$0")/.."
O=gpurun_out/${1:?tag}; mkdir -p $O
{
  for b in tools/mb_*; do
    [ -x $b ] && [ "${b%.sh}" = "$b" ] || continue
    echo "## $b"
    case $b in
      *lds_dma_align) timeout 60 $b ;;
      *)
        printf "side 607x884 B=168 x2 streams : "; timeout 100 $b tp1 607 884 168 2 3
        printf "pole 5040x1052 B=48 x1 (mask .55): "; S360_MB_MASKROWS=0.55 timeout 100 $b tp1 5040 1052 48 1 3 ;;
    esac
  done
} > $O/microbench.txt 2>&1
timeout 600 python bench.py --no-extras --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 900 python -m pytest tests/test_gpu_flow.py tests/test_gpu_ops.py tests/test_gpu_frame.py -m gpu -x -q > $O/pytest.log 2>&1
grep -E "passed|failed|error" $O/pytest.log | tail -2
cat $O/microbench.txt
