#!/bin/bash
# round 3, GPU call C: sweep with wait-free hot step, coalesced IIR column passes; full GPU suite.
cd "$(dirname "$0")/.."
O=gpurun_out/r3c; mkdir -p $O
( cd tools
  echo "## side level 607x884, 168 flows x 2 streams"
  for k in 6 8 11; do printf "perCU=%-2s " $k; S360_QUAD_WAVES_PER_CU=$k timeout 100 ./sweep_microbench tp1 607 884 168 2 3; done
  echo "## pole level 5040x1052, 55 % masked, 24 flows x 2"
  export S360_MB_MASKROWS=0.55
  for k in 6 8 11; do printf "perCU=%-2s " $k; S360_QUAD_WAVES_PER_CU=$k timeout 100 ./sweep_microbench tp1 5040 1052 24 2 3; done
) > $O/microbench.txt 2>&1
timeout 900 python bench.py --video-frames 40 > $O/bench.json 2> $O/bench.err
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
tail -3 $O/pytest.log
