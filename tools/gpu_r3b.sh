#!/bin/bash
# round 3, GPU call B: window request ahead on/off, packed pole warp on/off, batched sharpen; one full bench.
cd "$(dirname "$0")/.."
O=gpurun_out/r3b; mkdir -p $O
( cd tools
  echo "## side level 607x884, 168 flows x 2 streams"
  for a in 1 0; do for k in 8 11; do printf "ahead=%s perCU=%-2s " $a $k; S360_QUAD_AHEAD=$a S360_QUAD_WAVES_PER_CU=$k timeout 100 ./sweep_microbench tp1 607 884 168 2 3; done; done
  echo "## pole level 5040x1052, 55 % masked, 24 flows x 2"
  export S360_MB_MASKROWS=0.55
  for a in 1 0; do for k in 8 11; do printf "ahead=%s perCU=%-2s " $a $k; S360_QUAD_AHEAD=$a S360_QUAD_WAVES_PER_CU=$k timeout 100 ./sweep_microbench tp1 5040 1052 24 2 3; done; done
) > $O/microbench.txt 2>&1
timeout 900 python bench.py --video-frames 40 > $O/bench.json 2> $O/bench.err
S360_QUAD_AHEAD=0 S360_POLE_WARP_PACKED=0 timeout 600 python bench.py --no-extras --no-cpu-baseline > $O/bench_alt.json 2> $O/bench_alt.err
timeout 900 python -m pytest tests -m gpu -x -q -k "not fullsize" > $O/pytest.log 2>&1
tail -3 $O/pytest.log
