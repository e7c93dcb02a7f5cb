#!/bin/bash
# round 3, GPU call D: 3 vs 4 lanes per pixel in the throughput sweep.
cd "$(dirname "$0")/.."
O=gpurun_out/r3d; mkdir -p $O
( cd tools
  for l in 3 4; do
    echo "## lpp=$l side level 607x884, 168 flows x 2 streams"
    for k in 8; do printf "lpp=%s perCU=%-2s " $l $k; S360_QUAD_LPP=$l S360_QUAD_WAVES_PER_CU=$k timeout 100 ./sweep_microbench tp1 607 884 168 2 3; done
    echo "## lpp=$l pole level 5040x1052, 55 % masked, 24 flows x 2"
    for k in 8; do printf "lpp=%s perCU=%-2s " $l $k; S360_MB_MASKROWS=0.55 S360_QUAD_LPP=$l S360_QUAD_WAVES_PER_CU=$k timeout 100 ./sweep_microbench tp1 5040 1052 24 2 3; done
  done
) > $O/microbench.txt 2>&1
S360_QUAD_LPP=3 timeout 600 python bench.py --no-extras --no-cpu-baseline > $O/bench_lpp3.json 2> $O/bench_lpp3.err
S360_QUAD_LPP=4 timeout 600 python bench.py --no-extras --no-cpu-baseline > $O/bench_lpp4.json 2> $O/bench_lpp4.err
S360_QUAD_LPP=3 S360_QUAD_WAVES_PER_CU=6 timeout 600 python bench.py --no-extras --no-cpu-baseline > $O/bench_lpp3_cu6.json 2> $O/bench_lpp3_cu6.err
timeout 600 python -m pytest tests/test_gpu_flow.py tests/test_gpu_zz_variants.py -m gpu -x -q > $O/pytest.log 2>&1
tail -3 $O/pytest.log
