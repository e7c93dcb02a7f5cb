#!/bin/bash
# round 3, GPU call A: microbench of the quad sweep (LDS window on/off), the whole -m gpu suite, the bench.
cd "$(dirname "$0")/.."
O=gpurun_out/r3a; mkdir -p $O
free -g | head -2 > $O/host.txt; nproc >> $O/host.txt
( cd tools
  echo "## side level 607x884, 168 flows x 2 streams"
  for win in 1 0; do for k in 8 10 11 12; do printf "win=%s perCU=%-2s " $win $k; S360_QUAD_WIN=$win S360_QUAD_WAVES_PER_CU=$k timeout 100 ./sweep_microbench tp1 607 884 168 2 3; done; done
  echo "## pole level 5040x1052, 55 % masked, 24 flows x 2"
  export S360_MB_MASKROWS=0.55
  for win in 1 0; do for k in 10 11; do printf "win=%s perCU=%-2s " $win $k; S360_QUAD_WIN=$win S360_QUAD_WAVES_PER_CU=$k timeout 100 ./sweep_microbench tp1 5040 1052 24 2 3; done; done
) > $O/microbench.txt 2>&1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
S360_QUAD_WIN=0 timeout 600 python bench.py --no-extras --no-cpu-baseline > $O/bench_win0.json 2> $O/bench_win0.err
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
tail -5 $O/pytest.log
