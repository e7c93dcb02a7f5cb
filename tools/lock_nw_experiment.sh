#!/bin/bash
# the latency sweep kernel with 1 / 2 / 4 compute waves per workgroup (bands of 4 / 8 / 16 rows), alternating on
# one box, and the instrumented build's phase ticks for 1 and 2.   usage: bash tools/lock_nw_experiment.sh <tag>
TAG=${1:?tag}; O=gpurun_out/$TAG; mkdir -p $O
cd "$(dirname "$0")/.."
{
for rep in 1 2; do
  for nw in 2 1 4; do
    echo "== S360_LOCK_NW=$nw (pass $rep)"
    S360_LOCK_NW=$nw timeout 300 tools/sweep_microbench
  done
done
for nw in 2 1; do
  for cfg in "5040 1052 4" "613 128 4" "607 884 28"; do
    echo "== instrumented, S360_LOCK_NW=$nw: $cfg"
    S360_LOCK_NW=$nw timeout 120 tools/sweep_microbench_ts ts $cfg
  done
done
} > $O/lock_nw.txt 2>&1
cat $O/lock_nw.txt
