#!/bin/bash
# One GPU call of a round (through gpurun, from the repo root): usage  bash tools/gpu_job.sh <tag> <stage> [<stage> ..]
# Stages (each under its own timeout, outputs under gpurun_out/<tag>/):
#   tests:<k-expr>   pytest -m gpu -k "<k-expr>" over tests/ (":" alone = the whole -m gpu suite)
#   bench[:args]     python bench.py [args] -> bench.json, then the key figures of the line
#   profiles         tools/make_profiles.sh <tag> (kernel stats, isolated stats, FETCH / WRITE, SQ / VALU busy)
#   sh:<command>     anything else
cd "$(dirname "$0")/.."
TAG=${1:?tag}; shift
O=gpurun_out/$TAG; mkdir -p $O
for st in "$@"; do
  case "$st" in
    tests:*) k="${st#tests:}"; n=$(ls $O/pytest*.log 2>/dev/null | wc -l)
      if [ -z "$k" ]; then timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest$n.log 2>&1
      else timeout 1500 python -m pytest tests -m gpu -x -q -k "$k" > $O/pytest$n.log 2>&1; fi
      tail -4 $O/pytest$n.log ;;
    bench*) a="${st#bench}"; a="${a#:}"; n=$(ls $O/bench*.json 2>/dev/null | wc -l)
      timeout 1200 python bench.py $a > $O/bench$n.json 2> $O/bench$n.err
      python tools/bench_summary.py $O/bench$n.json || tail -c 800 $O/bench$n.err ;;
    profiles) bash tools/make_profiles.sh $TAG ${SLOTS:-22} > $O/make_profiles.log 2>&1; cp profiles/${TAG}_* $O/ 2>/dev/null; ls $O ;;
    sh:*) bash -c "${st#sh:}" > $O/sh_$(date +%s%N).log 2>&1; tail -5 $O/sh_*.log | tail -12 ;;
  esac
done
