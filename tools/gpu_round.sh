#!/bin/bash
# What a round's closing GPU call runs (through gpurun, from the repo root): the whole -m gpu suite, the default bench
# line, and the profile set of tools/make_profiles.sh; everything that should come back is copied under gpurun_out/<tag>.
#   usage: bash tools/gpu_round.sh <tag>          e.g.  gpurun --timeout 2400 -- 'bash tools/gpu_round.sh r03_v7'
cd "$(dirname "$0")/.."
TAG=${1:?tag}
O=gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
grep -E "passed|failed|error" $O/pytest.log | tail -2
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
bash tools/make_profiles.sh $TAG ${SLOTS:-22} > $O/make_profiles.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
ISO="python bench.py --inflight 1 --slots ${SLOTS:-22} --steps 2 --warmup 1 --no-extras --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d /tmp/s360_prof/sq -o sq -- $ISO > $O/sq.log 2>&1
{
  echo "# rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT on: $ISO"
  echo "# (one pass, SQ block only; quad-cycles; per kernel: launches, total, per launch)"
  for c in SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT; do
    python tools/rocpd_pmc.py /tmp/s360_prof/sq/sq_results.db $c | head -22; echo
  done
} > profiles/${TAG}_pmc_sq.txt 2>> $O/sq.log
python tools/valu_busy.py /tmp/s360_prof/sq/sq_results.db > profiles/${TAG}_valu_busy.txt 2>> $O/sq.log
cp profiles/${TAG}_* profiles/sweep_traffic.json $O/ 2>/dev/null
ls $O
