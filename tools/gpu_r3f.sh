#!/bin/bash
# round 3, GPU call F: the profile set of the round (kernel traces, FETCH / WRITE traffic) + SQ counters of one batched context.
cd "$(dirname "$0")/.."
O=gpurun_out/r3f; mkdir -p $O
bash tools/make_profiles.sh r03_v5 12 > $O/make_profiles.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
ISO="python bench.py --inflight 1 --slots 12 --steps 2 --warmup 1 --no-extras --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d /tmp/s360_prof/sq -o sq -- $ISO > $O/sq.log 2>&1
{
  echo "# rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT on: $ISO"
  echo "# (one pass, SQ block only; quad-cycles; per kernel: launches, total, per launch)"
  for c in SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT; do
    python tools/rocpd_pmc.py /tmp/s360_prof/sq/sq_results.db $c | head -22; echo
  done
} > profiles/r03_v5_pmc_sq.txt 2>> $O/sq.log
cp profiles/r03_v5_* profiles/sweep_traffic.json $O/ 2>/dev/null
ls -la profiles | tail -12
