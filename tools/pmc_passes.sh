#!/bin/bash
# The two SQ counter passes behind profiles/valu_busy.json (run on a GPU box from the repo root): one bench.py context alone with
# 22 frame slots (launches do not overlap), (1) the SQ activity counters -> <tag>_valu_busy.txt, (2) SQ_INSTS_* -> executed
# wave-level instructions per launch; then tools/valu_families.py folds both into profiles/valu_busy.json.
#   usage: bash tools/pmc_passes.sh <tag> [slots]
TAG=${1:?tag}; SLOTS=${2:-22}
cd "$(dirname "$0")/.."
O=gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
P=/tmp/s360_prof/$TAG; mkdir -p $P
ISO="python bench.py --inflight 1 --slots $SLOTS --steps 2 --warmup 1 --no-extras --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $P/sq -o sq -- $ISO > $O/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $P/in -o in -- $ISO > $O/in.log 2>&1
python tools/valu_busy.py $P/sq/sq_results.db > profiles/${TAG}_valu_busy.txt 2>> $O/sq.log
python tools/valu_busy.py $P/sq/sq_results.db --json > $O/busy.json 2>> $O/sq.log
python tools/valu_busy.py $P/in/in_results.db --json > $O/insts.json 2>> $O/in.log
python tools/valu_busy.py $P/in/in_results.db > profiles/${TAG}_valu_insts.txt 2>> $O/in.log
python tools/valu_families.py $TAG $O/busy.json $O/insts.json > $O/valu_busy.json && cp $O/valu_busy.json profiles/valu_busy.json
cp profiles/${TAG}_valu_busy.txt profiles/${TAG}_valu_insts.txt $O/ 2>/dev/null
head -30 profiles/${TAG}_valu_busy.txt
