#!/bin/bash
# rocprofv3 records of the device PNG encoder (run on a GPU box from the repo root): kernel-trace summary of tools/png_time.py and the
# SQ counter passes (executed instructions per launch; VALU / LDS busy) of k_png_band / _layout / _gather.   usage: bash tools/png_profile.sh <tag>
TAG=${1:?tag}; O=gpurun_out/$TAG; mkdir -p $O
cd "$(dirname "$0")/.."
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
P=/tmp/s360_prof/$TAG; mkdir -p $P
rocprofv3 --kernel-trace --stats -d $P/k1 -o k1 -- python tools/png_time.py > $O/run.log 2>&1
python tools/rocpd_kernel_stats.py $P/k1/k1_results.db "rocprofv3 --kernel-trace --stats summary ($TAG): python tools/png_time.py — one 8K frame rendered 7 times with s360_set_png_encode (latency sweep kernel), the device PNG encoder behind every frame" "k_png_band / k_png_layout / k_png_gather: surround360_amd/csrc/png.hip; $(grep '^{' $O/run.log | tail -1)" | head -16 > $O/${TAG}_png_kernel_stats.txt
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $P/k2 -o k2 -- python tools/png_time.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $P/k3 -o k3 -- python tools/png_time.py > /dev/null 2>&1
{
  echo "# SQ counter passes on python tools/png_time.py ($TAG): VALU / LDS busy, then executed wave-level instructions per launch"
  python tools/valu_busy.py $P/k3/k3_results.db | grep -E "^#|^kernel|k_png"
  python tools/valu_busy.py $P/k2/k2_results.db --json | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d['kernels'].items():
    if 'k_png' in k: print(k[:60], v)"
} > $O/${TAG}_png_pmc.txt
cut -c1-150 $O/${TAG}_png_kernel_stats.txt; cut -c1-220 $O/${TAG}_png_pmc.txt
