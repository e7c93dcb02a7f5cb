#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r04_j; mkdir -p $O
timeout 300 python tools/frame_time.py surround360_amd/libs360.so 24 > $O/frame_time.txt 2>&1
timeout 300 python tools/frame_time.py surround360_amd/libs360.so 24 >> $O/frame_time.txt 2>&1
timeout 300 python tools/isp_time.py --no-cpu > $O/isp_time.txt 2>&1
grep -v Warn $O/frame_time.txt; grep -v "Warn\|^W2026" $O/isp_time.txt | tail -4
