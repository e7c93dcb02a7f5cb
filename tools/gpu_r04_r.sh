#!/bin/bash
# (the k_sweep_quad_occ3 build was removed after this measurement: DESIGN.md section 5, profiles/r04_v5_*)
# How much does a wave more per SIMD buy the throughput sweep? One build (k_sweep_quad_occ3: its spill code is in every arm) at
# 4 / 8 / 12 persistent waves per CU = 1 / 2 / 3 per SIMD, on a launch with enough flows to keep twelve per CU busy (224 flows x
# up to 23 bands in flight each), and the default build at 4 / 8.
cd "$(dirname "$0")/.."
O=gpurun_out/r04_r; mkdir -p $O
{
echo "# side level 607x884, 224 flows, one stream (Gpx/s)"
for n in 4 8 12; do echo -n "occ3 build, $n waves per CU   : "; S360_QUAD_OCC3=1 S360_QUAD_WAVES_PER_CU=$n timeout 60 tools/sweep_microbench tp1 607 884 224 1 3; done
for n in 4 8; do echo -n "default LPP4, $n waves per CU : "; S360_QUAD_LPP=4 S360_QUAD_WAVES_PER_CU=$n timeout 60 tools/sweep_microbench tp1 607 884 224 1 3; done
for n in 4 8; do echo -n "default LPP3, $n waves per CU : "; S360_QUAD_LPP=3 S360_QUAD_WAVES_PER_CU=$n timeout 60 tools/sweep_microbench tp1 607 884 224 1 3; done
} > $O/microbench.txt 2>&1
cat $O/microbench.txt
