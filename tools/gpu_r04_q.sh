#!/bin/bash
# (S360_QUAD_OCC3 selected a second build of the kernel that was removed after this measurement: DESIGN.md section 5, profiles/r04_v5_*)
# The throughput sweep kernel held to three waves per SIMD (k_sweep_quad_occ3, S360_QUAD_OCC3) against the default, same box:
# micro-benchmark of a saturating side level and of a pole level, then the bench line (timed region + check) with and without.
cd "$(dirname "$0")/.."
O=gpurun_out/r04_q; mkdir -p $O
{
echo "# side level 607x884, 112 flows, one stream (Gpx/s)"
echo -n "LPP3 two waves (default)  : "; S360_QUAD_LPP=3 timeout 100 tools/sweep_microbench tp1 607 884 112 1 3
echo -n "LPP4 two waves            : "; S360_QUAD_LPP=4 timeout 100 tools/sweep_microbench tp1 607 884 112 1 3
echo -n "LPP4 three waves (occ3)   : "; S360_QUAD_OCC3=1 timeout 100 tools/sweep_microbench tp1 607 884 112 1 3
echo -n "LPP4 occ3, 8 waves per CU : "; S360_QUAD_OCC3=1 S360_QUAD_WAVES_PER_CU=8 timeout 100 tools/sweep_microbench tp1 607 884 112 1 3
echo "# pole level 5040x1052, 24 flows, 55 % of the rows masked"
echo -n "LPP4 two waves (default)  : "; S360_MB_MASKROWS=0.55 timeout 100 tools/sweep_microbench tp1 5040 1052 24 1 3
echo -n "LPP4 three waves (occ3=2) : "; S360_MB_MASKROWS=0.55 S360_QUAD_OCC3=2 timeout 100 tools/sweep_microbench tp1 5040 1052 24 1 3
} > $O/microbench.txt 2>&1
cat $O/microbench.txt
for v in 1 0; do
  S360_QUAD_OCC3=$v timeout 170 python bench.py --no-extras --no-cpu-baseline --steps 12 --warmup 4 > $O/bench_occ3_$v.json 2> $O/bench_occ3_$v.err
  python - "$O/bench_occ3_$v.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print(sys.argv[1], 'frames/s', round(d['value'],2), 'frac', round(r.get('frac',0),4), 'avg_launch_ms', r.get('avg_launch_ms'), 'checked', d.get('checked'), d.get('mismatching_frames_all_ranks'), 'sweep ms/frame', r.get('batch_alone_kernel_ms_per_frame',{}).get('flow_sweep'), d['data'][:48])
except Exception as e: print(sys.argv[1], 'failed', e)
PY
  tail -2 $O/bench_occ3_$v.err
done
