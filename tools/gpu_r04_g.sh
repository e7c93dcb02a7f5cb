#!/bin/bash
# round 4, seventh call: ISP kernel trace (where the 0.8 ms per image goes), the box's CPU limits, e2e variations
cd "$(dirname "$0")/.."
O=gpurun_out/r04_g; mkdir -p $O
{ echo "nproc $(nproc)"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null; grep -c processor /proc/cpuinfo; uptime; } > $O/cpu.txt 2>&1
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
P=/tmp/s360_prof_isp; mkdir -p $P
rocprofv3 --kernel-trace --stats -d $P/ks -o ks -- python tools/isp_time.py --no-cpu > $O/isp_time.txt 2>&1
python tools/rocpd_kernel_stats.py $P/ks/ks_results.db "rocprofv3 --kernel-trace --stats: python tools/isp_time.py --no-cpu" "ISP + 8K frames from raw" > $O/isp_kernel_stats.txt 2>&1
run() { n=$1; shift; env "$@" timeout 300 python bench.py --e2e-only 20 > $O/e2e_$n.json 2>> $O/e2e.err; }
run default
run t8 S360_PNG_THREADS=8
run enc2t16 S360_ENCODERS=2 S360_PNG_THREADS=16
cat $O/cpu.txt; grep -v Warn $O/isp_time.txt | tail -6; head -40 $O/isp_kernel_stats.txt
python - <<'PY'
import json
for n in ("default","t8","enc2t16"):
    try:
        e=json.load(open('gpurun_out/r04_g/e2e_%s.json'%n))['end_to_end_files']
        print(n,{k:e.get(k) for k in ('ms_per_frame_stream','ms_per_frame_steady','host_thread_ms_per_frame')})
    except Exception as ex: print(n,ex)
PY
