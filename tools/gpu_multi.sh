#!/bin/bash
# Hardware-day checklist for an N-GPU node (no curve is measured): tools/multi_gpu_check.py, each step in a
# process of its own under a timeout, the first failing step named.   usage: bash tools/gpu_multi.sh [N]
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c 'import __graft_entry__ as g; g.build()' || { echo "FAILED at step 0: build"; exit 1; }
exec python tools/multi_gpu_check.py "$@"
