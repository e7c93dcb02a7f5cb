#!/bin/bash
# first hardware run of host/TestRenderStereoPanorama --bin_list (capture containers -> ISP -> frame on the device)
cd "$(dirname "$0")/.."
O=gpurun_out/r04_t; mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_zz_unpacker.py tests/test_gpu_isp.py -m gpu -q > $O/pytest.log 2>&1
tail -6 $O/pytest.log
