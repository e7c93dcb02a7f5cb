#!/bin/bash
# first hardware run of the accelerated ISP's arithmetic (s360_isp_config.pipe): the kernels against their oracle, both host
# programs, and the reference's own Unpacker / Raw2Rgb --accelerate over the halide shim
cd "$(dirname "$0")/.."
O=gpurun_out/r04_s; mkdir -p $O
timeout 280 python -m pytest tests/test_gpu_isp.py tests/test_gpu_zz_unpacker.py tests/test_gpu_zzz_ref_binding.py tests/test_gpu_host.py -m gpu -q -k "isp or unpacker or raw2rgb or accelerated" > $O/pytest.log 2>&1
tail -15 $O/pytest.log
timeout 60 python tools/isp_time.py > $O/isp_time.txt 2>&1; tail -12 $O/isp_time.txt
