#!/usr/bin/env python
"""Writes tests/golden/isp_pipe_golden.npz: the outputs of the reference's accelerated ISP — its Halide generator
camera_isp/CameraIspGen.cpp EXECUTED over oracle/ref_shim/halide_eval under its own CameraIspPipe.h
(oracle/_ref/libref_isppipe.so; oracle/ref_ispgen.cpp, ref_isppipe.cpp) — for the cases of tests/test_cpu_isp.py::PIPE_CASES.
Run in the build container, where /root/reference exists:  python tests/golden/make_isp_pipe_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import isputil  # noqa: E402
import oracle_lib as O  # noqa: E402
import test_cpu_isp as T  # noqa: E402

assert O.ref_isp_pipe_lib() is not None, "needs /root/reference (make -C oracle ref)"
out = {}
for case in T.PIPE_CASES:
    name, w, h, bpp, fast, tone, off, unp = case
    out[T._pipe_id(case)] = O.ref_isp_pipe_run(isputil.CONFIGS[name], T._pipe_raw(case), bpp, bool(fast), tone, off, unpacker=bool(unp))
np.savez_compressed(os.path.join(HERE, "isp_pipe_golden.npz"), **out)
print("wrote %d cases, %d bytes" % (len(out), os.path.getsize(os.path.join(HERE, "isp_pipe_golden.npz"))))
