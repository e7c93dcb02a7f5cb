#!/usr/bin/env python
"""Writes tests/golden/isp_golden.npz: the outputs of the REFERENCE soft ISP (oracle/_ref/libref_isp.so = the
reference's CameraIsp.h compiled from /root/reference, see oracle/ref_isp.cpp) for the cases of tests/test_cpu_isp.py.
Run in the build container, where /root/reference exists:  python tests/golden/make_isp_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import isputil  # noqa: E402
import oracle_lib as O  # noqa: E402
import test_cpu_isp as T  # noqa: E402

assert O.ref_isp_lib() is not None, "needs /root/reference (make -C oracle ref)"
out = {}
for case in T.CASES:
    name, w, h, bpp, dm, rs, tone, off = case
    out[T._case_id(case)] = O.ref_isp_run(isputil.CONFIGS[name], T._raw(case), bpp, dm, rs, tone, off)
np.savez_compressed(os.path.join(HERE, "isp_golden.npz"), **out)
print("wrote %d cases, %d bytes" % (len(out), os.path.getsize(os.path.join(HERE, "isp_golden.npz"))))
