#!/usr/bin/env python
"""Writes tests/golden/refpin_golden.npz: the outputs of the REFERENCE's PixFlow.h / NovelView.cpp / CvUtil.cpp compiled
from /root/reference (oracle/_ref, see oracle/ref_pixflow.cpp, ref_render.cpp) for the cases of tests/test_cpu_refpin.py.
Run in the build container, where /root/reference exists:  python tests/golden/make_refpin_golden.py"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as O  # noqa: E402
import test_cpu_refpin as T  # noqa: E402

assert O.ref_lib("pixflow") is not None and O.ref_lib("render") is not None, "needs /root/reference (make -C oracle ref)"
rig = os.path.join(HERE, "rig_17cam.json")
R = T._render_inputs(O, rig, tempfile.mkdtemp())
out = {}
for c in T.FLOW_CASES:
    i0, i1, _ = T._flow_case(O, c)
    out["flow-%s-%s-%dx%d-%d" % c] = O.ref_compute_optical_flow(i0, i1, c[0], c[1])
(j0, j1, pf, i0, i1), _ = T._temporal(O)
out["flow-temporal"] = O.ref_compute_optical_flow(j0, j1, "pixflow_low", "LEFT", pf, i0, i1)
out["flatten"] = O.ref_flatten_layers(R["base"], R["top"])
for e in (31, 7, 5):
    out["feather-%d" % e] = O.ref_feather_alpha_channel(R["src"], e)
for off in (48.15, -196.53, 0.0, 0.5, -0.5, 159.9):
    out["offset-%g" % off] = O.ref_offset_horizontal_wrap(R["src"], off)
of = R["of"]
cl, _ = of.combine_lazy_novel_views(R["i0"], R["i1"], R["fl"], R["fr"])
out["novel-l"], out["novel-r"] = O.ref_combine_lazy_novel_views(R["i0"], R["i1"], R["fl"], R["fr"], cl.shape[1],
                                                                 of.num_novel_views, of.cam_image_width, of.verge_disp)
np.savez_compressed(os.path.join(HERE, "refpin_golden.npz"), **out)
print("wrote %d arrays, %d bytes" % (len(out), os.path.getsize(os.path.join(HERE, "refpin_golden.npz"))))
