"""Generates tests/golden/flow_small.npz: seeded inputs + the oracle's flow fields.
The reference (OpenCV-based) cannot be built or imported in this environment, so these
vectors pin OUR CPU restatement (regression guard and the GPU box's fixed comparison target), not the
reference binary: parity with the true reference remains "unpinned"."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from surround360_amd import synth  # noqa: E402

w, h, seed = 176, 208, 77
i0, i1 = synth.flow_pair(w, h, seed=seed)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "flow_small.npz"), w=w, h=h, seed=seed, i0=i0, i1=i1,
                    flow_low_left=O.compute_optical_flow(i0, i1, "pixflow_low", "LEFT"),
                    flow_search20_right=O.compute_optical_flow(i0, i1, "pixflow_search_20", "RIGHT"))
print("wrote flow_small.npz")
