#!/bin/bash
# first hardware run of k_isp_stuck (one workgroup, rows in sequence, global hand-over between its threads) + its cost
cd "$(dirname "$0")/.."
O=gpurun_out/r04_v; mkdir -p $O
timeout 120 python -m pytest tests/test_gpu_isp.py -m gpu -q -k "stuck or equals_oracle" > $O/pytest.log 2>&1
tail -4 $O/pytest.log
timeout 100 python - > $O/stuck_time.txt 2>&1 <<'PY'
import sys, time
sys.path.insert(0, "tests")
import numpy as np, torch
import isputil, oracle_lib as O
from surround360_amd import isp as I
for (sz, r, t, d, label) in ((2048, 0, 5, 0.1, "pass off"), (2048, 1, 5, 0.1, "the reference's no-op (threshold 5)"),
                             (2048, 1, 1, 0.02, "filter, few dark pixels"), (2048, 1, 1, 0.3, "filter, darkness 0.3"),
                             (1024, 1, 1, 2.0, "filter, every pixel (1024x1024)")):
    raw = isputil.bayer_frame(sz, sz, seed=2)
    js = isputil.stuck_pixel_config(r, t, d)
    isp = I.CameraIsp(I.config_from_json(js, 16))
    got = isp.get_image(raw)
    t0 = time.perf_counter()
    for _ in range(2):
        got = isp.get_image(raw)
    ms = 1e3 * (time.perf_counter() - t0) / 2
    isp.close()
    base = O.isp_run(O.isp_config_from_json(isputil.stuck_pixel_config(0, t, d), 16), raw)
    want = O.isp_run(O.isp_config_from_json(js, 16), raw) if sz <= 1024 or r == 0 or t == 5 else None
    print("%-40s %4d^2: %8.2f ms per image; differs from pass-off in %d samples%s" % (label, sz, ms, int((got != base).sum()),
          "" if want is None else "; equal to the oracle: %s" % bool(np.array_equal(got, want))), flush=True)
PY
cat $O/stuck_time.txt | grep -v amdgpu.ids
