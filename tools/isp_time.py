import sys, time, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch  # HIP runtime first
import isputil, oracle_lib as O
from surround360_amd import isp as I
js = isputil.CONFIG_FULL
raw = isputil.bayer_frame(2048, 2048, seed=1)
for bpp, dm in ((8, 2), (16, 2), (8, 0)):
    cfg = I.config_from_json(js, bpp, dm)
    isp = I.CameraIsp(cfg)
    out = isp.get_image(raw)
    t = time.perf_counter()
    for _ in range(5):
        out = isp.get_image(raw)
    ms = (time.perf_counter() - t) / 5 * 1e3
    isp.close()
    print("2048x2048 bpp%d dm%d: %.2f ms per image incl. PCIe both ways" % (bpp, dm, ms), flush=True)
t = time.perf_counter()
want = O.isp_run(O.isp_config_from_json(js, 16, 2), raw)
cpu = time.perf_counter() - t
cfg = I.config_from_json(js, 16, 2); isp = I.CameraIsp(cfg); got = isp.get_image(raw); isp.close()
print("oracle 2048x2048 bpp16 dm2: %.2f s; equal to GPU: %s" % (cpu, bool(np.array_equal(got, want))))
