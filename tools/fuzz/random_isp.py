"""Both ISP arithmetics (the soft CameraIsp and the accelerated CameraIspPipe, s360_isp_config.pipe) with random configurations (every JSON key drawn at random, four Bayer patterns, bilinear / edge-aware
demosaic, resize 1..8, 8- / 16-bit output, tone curve on / off, black-level offsets) and random image sizes on an emulated
build of the library against the oracle, bit for bit.
usage: python tools/fuzz/random_isp.py <libs360 build> <seed> <cases>"""
import json, os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests')
from surround360_amd import _capi
_capi.LIB_PATH = sys.argv[1]
from surround360_amd import isp as I
import numpy as np
import oracle_lib as O
random.seed(int(sys.argv[2])); n = int(sys.argv[3])
def v3(lo, hi): return [round(random.uniform(lo, hi), 4) for _ in range(3)]
bad = rej = 0
reasons = {}
for i in range(n):
    rng = np.random.default_rng(900 + i)
    c = {"bayerPattern": random.choice(["RGGB", "GRBG", "GBRG", "BGGR"])}
    if random.random() < .7: c["blackLevel"] = v3(0, 3000)
    if random.random() < .5: c["clampMin"] = v3(0, 0.05)
    if random.random() < .5: c["clampMax"] = v3(0.8, 1.0)
    if random.random() < .6: c["whiteBalanceGain"] = v3(0.8, 2.2)
    if random.random() < .5: c["vignetteRollOffH"] = [v3(0.9, 1.5) for _ in range(random.randint(1, 7))]
    if random.random() < .5: c["vignetteRollOffV"] = [v3(0.9, 1.5) for _ in range(random.randint(1, 7))]
    if random.random() < .6: c["ccm"] = [[round(random.uniform(-0.4, 0.4) + (1.0 if a == b else 0.0), 3) for b in range(3)] for a in range(3)]
    if random.random() < .6: c["sharpening"] = random.choice([v3(0.05, 1.0), [0.0, 0.5, 0.5], [0.0, 0.0, 0.0]])
    if random.random() < .5: c["sharpeningSupport"] = round(random.uniform(0.001, 0.05), 5)
    if random.random() < .5: c["noiseCore"] = round(random.uniform(10, 2000), 2)
    if random.random() < .5: c["saturation"] = round(random.uniform(0.5, 1.6), 3)
    if random.random() < .5: c["contrast"] = round(random.uniform(0.6, 1.4), 3)
    if random.random() < .5: c["lowKeyBoost"] = v3(-0.4, 0.4)
    if random.random() < .5: c["highKeyBoost"] = v3(-0.4, 0.4)
    if random.random() < .6: c["gamma"] = v3(0.3, 1.2)
    if random.random() < .4:  # thresholds 2..: the reference's no-op; 0, 1, negative or above the region: its in-place median filter
        c.update(stuckPixelRadius=random.randint(0, 3), stuckPixelThreshold=random.choice([5, 2, 1, 0, -2, 30, 12]),
                 stuckPixelDarknessThreshold=random.choice([0.1, 0.5, 2.0]))
    js = json.dumps({"CameraIsp": c})
    kw = dict(output_bpp=random.choice([8, 16]), demosaic_filter=random.choice([0, 2]), resize=random.choice([1, 1, 2, 4, 8]),
              disable_tone_curve=random.choice([0, 0, 1]), black_level_offset=random.choice([0, 0, 20, -15]))
    pipe = random.choice([0, 0, 1, 1, 2])  # 1 / 2: the accelerated pipeline's arithmetic (CameraIspPipe; resize 1, odd sizes too)
    if pipe: kw["resize"] = 1
    r = kw["resize"]
    w = 2 * r * random.randint(4, max(4, 120 // r)); h = 2 * r * random.randint(4, max(4, 80 // r))
    if pipe: w, h = random.randint(16, 150), random.randint(16, 110)
    raw = rng.integers(0, 65536, (h, w), dtype=np.uint16)
    if random.random() < .3: raw = (raw // 64) * 64 + 5000 // (1 + i % 3)  # darker, banded
    try:
        cfg = I.config_from_json(js, pipe=pipe, **kw)
        isp = I.CameraIsp(cfg)
    except _capi.S360Error as e:
        rej += 1
        reasons[str(e)[:70]] = reasons.get(str(e)[:70], 0) + 1
        continue
    try:
        got = isp.get_image(raw)
    except _capi.S360Error as e:
        rej += 1; isp.close()
        reasons[str(e)[:70]] = reasons.get(str(e)[:70], 0) + 1
        continue
    want = O.isp_pipe_run(O.isp_config_from_json(js, **kw), raw, fast=pipe == 2) if pipe else O.isp_run(O.isp_config_from_json(js, **kw), raw)
    if got.shape != want.shape or got.dtype != want.dtype or not np.array_equal(got, want):
        bad += 1
        print("DIFFER", i, w, h, pipe, kw, js[:200], flush=True)
    isp.close()
print("done: %d cases, %d refused by the library %s, %d differ" % (n, rej, reasons, bad))
