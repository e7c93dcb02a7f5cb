#!/usr/bin/env python
"""Writes tests/golden/refprogram_golden.json: SHA-256 digests of everything the REFERENCE'S OWN PROGRAM
(oracle/_ref/TestRenderStereoPanorama = /root/reference's TestRenderStereoPanorama.cpp and its libraries compiled over
oracle/ref_shim by `make -C oracle ref`) writes for the cases of tests/refprog.py. Run where /root/reference exists:
    make -C oracle ref && python tests/golden/make_refprogram_golden.py"""
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import refprog  # noqa: E402
import rigutil  # noqa: E402

if __name__ == "__main__":
    assert os.path.exists(refprog.REF_EXE), "build it first: make -C oracle ref"
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        rig = rigutil.scaled_rig_json(os.path.join(HERE, "rig_17cam.json"), os.path.join(tmp, "rig_small.json"), refprog.CAM / 2048.0)
        for name in refprog.CASES:
            out = refprog.run_case(refprog.REF_EXE, os.path.join(tmp, name), rig, name)
            res[name] = refprog.digests(out, name)
            print(name, len(res[name]), "files")
        if os.path.exists(refprog.REF_RAW2RGB):  # the ISP program (camera_isp/Raw2Rgb.cpp)
            import hashlib
            import isputil
            res["raw2rgb"] = {}
            for name in refprog.RAW_CASES:
                _, outp = refprog.run_raw_case(refprog.REF_RAW2RGB, os.path.join(tmp, "r_" + name), isputil.CONFIG_FULL, name)
                a = refprog.png_pixels_bgr(outp)
                res["raw2rgb"][name] = hashlib.sha256(repr((a.shape, str(a.dtype))).encode() + a.tobytes()).hexdigest()
            print("raw2rgb", len(res["raw2rgb"]), "files")
    json.dump(res, open(refprog.GOLDEN, "w"), indent=0, sort_keys=True)
