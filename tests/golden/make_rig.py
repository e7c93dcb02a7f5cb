"""Regenerates tests/golden/rig_17cam.json from the reference's default rig description
(surround360_render/res/config/camera_rig.json, data — not source). Run in the build
container only; /root/reference does not exist on the GPU box.  Values are carried over
verbatim (json round-trip of IEEE doubles is exact); only the serialisation differs."""
import json, sys
src = "/root/reference/surround360_render/res/config/camera_rig.json"
rig = json.load(open(src))
rig["_provenance"] = "numeric copy of facebookarchive/Surround360 surround360_render/res/config/camera_rig.json (rig fixture, RIG_JSON.md format)"
json.dump(rig, open(sys.argv[1] if len(sys.argv) > 1 else "tests/golden/rig_17cam.json", "w"), separators=(",", ":"), sort_keys=True)
