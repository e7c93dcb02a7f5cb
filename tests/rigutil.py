"""Test helpers: scaled rig JSON (the reference's Camera::createRescaledCamera, Camera.cpp:271-288, applied to
the JSON) and synthetic frames, so that full-frame parity runs at sizes the CPU oracle finishes in seconds."""
import json
import os

import numpy as np

from surround360_amd import synth


def scaled_rig_json(src_path, dst_path, scale):
    rig = json.load(open(src_path))
    rig.pop("_provenance", None)
    for c in rig["cameras"]:
        res = [int(c["resolution"][0] * scale), int(c["resolution"][1] * scale)]
        sx, sy = res[0] / c["resolution"][0], res[1] / c["resolution"][1]
        c["principal"] = [c["principal"][0] * sx, c["principal"][1] * sy]
        c["focal"] = [c["focal"][0] * sx, c["focal"][1] * sy]
        c["resolution"] = res
    json.dump(rig, open(dst_path, "w"))
    return dst_path


def frame_inputs(rig_path, size, seed=360, yaw_deg=0.0, world_h=1024):
    side, top, bottom = synth.rig_frame(rig_path, size=size, world_h=world_h, seed=seed, yaw_deg=yaw_deg)
    return [np.ascontiguousarray(s) for s in side], np.ascontiguousarray(top), np.ascontiguousarray(bottom)


def pole_removal_inputs(rig_path, size, seed=360, yaw_deg=0.0, world_h=1024):
    """Everything --enable_pole_removal needs: (side, top, bottom, {id: image}) plus two synthetic red pole masks
    (BGR, pure red = masked, like res/pole_masks/*.png): a tripod-leg-like wedge in each bottom camera."""
    side, top, bottom, imgs = synth.rig_frame(rig_path, size=size, world_h=world_h, seed=seed, yaw_deg=yaw_deg,
                                              return_all=True)
    yy, xx = np.mgrid[0:size, 0:size]
    def mask(cx, half_w):
        m = np.full((size, size, 3), 255, np.uint8)
        red = (np.abs(xx - cx) < half_w + (yy * 0.04)) & (yy > size * 0.35)
        m[red] = (0, 0, 255)
        return m
    return ([np.ascontiguousarray(s) for s in side], np.ascontiguousarray(top), np.ascontiguousarray(bottom),
            {k: np.ascontiguousarray(v) for k, v in imgs.items()}, mask(size * 0.5, size * 0.04), mask(size * 0.45, size * 0.05))
