"""Seeded synthetic inputs: there is no sample footage offline, so every
config is driven by band-limited noise textures.

  * flow_pair(): two BGRA images related by a smooth disparity (config 2).
  * rig_frame(): 17 camera images rendered from one equirect world texture through the
    rig's camera model, with parallax from per-pixel depth (configs 1, 3-5).

numpy only; used by tests and bench.py to build inputs (host side, outside timed regions).
"""
import json

import numpy as np


def _smooth_noise(rng, h, w, octaves=3, base=8):
    """Sum of bilinearly upsampled white-noise grids (band-limited)."""
    out = np.zeros((h, w), np.float32)
    amp = 1.0
    for o in range(octaves):
        gh, gw = base * (2 ** o) + 1, base * (2 ** o) * max(1, w // h) + 1
        g = rng.random((gh, gw), dtype=np.float32)
        ys = np.linspace(0, gh - 1, h, dtype=np.float32)
        xs = np.linspace(0, gw - 1, w, dtype=np.float32)
        y0 = np.clip(ys.astype(np.int32), 0, gh - 2)
        x0 = np.clip(xs.astype(np.int32), 0, gw - 2)
        fy = (ys - y0)[:, None]
        fx = (xs - x0)[None, :]
        a = g[y0][:, x0]
        b = g[y0][:, x0 + 1]
        c = g[y0 + 1][:, x0]
        d = g[y0 + 1][:, x0 + 1]
        out += amp * ((a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy)
        amp *= 0.5
    out -= out.min()
    out /= max(out.max(), 1e-6)
    return out


def texture_bgr(h, w, seed=360, octaves=5, base=6):
    rng = np.random.default_rng(seed)
    chans = [_smooth_noise(rng, h, w, octaves, base) for _ in range(3)]
    img = np.stack(chans, axis=-1)
    return np.clip(img * 255.0, 0, 255).astype(np.uint8)


def _bilinear_sample(img, x, y, wrap_x=False):
    h, w = img.shape[:2]
    if wrap_x:
        x = np.mod(x, w)
    else:
        x = np.clip(x, 0, w - 1.001)
    y = np.clip(y, 0, h - 1.001)
    x0 = np.floor(x).astype(np.int32)
    y0 = np.floor(y).astype(np.int32)
    x1 = (x0 + 1) % w if wrap_x else np.minimum(x0 + 1, w - 1)
    y1 = np.minimum(y0 + 1, h - 1)
    fx = (x - x0)[..., None].astype(np.float32)
    fy = (y - y0)[..., None].astype(np.float32)
    a = img[y0, x0].astype(np.float32)
    b = img[y0, x1].astype(np.float32)
    c = img[y1, x0].astype(np.float32)
    d = img[y1, x1].astype(np.float32)
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def side_alpha_ramp(img_bgra, feather=100):
    """Alpha ramp of projectSideToSpherical (TestRenderStereoPanorama.cpp:116-125)."""
    h = img_bgra.shape[0]
    for y in range(min(feather, h // 2)):
        a = np.uint8(np.float32(255.0) * np.float32(y + 0.5) / np.float32(feather))
        img_bgra[y, :, 3] = a
        img_bgra[h - 1 - y, :, 3] = a
    return img_bgra


def flow_pair(w, h, seed=360, max_disp=None, feather=None):
    """I0 = texture; I1 = I0 sampled through a smooth horizontal disparity (+ small vertical jitter)."""
    rng = np.random.default_rng(seed + 1)
    tex = texture_bgr(h, w + 128, seed)
    max_disp = max_disp if max_disp is not None else max(4.0, 40.0 * w / 2048.0)
    depth = _smooth_noise(rng, h, w, 2, 2)
    disp = (0.2 + 0.8 * depth) * max_disp
    jit = (_smooth_noise(rng, h, w, 2, 3) - 0.5) * 2.0 * min(1.0, max_disp / 8.0)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    i0 = _bilinear_sample(tex, xx + 64, yy)
    i1 = _bilinear_sample(tex, xx + 64 + disp, yy + jit)
    out = []
    for im in (i0, i1):
        bgra = np.concatenate([np.clip(im, 0, 255).astype(np.uint8), np.full((h, w, 1), 255, np.uint8)], axis=-1)
        feather_n = feather if feather is not None else max(2, int(100 * h / 2048))
        out.append(side_alpha_ramp(np.ascontiguousarray(bgra), feather_n))
    return out[0], out[1]


def world_texture(h=2048, seed=360):
    """Equirect world (2h x h) BGR + a depth map in rig units (cm): far field + near blobs."""
    w = 2 * h
    tex = texture_bgr(h, w, seed, octaves=6, base=8)
    rng = np.random.default_rng(seed + 7)
    blobs = _smooth_noise(rng, h, w, 2, 3)
    depth = np.where(blobs > 0.72, 200.0, np.where(blobs > 0.6, 500.0, 1.0e6)).astype(np.float32)
    # high-contrast markers on the near objects so that parallax is visible to the flow
    near = depth < 1e5
    tex = tex.copy()
    tex[near] = (tex[near].astype(np.int32) * 3 // 4 + np.array([40, 10, 60])).clip(0, 255).astype(np.uint8)
    return tex, depth


def _camera_rays(cam, res):
    """Unit rays (rig space) through the pixel centres of camera `cam` (rig JSON dict): inverse of
    Camera::pixel for zero distortion (Camera.h:143-150, 264-284)."""
    w, h = res
    sx = cam["resolution"][0] / w
    fwd, up, right = (np.asarray(cam[k], np.float64) for k in ("forward", "up", "right"))
    R = np.stack([right, up, -fwd])
    # orthonormalise like Camera::setRotation (close enough for input synthesis)
    u, _, vt = np.linalg.svd(R)
    R = u @ vt
    px = (np.arange(w) + 0.5) * sx
    py = (np.arange(h) + 0.5) * sx
    X, Y = np.meshgrid(px, py)
    pr = cam.get("principal", [cam["resolution"][0] / 2, cam["resolution"][1] / 2])
    sxn = (X - pr[0]) / cam["focal"][0]
    syn = (Y - pr[1]) / cam["focal"][1]
    r = np.sqrt(sxn * sxn + syn * syn) + 1e-12
    ang = r if cam["type"] == "FTHETA" else np.arctan(r)
    s = np.sin(ang) / r
    unit = np.stack([s * sxn, s * syn, -np.cos(ang)], axis=-1)
    return unit @ R  # R^T * unit per pixel


def rig_frame(rig_json_path, size=2048, world_h=2048, seed=360, yaw_deg=0.0, return_all=False):
    """Render every camera of the rig from the seeded world. Returns (list of side BGR images in
    rigSideOnly order, top BGR, bottom BGR). One fixed-point iteration places each ray's hit
    point at the world depth seen from the rig centre, which gives consistent parallax between
    adjacent cameras. return_all=True also returns the dict {camera id: image} of every camera of the rig (the
    secondary bottom camera of pole removal is not one of the three standard outputs)."""
    with open(rig_json_path) as f:
        cams = json.load(f)["cameras"]
    tex, depth = world_texture(world_h, seed)
    H, W = tex.shape[:2]
    imgs = {}
    for cam in cams:
        rays = _camera_rays(cam, (size, size))
        org = np.asarray(cam["origin"], np.float64)
        p = org + rays * 1.0e6
        for _ in range(2):
            n = p / np.linalg.norm(p, axis=-1, keepdims=True)
            th = np.arctan2(n[..., 1], n[..., 0]) + np.deg2rad(yaw_deg)
            ph = np.arccos(np.clip(n[..., 2], -1, 1))
            u = (np.mod(-th, 2 * np.pi)) / (2 * np.pi) * W
            v = ph / np.pi * H
            d = depth[np.clip(v.astype(np.int32), 0, H - 1), np.clip(u.astype(np.int32), 0, W - 1)]
            # intersect the ray with the sphere of radius d around the rig centre
            b = (rays * org).sum(-1)
            c = (org * org).sum() - d.astype(np.float64) ** 2
            t = -b + np.sqrt(np.maximum(b * b - c, 0))
            p = org + rays * t[..., None]
        imgs[cam["id"]] = np.clip(_bilinear_sample(tex, u.astype(np.float32), v.astype(np.float32), wrap_x=True),
                                  0, 255).astype(np.uint8)
    side = [imgs[c["id"]] for c in cams if "side" in c.get("group", "")]
    # RigDescription::findCameraByDirection(+-Z) with axis distance <= 1 (RigDescription.cpp:33-47)
    def axis_dist(c):
        f = np.asarray(c["forward"], np.float64)
        o = np.asarray(c["origin"], np.float64)
        return np.linalg.norm(-o - f * np.dot(f, -o))
    ok = [c for c in cams if axis_dist(c) <= 1.0]
    top = max(ok, key=lambda c: c["forward"][2])
    bottom = max(ok, key=lambda c: -c["forward"][2])
    if return_all:
        return side, imgs[top["id"]], imgs[bottom["id"]], imgs
    return side, imgs[top["id"]], imgs[bottom["id"]]


# ------------------------------------------------------------------------------------------------------------------
# torch versions of the generators above for the full-size configurations (17 x 2048^2 cameras per frame, a 190-frame
# stream): the numpy ray caster takes minutes per frame, this one milliseconds on the GPU (it also runs on CPU
# tensors). Same construction — seeded band-limited world texture + depth, rays through the rig's camera model, one
# fixed-point iteration for parallax — but NOT bit-identical to the numpy path; inputs only have to be the same for
# the two sides of a comparison, which get the same arrays.
class World:
    """Seeded equirect world (2h x h BGR float texture + depth in cm) resident on `device`; `disc` adds one moving
    high-contrast disc (BASELINE configs[4]: "one disc translating")."""

    def __init__(self, h=4096, seed=360, device="cpu"):
        import torch
        import torch.nn.functional as F
        self.torch, self.F = torch, F
        self.h, self.w, self.device = h, 2 * h, torch.device(device)
        g = torch.Generator(device="cpu").manual_seed(seed)

        def smooth(octaves, base, ch):
            out = torch.zeros(ch, h, 2 * h, device=self.device)
            amp = 1.0
            for o in range(octaves):
                gh, gw = base * (2 ** o) + 1, 2 * base * (2 ** o) + 1
                grid = torch.rand(1, ch, gh, gw, generator=g)
                grid[..., -1] = grid[..., 0]  # periodic in azimuth: no seam where the equirect wraps
                grid = grid.to(self.device)
                out += amp * F.interpolate(grid, size=(h, 2 * h), mode="bilinear", align_corners=True)[0]
                amp *= 0.5
            out -= out.amin(dim=(1, 2), keepdim=True)
            out /= out.amax(dim=(1, 2), keepdim=True).clamp_min(1e-6)
            return out

        tex = (smooth(7, 8, 3) * 255.0).clamp(0, 255)
        blobs = smooth(2, 3, 1)[0]
        depth = torch.where(blobs > 0.72, 200.0, torch.where(blobs > 0.6, 500.0, 1.0e6))
        near = depth < 1e5
        mark = torch.tensor([40.0, 10.0, 60.0], device=self.device)[:, None, None]
        tex = torch.where(near[None], (tex * 0.75 + mark).clamp(0, 255), tex)
        self.tex, self.depth = tex.contiguous(), depth.contiguous()


def _rays_torch(cam, size, device, torch):
    sx = cam["resolution"][0] / size
    fwd, up, right = (np.asarray(cam[k], np.float64) for k in ("forward", "up", "right"))
    Rm = np.stack([right, up, -fwd])
    u, _, vt = np.linalg.svd(Rm)
    Rm = torch.tensor(u @ vt, dtype=torch.float32, device=device)
    p = (torch.arange(size, device=device, dtype=torch.float32) + 0.5) * sx
    Y, X = torch.meshgrid(p, p, indexing="ij")
    pr = cam.get("principal", [cam["resolution"][0] / 2, cam["resolution"][1] / 2])
    sxn = (X - pr[0]) / cam["focal"][0]
    syn = (Y - pr[1]) / cam["focal"][1]
    r = torch.sqrt(sxn * sxn + syn * syn) + 1e-12
    ang = r if cam["type"] == "FTHETA" else torch.atan(r)
    s = torch.sin(ang) / r
    unit = torch.stack([s * sxn, s * syn, -torch.cos(ang)], dim=-1)
    return unit @ Rm


class RigRenderer:
    """Renders the cameras of a rig JSON from a World; rays are cached per camera. frame(yaw_deg, disc_deg) returns
    ([side BGR uint8 HxWx3 ...], top, bottom) as tensors on the world's device."""

    def __init__(self, rig_json_path, world, size=2048):
        self.world, self.size = world, size
        torch = world.torch
        with open(rig_json_path) as f:
            self.cams = json.load(f)["cameras"]
        self.rays = [_rays_torch(c, size, world.device, torch) for c in self.cams]
        self.org = [torch.tensor(c["origin"], dtype=torch.float32, device=world.device) for c in self.cams]

        def axis_dist(c):
            fw = np.asarray(c["forward"], np.float64)
            o = np.asarray(c["origin"], np.float64)
            return np.linalg.norm(-o - fw * np.dot(fw, -o))
        ok = [i for i, c in enumerate(self.cams) if axis_dist(c) <= 1.0]
        self.top = max(ok, key=lambda i: self.cams[i]["forward"][2])
        self.bottom = max(ok, key=lambda i: -self.cams[i]["forward"][2])
        self.side = [i for i, c in enumerate(self.cams) if "side" in c.get("group", "")]

    def _camera(self, i, yaw_deg, disc_deg):
        W = self.world
        torch, F = W.torch, W.F
        rays, org = self.rays[i], self.org[i]
        p = org + rays * 1.0e6
        for _ in range(2):
            n = p / p.norm(dim=-1, keepdim=True)
            th = torch.atan2(n[..., 1], n[..., 0]) + float(np.deg2rad(yaw_deg))
            ph = torch.acos(n[..., 2].clamp(-1, 1))
            u = torch.remainder(-th, 2 * np.pi) / (2 * np.pi) * W.w
            v = ph / np.pi * W.h
            d = W.depth[v.long().clamp(0, W.h - 1), u.long().clamp(0, W.w - 1)]
            b = (rays * org).sum(-1)
            c = (org * org).sum() - d * d
            t = -b + torch.sqrt((b * b - c).clamp_min(0))
            p = org + rays * t[..., None]
        # bilinear, wrapping in x: grid_sample on a texture padded by one column
        tex = torch.cat([W.tex, W.tex[:, :, :1]], dim=2)[None]
        gx = (u / W.w) * 2 - 1  # align_corners=True over W.w + 1 columns: x in [0, W.w] -> [-1, 1]
        gy = (v.clamp(0, W.h - 1.001) / (W.h - 1)) * 2 - 1
        img = F.grid_sample(tex, torch.stack([gx, gy], -1)[None], mode="bilinear", padding_mode="border", align_corners=True)[0]
        if disc_deg is not None:  # a dark disc with a bright rim, 3 degrees across, on the horizon at azimuth disc_deg
            az = torch.remainder(th - float(np.deg2rad(disc_deg)) + np.pi, 2 * np.pi) - np.pi
            rr = torch.sqrt(az * az + (ph - np.pi / 2) ** 2) / float(np.deg2rad(1.5))
            img = torch.where((rr < 1.0)[None], torch.where((rr < 0.8)[None], img * 0.2, img * 0 + 250.0), img)
        return img.clamp(0, 255).to(torch.uint8).permute(1, 2, 0).contiguous()

    def frame(self, yaw_deg=0.0, disc_deg=None):
        imgs = [self._camera(i, yaw_deg, disc_deg) for i in range(len(self.cams))]
        return [imgs[i] for i in self.side], imgs[self.top], imgs[self.bottom]

    def frame_all_numpy(self, yaw_deg=0.0, disc_deg=None):
        """Every camera of the rig in JSON order (pole removal needs the secondary bottom camera as well)."""
        return [self._camera(i, yaw_deg, disc_deg).cpu().numpy() for i in range(len(self.cams))]

    def frame_numpy(self, yaw_deg=0.0, disc_deg=None):
        side, top, bottom = self.frame(yaw_deg, disc_deg)
        return [s.cpu().numpy() for s in side], top.cpu().numpy(), bottom.cpu().numpy()
