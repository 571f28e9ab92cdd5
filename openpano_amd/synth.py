"""Seeded synthetic image sets for the BASELINE.json configs (no network, no reference data).

The reference's example data cannot be downloaded (run_test.py:43 needs the network), so every
config is restated as a procedural scene: a large textured "world" canvas (sum of anisotropic
Gaussian blobs + multi-octave value noise, which gives SIFT a realistic 0.5-3 k keypoints per
0.6 MP working image) from which overlapping views are cut under small seeded homographies.
float32 RGB in [0,1], HWC -- the reference's ``Mat32f`` layout (lib/mat.h:8-57).
"""
from __future__ import annotations

import numpy as np


def _value_noise(rng, h, w, octaves=6, rows=64):
    out = np.zeros((h, w), np.float32)
    amp = 1.0
    for o in range(octaves):
        gh, gw = max(2, h >> (octaves - o)), max(2, w >> (octaves - o))
        g = rng.random((gh + 1, gw + 1), dtype=np.float32)
        ys = np.linspace(0, gh, h, endpoint=False, dtype=np.float32)
        xs = np.linspace(0, gw, w, endpoint=False, dtype=np.float32)
        y0 = ys.astype(np.int32); x0 = xs.astype(np.int32)
        fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
        # the same element-wise expression as over whole planes, a band of rows at a time (whole-plane temporaries of a
        # 38-view world are ~80 MB each: the generator spent its time in page faults); the column gather once per octave
        gx0 = g[:, x0]; gx1 = g[:, x0 + 1]
        for r0 in range(0, h, rows):
            r1 = min(h, r0 + rows)
            yy = y0[r0:r1]; f = fy[r0:r1]
            a = gx0[yy]; b = gx1[yy]; c = gx0[yy + 1]; d = gx1[yy + 1]
            out[r0:r1] += amp * ((a * (1 - fx) + b * fx) * (1 - f) + (c * (1 - fx) + d * fx) * f)
        amp *= 0.6
    out -= out.min(); out /= max(out.max(), 1e-6)
    return out


def make_world(seed: int, h: int, w: int, work_scale: float = 1.0, density: float = 350.0) -> np.ndarray:
    """Textured RGB canvas, float32 HWC in [0,1].

    ``work_scale`` = SIFT working-size ratio of the views that will be cut from it
    (``1600 / (view_h + view_w)``, feature.cc:33); blob sizes/density are specified in
    *working* pixels (one blob per ``density`` working px^2, sigma 1.3-4.5 working px) so
    every config yields ~0.9 k keypoints per view, like the reference's natural textures.
    """
    rng = np.random.default_rng(seed)
    sc = float(work_scale)
    img = np.stack([_value_noise(rng, h, w) for _ in range(3)], axis=-1) * 0.3 + 0.35
    nblobs = int(h * w * sc * sc / density)
    cy = rng.uniform(0, h, nblobs); cx = rng.uniform(0, w, nblobs)
    sy = rng.uniform(1.3, 4.5, nblobs) / sc; sx = sy * rng.uniform(0.6, 1.6, nblobs)
    amp = rng.uniform(-0.65, 0.65, (nblobs, 3)).astype(np.float32)
    for i in range(nblobs):
        r = int(3 * max(sy[i], sx[i])) + 1
        y0, y1 = max(0, int(cy[i]) - r), min(h, int(cy[i]) + r + 1)
        x0, x1 = max(0, int(cx[i]) - r), min(w, int(cx[i]) + r + 1)
        if y0 >= y1 or x0 >= x1:
            continue
        yy = np.arange(y0, y1, dtype=np.float32)[:, None] - cy[i]
        xx = np.arange(x0, x1, dtype=np.float32)[None, :] - cx[i]
        g = np.exp(-(yy * yy) / (2 * sy[i] ** 2) - (xx * xx) / (2 * sx[i] ** 2)).astype(np.float32)
        img[y0:y1, x0:x1] += g[..., None] * amp[i]
    # sharp-edged bars: exercise the edge-response and contrast rejections (extrema.cc:152-168, :94)
    nbars = nblobs // 8
    by = rng.uniform(0, h, nbars); bx = rng.uniform(0, w, nbars)
    bl = rng.uniform(12, 60, nbars) / sc; bw = rng.uniform(1.0, 4.0, nbars) / sc
    ba = rng.uniform(0, np.pi, nbars)
    bamp = rng.uniform(-0.5, 0.5, (nbars, 3)).astype(np.float32)
    for i in range(nbars):
        r = int(bl[i]) + 2
        y0, y1 = max(0, int(by[i]) - r), min(h, int(by[i]) + r + 1)
        x0, x1 = max(0, int(bx[i]) - r), min(w, int(bx[i]) + r + 1)
        if y0 >= y1 or x0 >= x1:
            continue
        yy = np.arange(y0, y1, dtype=np.float32)[:, None] - by[i]
        xx = np.arange(x0, x1, dtype=np.float32)[None, :] - bx[i]
        u = xx * np.cos(ba[i]) + yy * np.sin(ba[i]); v = -xx * np.sin(ba[i]) + yy * np.cos(ba[i])
        m = np.clip(bl[i] - np.abs(u), 0, 1) * np.clip(bw[i] - np.abs(v), 0, 1)
        img[y0:y1, x0:x1] += m[..., None].astype(np.float32) * bamp[i]
    return np.clip(img, 0.0, 1.0).astype(np.float32)


def _bilinear_sample(world, ys, xs):
    h, w, _ = world.shape
    ys = np.clip(ys, 0, h - 1.001); xs = np.clip(xs, 0, w - 1.001)
    y0 = np.floor(ys).astype(np.int32); x0 = np.floor(xs).astype(np.int32)
    fy = (ys - y0)[..., None].astype(np.float32); fx = (xs - x0)[..., None].astype(np.float32)
    return (world[y0, x0] * (1 - fx) + world[y0, x0 + 1] * fx) * (1 - fy) + \
           (world[y0 + 1, x0] * (1 - fx) + world[y0 + 1, x0 + 1] * fx) * fy


def cut_view(world, top, left, h, w, seed, rot_deg=2.0, persp=1e-4):
    """One h x w view of ``world`` at (top,left) under a small seeded homography."""
    rng = np.random.default_rng(seed)
    a = np.deg2rad(rng.uniform(-rot_deg, rot_deg))
    H = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0],
                  [rng.uniform(-persp, persp), rng.uniform(-persp, persp), 1.0]])
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64) - h / 2, np.arange(w, dtype=np.float64) - w / 2, indexing="ij")
    z = H[2, 0] * xx + H[2, 1] * yy + 1.0
    sx = (H[0, 0] * xx + H[0, 1] * yy) / z + left + w / 2
    sy = (H[1, 0] * xx + H[1, 1] * yy) / z + top + h / 2
    return np.ascontiguousarray(_bilinear_sample(world, sy, sx).astype(np.float32))


def image_set(n: int, h: int, w: int, seed: int, overlap: float = 0.45, rows: int = 1, shuffle: bool = False, first: int = None):
    """``n`` overlapping h x w views on a ``rows`` x ceil(n/rows) grid over one world canvas.
    ``first`` (without ``shuffle``): only views 0..first-1 of the same set are cut (same pixels; the world is painted whole)."""
    cols = -(-n // rows)
    step_x = int(w * (1 - overlap)); step_y = int(h * (1 - overlap))
    margin = 24
    world = make_world(seed, step_y * (rows - 1) + h + 2 * margin, step_x * (cols - 1) + w + 2 * margin,
                       work_scale=1600.0 / (h + w))
    views = []
    for i in range(n if (first is None or shuffle) else min(n, first)):
        r, c = divmod(i, cols)
        views.append(cut_view(world, margin + r * step_y, margin + c * step_x, h, w, seed * 100 + i))
    if shuffle:
        order = np.random.default_rng(seed).permutation(n)
        views = [views[j] for j in order]
    return views


# BASELINE.json configs restated
CONFIGS = {
    "cfg1_2x600x400_cyl": dict(n=2, h=400, w=600, seed=11, overlap=0.58),
    "cfg2_11x600x400": dict(n=11, h=400, w=600, seed=22, overlap=0.40),
    "cfg3_13x1500x1112": dict(n=13, h=1112, w=1500, seed=33, overlap=0.40),
    "cfg4_38x1300x867": dict(n=38, h=867, w=1300, seed=38, overlap=0.45, rows=2, shuffle=True),
}


def pano_scene(n: int, h: int, w: int, seed: int, proj: str = "flat", focal: float = None, step: float = 0.55):
    """A rendering test scene: ``n`` h x w views of one world plus the homographies
    (``ImageComponent::homo``, centred image plane -> space, stitch/stitcher_image.hh:38-42) a
    successful stitch would have estimated.

    proj = "flat": translations with a small rotation / perspective jitter (TRANS / plain homography
    mode); "camera": K^-1 then a yaw/pitch/roll rotation (ESTIMATE_CAMERA / CYLINDER modes, where the
    homography maps pixels to rays).  Views are cut from a common world so overlaps agree.
    """
    rng = np.random.default_rng(seed)
    dx = int(w * step)
    world = make_world(seed, h + 80, dx * (n - 1) + w + 80, work_scale=1600.0 / (h + w), density=500.0)
    views, homos = [], []
    f = float(focal or (0.9 * w))
    for i in range(n):
        top = 40 + int(rng.integers(-12, 13)); left = 40 + i * dx
        views.append(np.ascontiguousarray(world[top: top + h, left: left + w]))
        jit = np.deg2rad(rng.uniform(-1.5, 1.5))
        if proj == "flat":
            H = np.array([[np.cos(jit), -np.sin(jit), (i - n // 2) * dx],
                          [np.sin(jit), np.cos(jit), top - 40.0],
                          [rng.uniform(-5e-5, 5e-5), rng.uniform(-5e-5, 5e-5), 1.0]])
        else:
            yaw = (i - n // 2) * dx / f; pitch = (top - 40.0) / f; roll = jit
            Ry = np.array([[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0], [-np.sin(yaw), 0, np.cos(yaw)]])
            Rx = np.array([[1, 0, 0], [0, np.cos(pitch), np.sin(pitch)], [0, -np.sin(pitch), np.cos(pitch)]])
            Rz = np.array([[np.cos(roll), -np.sin(roll), 0], [np.sin(roll), np.cos(roll), 0], [0, 0, 1]])
            H = Ry @ Rx @ Rz @ np.diag([1.0 / f, 1.0 / f, 1.0])
        homos.append(H)
    return views, np.stack(homos)


def rotating_views(n: int, h: int, w: int, seed: int, focal: float = None, step_deg: float = 20.0, rows: int = 1):
    """``n`` h x w views of a camera ROTATING about its centre (the ESTIMATE_CAMERA model: pixel
    = K R ray, stitch/camera.hh), rendered from one equirectangular world texture, so that any two
    overlapping views are related by the exact homography K R_a R_b^T K^-1.  Pixel coordinates are
    centred ((c + 0.5 - w/2, r + 0.5 - h/2), like the keypoints of FeatureDetector::detect_feature).
    Returns (views, focal, [R_i])."""
    rng = np.random.default_rng(seed)
    f = float(focal or 0.95 * w)
    cols = -(-n // rows)
    Rs = []
    for k in range(n):
        r, c = divmod(k, cols)
        yaw = np.deg2rad((c - (cols - 1) / 2) * step_deg + rng.uniform(-1.0, 1.0))
        pitch = np.deg2rad((r - (rows - 1) / 2) * 0.55 * np.rad2deg(2 * np.arctan(h / 2 / f)) + rng.uniform(-1.0, 1.0))
        roll = np.deg2rad(rng.uniform(-1.5, 1.5))
        cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
        Ry = np.array([[cy, 0, -sy], [0, 1, 0], [sy, 0, cy]])
        Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
        Rz = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])
        Rs.append(Rz @ Rx @ Ry)
    half_x = np.deg2rad((cols - 1) / 2 * step_deg + 4) + np.arctan(np.hypot(w, h) / 2 / f)
    half_y = np.deg2rad(4) + (rows - 1) / 2 * 0.55 * 2 * np.arctan(h / 2 / f) + np.arctan(np.hypot(w, h) / 2 / f)
    Hw, Ww = int(2 * half_y * f) + 8, int(2 * half_x * f) + 8
    world = make_world(seed, Hw, Ww, work_scale=1600.0 / (h + w), density=500.0)
    u = np.arange(w) + 0.5 - w / 2; v = np.arange(h) + 0.5 - h / 2
    uu, vv = np.meshgrid(u, v)
    rays = np.stack([uu / f, vv / f, np.ones_like(uu)], -1)
    views = []
    for R in Rs:
        X = rays @ R                                   # R^T ray, row-vector form
        theta = np.arctan2(X[..., 0], X[..., 2]); phi = np.arctan2(X[..., 1], np.hypot(X[..., 0], X[..., 2]))
        xs = (theta + half_x) * f; ys = (phi + half_y) * f
        xs = np.clip(xs, 0, Ww - 1.001); ys = np.clip(ys, 0, Hw - 1.001)
        views.append(np.ascontiguousarray(_bilinear_sample(world, ys, xs).astype(np.float32)))
    return views, f, Rs


_c5_base = {}


def _torch_world(seed: int, h: int, w: int, device, density: float):
    """A ``make_world``-like canvas built with torch ops on ``device`` (3 x h x w, float32): seeded
    impulses blurred by separable anisotropic Gaussians (8 size classes, sigma 1.3-4.5 px) over
    multi-octave value noise.  Positions / amplitudes come from a numpy generator, so the scene is
    the same on every rank; milliseconds instead of the seconds of the per-blob numpy loop."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(seed)
    img = torch.zeros((3, h, w), device=device)
    amp = 1.0
    for o in range(6):                                         # value noise, coarse to fine
        gh, gw = max(2, h >> (6 - o)), max(2, w >> (6 - o))
        g = torch.from_numpy(rng.random((1, 3, gh + 1, gw + 1), dtype=np.float32)).to(device)
        img += amp * F.interpolate(g, size=(h, w), mode="bilinear", align_corners=True)[0]
        amp *= 0.6
    img -= img.amin(); img /= img.amax().clamp_min(1e-6)
    img = img * 0.3 + 0.35
    nblobs = int(h * w / density)
    classes = 8
    for c in range(classes):
        sy = 1.3 + (4.5 - 1.3) * (c + 0.5) / classes
        sx = sy * float(rng.uniform(0.6, 1.6))
        nb = nblobs // classes
        ys = rng.integers(0, h, nb); xs = rng.integers(0, w, nb)
        a = rng.uniform(-0.65, 0.65, (nb, 3)).astype(np.float32) * np.float32(2 * np.pi * sx * sy)
        imp = torch.zeros((3, h * w), device=device)
        idx = torch.from_numpy((ys * w + xs).astype(np.int64)).to(device)
        imp.index_add_(1, idx, torch.from_numpy(a.T.copy()).to(device))
        imp = imp.view(3, 1, h, w)

        def kern(sig):
            r = int(3 * sig) + 1
            t = torch.arange(-r, r + 1, device=device, dtype=torch.float32)
            k = torch.exp(-t * t / (2 * sig * sig)); return k / k.sum(), r
        ky, ry = kern(sy); kx, rx = kern(sx)
        imp = F.conv2d(imp, ky.view(1, 1, -1, 1), padding=(ry, 0))
        imp = F.conv2d(imp, kx.view(1, 1, 1, -1), padding=(0, rx))
        img += imp.view(3, h, w)
    return img.clamp_(0, 1)


def config5_views(indices, device, h: int = 3000, w: int = 4000, group: int = 8, density: float = 72.0):
    """BASELINE config 5 restated: 4000x3000 uint8 RGB images, groups of
    ``group`` sharing one base texture under seeded homographies so that true matches exist.

    The base texture of a group is a seeded blob canvas at a quarter of the resolution (~ the
    914x685 SIFT working size; blob density chosen for K ~ 3-5 k keypoints per image); each view
    resamples it bilinearly on ``device`` (torch is plumbing here: a 36 MB image in milliseconds
    instead of seconds of numpy) and adds seeded sensor noise.  Returns uint8 (h, w, 3) tensors.
    """
    import torch
    import torch.nn.functional as F
    out = []
    bh, bw = h // 4 + 100, w // 4 + 100
    ys = torch.arange(h, device=device, dtype=torch.float32) - h / 2
    xs = torch.arange(w, device=device, dtype=torch.float32) - w / 2
    yy, xx = torch.meshgrid(ys, xs, indexing="ij")
    for i in indices:
        g = int(i) // group
        key = (g, str(device), bh, bw, density)
        if key not in _c5_base:
            _c5_base.clear()                                   # one resident base texture is enough
            _c5_base[key] = _torch_world(50000 + g, bh, bw, device, density)[None]     # 1 x 3 x bh x bw
        base = _c5_base[key]
        rng = np.random.default_rng(51000 + int(i))
        a = np.deg2rad(rng.uniform(-2.0, 2.0))
        tx, ty = rng.uniform(-40, 40, 2)                       # base pixels
        px, py = rng.uniform(-2e-5, 2e-5, 2)
        z = px * xx + py * yy + 1.0
        sx = ((np.cos(a) * xx - np.sin(a) * yy) / z) / 4 + (bw / 2 + tx)
        sy = ((np.sin(a) * xx + np.cos(a) * yy) / z) / 4 + (bh / 2 + ty)
        grid = torch.stack([sx / (bw - 1) * 2 - 1, sy / (bh - 1) * 2 - 1], -1)[None]
        v = F.grid_sample(base, grid, mode="bilinear", padding_mode="border", align_corners=True)[0].permute(1, 2, 0)
        gen = torch.Generator(device=device); gen.manual_seed(52000 + int(i))
        v = v + 0.01 * torch.randn(v.shape, generator=gen, device=device, dtype=torch.float32)
        out.append((v.clamp_(0, 1) * 255 + 0.5).to(torch.uint8).contiguous())
    return out
