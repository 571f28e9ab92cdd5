"""Natural-texture restatement of the BASELINE configs -- TEST INPUTS.

Crops of the panoramas under tests/golden/natural/ (published example results of the reference
repository), each under a small seeded homography like openpano_amd.synth.cut_view.  Decoding uses
PIL; tests that compare against *committed* goldens read the decoded bytes stored in the golden
file instead, so a different libjpeg cannot break them.
"""
from __future__ import annotations

import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
NAT_DIR = os.path.join(HERE, "golden", "natural")
_cache = {}


def available():
    try:
        import PIL  # noqa: F401
    except ImportError:
        return False
    return os.path.exists(os.path.join(NAT_DIR, "uav.jpg"))


def load(name: str) -> np.ndarray:
    """decoded uint8 RGB (H, W, 3) of tests/golden/natural/<name>.jpg"""
    if name not in _cache:
        from PIL import Image
        _cache[name] = np.ascontiguousarray(np.asarray(Image.open(os.path.join(NAT_DIR, name + ".jpg")).convert("RGB")))
    return _cache[name]


def u8_to_f32(u8):
    # read_img (lib/imgio.cc:43-60): (float)byte / 255.0 evaluated in double
    return (u8.astype(np.float64) / 255.0).astype(np.float32)


def crop_u8(name, top, left, h, w, seed=None, rot_deg=2.0, persp=1e-4):
    """h x w uint8 view of panorama ``name``; with ``seed`` a small seeded homography (bilinear
    resampling) is applied, otherwise the bytes are cut as they are."""
    src = load(name)
    if seed is None:
        return np.ascontiguousarray(src[top: top + h, left: left + w])
    from openpano_amd.synth import cut_view
    m = 40                                                     # margin for the rotated footprint
    t0, l0 = max(0, top - m), max(0, left - m)
    sub = u8_to_f32(src[t0: top + h + m, l0: left + w + m])
    v = cut_view(sub, top - t0, left - l0, h, w, seed, rot_deg=rot_deg, persp=persp)
    return (np.clip(v, 0, 1) * 255 + 0.5).astype(np.uint8)


def config_views(k: int, n: int | None = None):
    """uint8 views of BASELINE config k (1..4) cut from natural texture:
      1: 2 ordered 600x400 (uav, 250 px apart);   2: 11 ordered 600x400 sliding over CMU0-all;
      3: 13 ordered 1500x1112 sliding over CMU0-all;  4: 38 unordered 1300x867 on a 2 x 19 grid over uav
    (native resolution: the x2 up-sampling SURVEY proposed leaves < 200 keypoints per view, native
    crops give the ~0.9 k the survey planned for).  ``n`` limits the number of views."""
    if k == 1:
        views = [crop_u8("uav", 700, 1500 + 250 * i, 400, 600) for i in range(2)]
    elif k == 2:
        views = [crop_u8("CMU0-all", 500, 60 + 240 * i, 400, 600, seed=1100 + i) for i in range(11)]
    elif k == 3:
        views = [crop_u8("CMU0-all", 190, 60 + 280 * i, 1112, 1500, seed=1300 + i) for i in range(13)]
    elif k == 4:
        views = []
        for i in range(38):
            r, c = divmod(i, 19)
            views.append(crop_u8("uav", 60 + 640 * r, 60 + 140 * c, 867, 1300, seed=3800 + i))
        order = np.random.default_rng(38).permutation(38)
        views = [views[j] for j in order]
    else:
        raise ValueError(k)
    return views[:n] if n else views
