"""CPU: the C restatement of the rendering path (oracle/blend_oracle.c) against the reference's
own ConnectedImages::blend / LinearBlender / MultiBandBlender / CylinderWarper compiled in place
(oracle/_ref).  Bit-exact: both sides are the same glibc, same fp order, -ffp-contract=off."""
import numpy as np
import pytest

from openpano_amd import synth
from openpano_amd.config import PanoConfig


def _cfg(**kv):
    base = dict(ESTIMATE_CAMERA=1, ORDERED_INPUT=0, LAZY_READ=0, MULTIBAND=0)
    base.update(kv)
    return PanoConfig(**base)


CASES = [
    ("flat", 0, dict(ESTIMATE_CAMERA=0, TRANS=1, ORDERED_INPUT=1)),
    ("camera", 1, dict(ESTIMATE_CAMERA=0, CYLINDER=1, ORDERED_INPUT=1)),
    ("camera", 2, dict()),
    ("camera", 2, dict(LAZY_READ=1)),
    ("camera", 2, dict(MULTIBAND=3)),
    ("flat", 0, dict(ESTIMATE_CAMERA=0, TRANS=1, ORDERED_INPUT=1, MULTIBAND=5)),
    ("camera", 2, dict(MULTIBAND=1)),
]


@pytest.mark.parametrize("proj,method,over", CASES)
def test_blend_oracle_equals_reference(ref, proj, method, over):
    from checkers import Oracle
    cfg = _cfg(**over)
    views, homos = synth.pano_scene(4, 120, 160, seed=5 + method, proj=proj)
    ref.set_config(**{k: v for k, v in cfg.raw_items()})
    ref.lib.ref_set_threads(1)
    want, wmeta = ref.blend(views, homos, method, 2)
    got, gmeta = Oracle(cfg).blend(views, homos, method, 2, cfg)
    assert want.shape == got.shape and want.shape[0] > 50 and want.shape[1] > 200
    for k in ("geom", "ranges", "homo_inv"):
        assert np.array_equal(wmeta[k], gmeta[k]), k
    assert (want >= 0).mean() > 0.5          # the canvas is mostly covered
    assert np.array_equal(want, got)


def test_blend_no_pixel_sources(ref):
    """Color::NO (-1) pixels inside a source (the output of CylinderWarper) propagate: interpolate
    returns NO if any tap is NO (lib/imgproc.cc:144-153)."""
    from checkers import Oracle
    cfg = _cfg(ESTIMATE_CAMERA=0, TRANS=1, ORDERED_INPUT=1)
    views, homos = synth.pano_scene(3, 96, 128, seed=9, proj="flat")
    views = [v.copy() for v in views]
    views[1][:20, :30] = -1.0
    views[2][40:, 100:] = -1.0
    ref.set_config(**{k: v for k, v in cfg.raw_items()})
    for mb in (0, 2):
        cfg2 = _cfg(ESTIMATE_CAMERA=0, TRANS=1, ORDERED_INPUT=1, MULTIBAND=mb)
        ref.set_config(MULTIBAND=mb)
        want, _ = ref.blend(views, homos, 0, 1)
        got, _ = Oracle(cfg2).blend(views, homos, 0, 1, cfg2)
        assert np.array_equal(want, got), mb
    ref.set_config(MULTIBAND=0)


@pytest.mark.parametrize("h,w,hf", [(100, 150, 1.0), (131, 97, 0.9)])
def test_cyl_warp_oracle_equals_reference(ref, oracle, cfg, h, w, hf):
    world = synth.make_world(77, h + 10, w + 10)
    img = world[5:5 + h, 5:5 + w]
    rng = np.random.default_rng(3)
    pts = np.stack([rng.uniform(-w / 2, w / 2, 50), rng.uniform(-h / 2, h / 2, 50)], axis=1)
    ref.set_config(**{k: v for k, v in cfg.raw_items()})
    want, wp = ref.cyl_warp(img, hf, pts)
    got, gp = oracle.cyl_warp(img, hf, pts)
    assert want.shape == got.shape
    assert np.array_equal(wp, gp)
    assert np.array_equal(want, got)
    assert (want < 0).any() and (want >= 0).mean() > 0.5


def _holey_canvas(seed, h, w):
    """a blend-like canvas: valid interior with Color::NO borders, notches and holes"""
    rng = np.random.default_rng(seed)
    m = rng.random((h, w, 3), dtype=np.float32)
    m[: rng.integers(2, 9)] = -1; m[-rng.integers(1, 7):] = -1
    m[:, : rng.integers(1, 12)] = -1; m[:, -rng.integers(3, 15):] = -1
    for _ in range(6):
        y, x = rng.integers(0, h), rng.integers(0, w)
        m[y: y + rng.integers(1, 10), x: x + rng.integers(1, 14)] = -1
    return m


@pytest.mark.parametrize("seed,h,w", [(1, 60, 90), (2, 75, 41), (3, 33, 200), (4, 5, 7)])
def test_crop_oracle_equals_reference(ref, oracle, seed, h, w):
    m = _holey_canvas(seed, h, w)
    want = ref.crop(m)
    got, _ = oracle.crop(m)
    assert want.shape == got.shape and np.array_equal(want, got)
    assert want.size > 0 or h < 10


def test_crop_all_invalid_and_all_valid(ref, oracle):
    full = np.random.default_rng(0).random((20, 30, 3), dtype=np.float32)
    assert np.array_equal(ref.crop(full), oracle.crop(full)[0]) and oracle.crop(full)[0].shape == (20, 30, 3)
    none = np.full((10, 12, 3), -1, np.float32)
    assert ref.crop(none).size == 0 and oracle.crop(none)[0].size == 0
