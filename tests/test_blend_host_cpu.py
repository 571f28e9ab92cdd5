"""CPU: the HOST-side geometry of the C-ABI (op_blend_prepare, op_blend_canvas_dims,
op_cyl_warp_shape -- fp64 on the host libm, like the reference keeps them) against the oracle.
No GPU: these entry points never touch a device."""
import ctypes as C

import numpy as np
import pytest

from openpano_amd import hip, synth
from openpano_amd.config import PanoConfig


@pytest.mark.parametrize("proj,method", [("flat", 0), ("camera", 1), ("camera", 2)])
def test_blend_prepare_equals_oracle(oracle, proj, method):
    cfg = PanoConfig(MAX_OUTPUT_SIZE=300)        # forces the resolution rescale branch too
    views, homos = synth.pano_scene(5, 90, 130, seed=21 + method, proj=proj)
    shapes = [(v.shape[1], v.shape[0]) for v in views]
    g, hinv, ranges = hip.blend_prepare(cfg, shapes, homos, method, 2)
    _, meta = oracle.blend(views, homos, method, 2, PanoConfig(MAX_OUTPUT_SIZE=300, LAZY_READ=0))
    got = np.array([g.proj_min[0], g.proj_min[1], g.proj_max[0], g.proj_max[1], g.resolution[0], g.resolution[1]])
    assert np.array_equal(got, meta["geom"])
    assert np.array_equal(hinv, meta["homo_inv"]) and np.array_equal(ranges, meta["ranges"])


def test_blend_prepare_rejects_singular_homography():
    cfg = PanoConfig()
    with pytest.raises(hip.OpenPanoHipError, match="not invertible"):
        hip.blend_prepare(cfg, [(100, 80)] * 2, np.stack([np.eye(3), np.zeros((3, 3))]), 0, 0)


@pytest.mark.parametrize("w,h,hf", [(150, 100, 1.0), (97, 131, 0.9), (600, 400, 1.0)])
def test_cyl_warp_shape_equals_oracle(oracle, cfg, w, h, hf):
    rng = np.random.default_rng(5)
    pts = np.stack([rng.uniform(-w / 2, w / 2, 40), rng.uniform(-h / 2, h / 2, 40)], axis=1)
    nw, nh, off, p = hip.cyl_warp_shape(cfg, w, h, hf, pts)
    onw, onh = C.c_int(), C.c_int(); ooff = np.zeros(2); op = pts.copy()
    oracle.lib.orc_cyl_shape(w, h, hf, cfg.FOCAL_LENGTH, op.reshape(-1), len(op), C.byref(onw), C.byref(onh), ooff)
    assert (nw, nh) == (onw.value, onh.value)
    assert np.array_equal(off, ooff) and np.array_equal(p, op)
