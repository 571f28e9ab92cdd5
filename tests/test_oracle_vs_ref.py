"""CPU, build container only: the C oracle against the reference's own sources compiled in place
(oracle/_ref) on larger seeded views, stage by stage, bit-exact.  Skipped where _ref is absent
(e.g. on the GPU box if it was not prebuilt) -- the committed goldens cover that case."""
import numpy as np
import pytest

from openpano_amd import synth

VIEWS = [
    ("cfg2_600x400", 400, 600, 22),
    ("cfg4_1300x867", 867, 1300, 38),
    ("cfg3_1500x1112", 1112, 1500, 33),
    ("odd_333x777", 333, 777, 5),
]


def _view(h, w, seed):
    world = synth.make_world(seed, h + 48, w + 48, work_scale=1600.0 / (h + w))
    return synth.cut_view(world, 24, 24, h, w, seed)


@pytest.mark.parametrize("name,h,w,seed", VIEWS, ids=[v[0] for v in VIEWS])
def test_staged_sift_bit_exact(oracle, ref, name, h, w, seed):
    img = _view(h, w, seed)
    so = oracle.sift_stages(img)
    sr = ref.sift_stages(img)
    assert so.dims == sr.dims
    assert np.array_equal(so.work, sr.work)
    for kind in ("gauss", "dog", "mag", "ort"):
        a, b = getattr(so, kind), getattr(sr, kind)
        assert a.keys() == b.keys()
        for k in b:
            assert np.array_equal(a[k], b[k]), (kind, k)
    for k in sr.raw:
        assert np.array_equal(so.raw[k], sr.raw[k]), k
    for nm in ("refined", "oriented"):
        a, b = getattr(so, nm), getattr(sr, nm)
        for f in ("ints", "real", "fl"):
            assert np.array_equal(a[f], b[f]), (nm, f)
    assert len(sr.desc) > 300
    assert np.array_equal(so.desc, sr.desc)
    assert np.array_equal(so.coor, sr.coor)


def test_detect_feature_and_matchers(oracle, ref):
    from checkers import sort_features
    world = synth.make_world(77, 448, 1000, work_scale=1600.0 / (400 + 600))
    a = synth.cut_view(world, 24, 24, 400, 600, 1)
    b = synth.cut_view(world, 24, 280, 400, 600, 2)
    da, ca = sort_features(*oracle.detect_feature(a))
    ra, rca = sort_features(*ref.detect_feature(a))
    assert np.array_equal(da, ra) and np.array_equal(ca, rca)
    db, _ = oracle.detect_feature(b)
    po = oracle.match_exact(da, db)
    pr = ref.match_exact(da, db)
    assert len(pr) > 20
    assert np.array_equal(po, pr)
    assert np.array_equal(oracle.match_exact(db, da), ref.match_exact(db, da))
    # the kd-forest matcher the stitcher ships is approximate: it must agree with
    # the exact one on almost every pair, which is what lets an exact GPU matcher stand in for it
    pf = set(map(tuple, ref.match_flann(da, db)))
    pe = set(map(tuple, pr))
    assert len(pf & pe) >= 0.9 * len(pe)


def test_euclidean_sqr_early_out(oracle, ref):
    rng = np.random.default_rng(0)
    for _ in range(200):
        x = rng.random(128, dtype=np.float32) * 60
        y = rng.random(128, dtype=np.float32) * 60
        for thres in (np.float32(3.4e38), np.float32(1e4), np.float32(50.0)):
            assert oracle.euclidean_sqr(x, y, thres) == ref.euclidean_sqr(x, y, thres)


def test_gauss_kernels(oracle, ref):
    s = np.float32(1.4142135623)
    for _ in range(6):
        assert np.array_equal(oracle.gauss_kernel(s), ref.gauss_kernel(s))
        s = np.float32(s * np.float32(1.4142135623))
