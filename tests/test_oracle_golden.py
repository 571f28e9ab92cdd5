"""CPU: the C oracle (oracle/liboracle.so) against the committed golden vectors, which are
outputs of the reference's own sources (tests/golden/make_golden.py).  Bit-exact everywhere."""
import glob
import os
import zlib

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = sorted(glob.glob(os.path.join(HERE, "golden", "sift_*.npz"))) + \
    sorted(glob.glob(os.path.join(HERE, "golden", "nat_*x*.npz")))      # natural texture (tests/natural.py)


def u8_to_f32(u8):
    return (u8.astype(np.float64) / 255.0).astype(np.float32)


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_sift_stages_match_golden(oracle, path):
    g = np.load(path)
    st = oracle.sift_stages(u8_to_f32(g["img"]))
    assert np.array_equal(np.array(st.dims, np.int32), g["dims"])
    assert crc(st.work) == int(g["work_crc"])
    for kind in ("dog", "mag", "ort"):
        planes = getattr(st, kind)
        for key, want in zip(g[kind + "_keys"], g[kind + "_crc"]):
            assert crc(planes[tuple(int(v) for v in key)]) == int(want), (kind, key)
    raw = np.array([[len(st.raw[(o, s)]) for s in range(1, 5)] for o in range(4)], np.int32)
    assert np.array_equal(raw, g["raw_counts"])
    assert np.array_equal(st.refined["ints"], g["refined_ints"])
    assert np.array_equal(st.refined["real"], g["refined_real"])
    assert np.array_equal(st.refined["fl"][:, 1], g["refined_sf"])
    assert np.array_equal(st.oriented["ints"], g["oriented_ints"])
    assert np.array_equal(st.oriented["fl"][:, 0], g["oriented_dir"])
    assert np.array_equal(st.desc, g["desc"])
    assert np.array_equal(st.coor, g["coor"])
    # all RootSIFT descriptors have L2 norm DESC_INT_FACTOR (sift.cc:40-43)
    assert np.allclose(np.linalg.norm(st.desc.astype(np.float64), axis=1), 512.0, rtol=1e-5)


def test_detect_feature_coordinates(oracle):
    g = np.load(CASES[0])
    img = u8_to_f32(g["img"])
    desc, coor = oracle.detect_feature(img)
    assert len(desc) == int(g["n_detect"])
    assert np.array_equal(desc, g["desc"])
    # feature.cc:23-26: (c - 0.5) * {w, h}
    assert np.array_equal(coor[:, 0], (g["coor"][:, 0] - 0.5) * img.shape[1])
    assert np.array_equal(coor[:, 1], (g["coor"][:, 1] - 0.5) * img.shape[0])


def test_exact_matcher_golden(oracle):
    a = np.load(os.path.join(HERE, "golden", "sift_a_240x320.npz"))
    b = np.load(os.path.join(HERE, "golden", "sift_b_240x320.npz"))
    m = np.load(os.path.join(HERE, "golden", "match_ab.npz"))
    pairs = oracle.match_exact(a["desc"], b["desc"])
    assert np.array_equal(pairs, m["pairs"])
    # swapping the arguments swaps the pair columns (matcher.cc:21-28,68-69)
    rp = oracle.match_exact(b["desc"], a["desc"])
    assert sorted(map(tuple, rp[:, ::-1])) == sorted(map(tuple, pairs))


def test_natural_pair_match_and_ransac_golden(oracle):
    """config 1's natural-texture pair: exact matcher and TransformEstimation (homography and the
    CYLINDER-mode affine) of the oracle against what the reference produced (nat_match_uav.npz)."""
    from openpano_amd.config import PanoConfig
    a = np.load(os.path.join(HERE, "golden", "nat_uav_a_400x600.npz"))
    b = np.load(os.path.join(HERE, "golden", "nat_uav_b_400x600.npz"))
    g = np.load(os.path.join(HERE, "golden", "nat_match_uav.npz"))
    pairs = oracle.match_exact(a["desc"], b["desc"])
    assert np.array_equal(pairs, g["pairs"]) and len(pairs) > 100
    ca, cb = g["coor_a"], g["coor_b"]
    assert np.array_equal(ca, (a["coor"] - 0.5) * np.array([600.0, 400.0]))
    sh = tuple(int(v) for v in g["shape"])
    cyl = PanoConfig(CYLINDER=1, ESTIMATE_CAMERA=0, ORDERED_INPUT=1)
    for mode, c in (("homo", None), ("affine", cyl)):
        r = oracle.ransac(pairs, ca, cb, sh, sh, 38, cfg=c)
        assert bool(r["ok"]) == bool(g[mode + "_ok"]) and np.float32(r["confidence"]) == g[mode + "_conf"], mode
        pts = np.array([[ca[pairs[k][0]][0], ca[pairs[k][0]][1], cb[pairs[k][1]][0], cb[pairs[k][1]][1]] for k in r["inliers"]])
        assert np.array_equal(pts, g[mode + "_pts"]), mode
        assert np.allclose(r["homo"], g[mode + "_homo"], rtol=1e-7, atol=1e-9), mode


def test_matcher_edge_cases(oracle):
    a = np.load(os.path.join(HERE, "golden", "sift_a_240x320.npz"))["desc"]
    assert len(oracle.match_exact(a[:0], a)) == 0          # empty set
    # one-vs-one: min = 0, next_min stays FLT_MAX, 0 > 0.64*FLT_MAX is false -> accepted
    assert np.array_equal(oracle.match_exact(a[:1], a[:1]), [[0, 0]])
    # identical sets: every descriptor's best is itself at distance 0 -> accepted (0 > 0.64*d is false)
    p = oracle.match_exact(a[:50], a[:50])
    assert np.array_equal(p, np.stack([np.arange(50), np.arange(50)], 1))


def test_blend_oracle_matches_golden():
    """oracle/blend_oracle.c against the committed output of the reference's own
    ConnectedImages::blend (LinearBlender and MultiBandBlender(3)); runs anywhere."""
    import os
    from checkers import Oracle
    from openpano_amd.config import PanoConfig
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "blend_sph_linear.npz"))
    views = [(v.astype(np.float64) / 255.0).astype(np.float32) for v in z["views"]]
    for key, mb in (("linear", 0), ("multiband3", 3)):
        cfg = PanoConfig(LAZY_READ=0, MULTIBAND=mb)
        got, meta = Oracle(cfg).blend(views, z["homos"], 2, int(z["identity_idx"]), cfg)
        assert np.array_equal(got, z["canvas_" + key]), key
        assert np.array_equal(meta["geom"], z["geom"]) and np.array_equal(meta["ranges"], z["ranges"])


def _ransac_golden_check(run, g):
    """run(match, ca, cb, shape, seed, affine) -> dict(ok, confidence, homo, inliers)"""
    m, ca, cb = g["match"], g["coor_a"], g["coor_b"]
    shape = tuple(int(v) for v in g["shape"])
    for mode in ("homo", "affine"):
        for seed in (7, 20240917):
            r = run(m, ca, cb, shape, seed, mode == "affine")
            assert bool(r["ok"]) == bool(g[f"{mode}_{seed}_ok"]), (mode, seed)
            assert np.float32(r["confidence"]) == g[f"{mode}_{seed}_conf"], (mode, seed)
            if r["ok"]:
                pts = np.array([[ca[m[k][0]][0], ca[m[k][0]][1], cb[m[k][1]][0], cb[m[k][1]][1]] for k in r["inliers"]])
                assert np.array_equal(pts, g[f"{mode}_{seed}_pts"]), (mode, seed)
                # the reference solves the DLT with JacobiSVD, this side with Givens QR: equal to rounding
                assert np.allclose(r["homo"], g[f"{mode}_{seed}_homo"], rtol=1e-7, atol=1e-9), (mode, seed)


def test_ransac_oracle_matches_golden(oracle):
    """oracle/ransac_oracle.c against the committed output of the reference's own
    TransformEstimation::get_transform (seed-injected); runs anywhere."""
    from openpano_amd.config import PanoConfig
    g = np.load(os.path.join(HERE, "golden", "ransac_ab.npz"))
    cyl = PanoConfig(CYLINDER=1, ESTIMATE_CAMERA=0, ORDERED_INPUT=1)
    _ransac_golden_check(lambda m, ca, cb, sh, seed, aff: oracle.ransac(m, ca, cb, sh, sh, seed, cfg=cyl if aff else None), g)


def test_camera_estimation_golden():
    """openpano_amd/libpano_host.so against cameras the reference's own CameraEstimator produced
    (tests/golden/make_golden.py camera): runs anywhere, no oracle/_ref needed."""
    import os
    from camera_util import host_impl
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "camera_scene.npz"))
    table, at = [], 0
    for e in range(len(g["ij"])):
        c = int(g["cnt"][e])
        table.append((int(g["ij"][e, 0]), int(g["ij"][e, 1]), float(g["conf"][e]), g["homo"][e], g["pts"][at: at + c])); at += c
    host = host_impl()
    for name, mode in (("shipped", dict(MULTIPASS_BA=1, STRAIGHTEN=1, LM_LAMBDA=5.0)), ("oneshot", dict(MULTIPASS_BA=0, STRAIGHTEN=0, LM_LAMBDA=5.0))):
        host.config(**mode)
        try:
            cams = host.estimate(g["shapes"], table)
        finally:
            host.config(MULTIPASS_BA=1, STRAIGHTEN=1, LM_LAMBDA=5.0)
        assert np.array_equal(cams, g["cameras_" + name]), name
        assert np.all(np.abs(cams[:, 0] / float(g["focal"]) - 1) < 0.03)
