"""GPU parity of the batched RANSAC (C-ABI op_ransac_pairs) against the oracle with the same
per-pair mt19937 seeds: winning hypothesis, inlier set, refit homography (bit-exact), acceptance
decision and confidence."""
import numpy as np
import pytest

from openpano_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from openpano_amd import hip
    c = hip.Context(0)
    yield c
    c.close()


def _check(oracle, got, m, ca, cb, s1, s2, seed, cfg=None):
    want = oracle.ransac(m, ca, cb, s1, s2, seed, cfg=cfg)
    assert got["best_hyp"] == want["best_hyp"] and got["best_count"] == want["best_count"]
    assert got["ok"] == want["ok"]
    assert got["confidence"] == want["confidence"]
    assert np.array_equal(got["inliers"], want["inliers"])
    if want["ok"]:
        assert np.array_equal(got["homo"], want["homo"])
    return want


def test_all_pairs_ransac_vs_oracle(ctx, oracle, cfg):
    from openpano_amd import hip
    n = 5
    views = synth.image_set(n, 400, 600, seed=22, overlap=0.45)
    f = hip.sift_batch(ctx, cfg, views)
    pairs = [(i, j) for i in range(n) for j in range(i + 1, n)] + [(3, 1)]
    mh = hip.match_pairs_handle(ctx, cfg, f, pairs)
    lists = mh.lists()
    seeds = [1000 + 7 * k for k in range(len(pairs))]
    res = hip.ransac_pairs(ctx, cfg, f, mh, pairs, [(600, 400)] * n, seeds=seeds)
    coors = [f.get(i)[1] for i in range(n)]
    nok = 0
    for k, (i, j) in enumerate(pairs):
        w = _check(oracle, res[k], lists[k], coors[i], coors[j], (600, 400), (600, 400), seeds[k])
        nok += w["ok"]
    assert nok >= 4          # the adjacent views overlap by 45 %
    # same seeds -> same result (reproducible, unlike the reference's random_device seeding)
    res2 = hip.ransac_pairs(ctx, cfg, f, mh, pairs, [(600, 400)] * n, seeds=seeds)
    for a, b in zip(res, res2):
        assert a["best_hyp"] == b["best_hyp"] and np.array_equal(a["homo"], b["homo"])
    mh.free(); f.free()


def test_affine_and_degenerate_pairs(ctx, oracle, cfg):
    from openpano_amd import hip
    from openpano_amd.config import PanoConfig
    cyl = PanoConfig(CYLINDER=1, ESTIMATE_CAMERA=0, ORDERED_INPUT=1)
    views = synth.image_set(3, 400, 600, seed=5, overlap=0.5)
    f = hip.sift_batch(ctx, cfg, views)
    coors = [f.get(i)[1] for i in range(3)]
    pairs = [(0, 1), (1, 2), (0, 2), (2, 0)]
    mh = hip.match_pairs_handle(ctx, cfg, f, pairs)
    lists = mh.lists()
    # degenerate inputs through host-provided match lists: empty, 7 matches, all-identical match
    lists2 = [lists[0], lists[1][:7], lists[2][:0], np.repeat(lists[0][:1], 20, axis=0)]
    mh2 = hip.Matches.from_host(lists2)
    seeds = [11, 12, 13, 14]
    res = hip.ransac_pairs(ctx, cyl, f, mh2, pairs, [(600, 400)] * 3, seeds=seeds)
    for k, (i, j) in enumerate(pairs):
        _check(oracle, res[k], lists2[k], coors[i], coors[j], (600, 400), (600, 400), seeds[k], cfg=cyl)
    assert res[0]["ok"] and res[0]["homo"][2, 0] == 0 and res[0]["homo"][2, 2] == 1
    assert not res[1]["ok"] and not res[2]["ok"] and not res[3]["ok"]
    mh.free(); mh2.free(); f.free()


def test_ransac_golden_fixture(ctx, cfg):
    """Committed output of the REFERENCE's TransformEstimation (tests/golden/ransac_ab.npz)."""
    import os
    from openpano_amd import hip
    from openpano_amd.config import PanoConfig
    from test_oracle_golden import _ransac_golden_check
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ransac_ab.npz"))
    cyl = PanoConfig(CYLINDER=1, ESTIMATE_CAMERA=0, ORDERED_INPUT=1)

    def run(m, ca, cb, shape, seed, affine):
        f = hip.Features.from_host(ctx, [np.zeros((len(ca), 128), np.float32), np.zeros((len(cb), 128), np.float32)], [ca, cb])
        mh = hip.Matches.from_host([m])
        r = hip.ransac_pairs(ctx, cyl if affine else cfg, f, mh, [(0, 1)], [shape, shape], seeds=[seed])[0]
        mh.free(); f.free()
        return r
    _ransac_golden_check(run, g)


def test_ransac_rejects_foreign_match_indices(ctx, cfg):
    """match lists can come from op_matches_from_host / another op_features; an index
    outside the image's keypoints is OP_ERR_INVALID, not a host heap read out of bounds."""
    from openpano_amd import hip
    ca = np.random.default_rng(0).random((20, 2)) * 100
    f = hip.Features.from_host(ctx, [np.zeros((20, 128), np.float32), np.zeros((10, 128), np.float32)], [ca, ca[:10]])
    bad = hip.Matches.from_host([np.array([[0, 0], [19, 10]], np.int32)])          # second index: 10 >= counts[1]
    with pytest.raises(hip.OpenPanoHipError):
        hip.ransac_pairs(ctx, cfg, f, bad, [(0, 1)], [(100, 100)] * 2, seeds=[1])
    two = hip.Matches.from_host([np.zeros((0, 2), np.int32)] * 2)
    with pytest.raises(hip.OpenPanoHipError):                                      # op_matches of 2 pairs, pair list of 1
        hip.ransac_pairs(ctx, cfg, f, two, [(0, 1)], [(100, 100)] * 2, seeds=[1])
    bad.free(); two.free(); f.free()


def test_large_match_lists_span_several_point_chunks(ctx, oracle, cfg):
    """Config-5-sized pairs: thousands of matches between two images of K ~ 4 k keypoints (the hypothesis kernel
    stages the pair's points through LDS 512 at a time, the acceptance gates walk every keypoint), planted on a
    true homography with 35 % outliers; m = 513 / 1024 / 1025 / 2600 / 4000 sit on and across the chunk
    boundaries.  Winner, inlier set, homography and confidence equal the oracle's."""
    from openpano_amd import hip
    rng = np.random.default_rng(77)
    K, W, H = 4000, 4000, 3000
    ca = np.stack([rng.uniform(-W / 2, W / 2, K), rng.uniform(-H / 2, H / 2, K)], 1)
    Ht = np.array([[1.01, 0.02, 310.0], [-0.015, 0.99, -42.0], [2e-6, -1e-6, 1.0]])
    q = np.concatenate([ca, np.ones((K, 1))], 1) @ np.linalg.inv(Ht).T
    cb = q[:, :2] / q[:, 2:3] + rng.normal(0, 0.6, (K, 2))
    out = rng.random(K) < 0.35
    cb[out] = np.stack([rng.uniform(-W / 2, W / 2, out.sum()), rng.uniform(-H / 2, H / 2, out.sum())], 1)
    f = hip.Features.from_host(ctx, [np.zeros((K, 128), np.float32)] * 2, [ca, cb])
    sizes = [513, 1024, 1025, 2600, 4000]
    lists = []
    for m in sizes:
        a = np.sort(rng.choice(K, m, replace=False)).astype(np.int32)
        lists.append(np.stack([a, a], 1))
    mh = hip.Matches.from_host(lists)
    pairs = [(0, 1)] * len(sizes)
    seeds = [500 + k for k in range(len(sizes))]
    res = hip.ransac_pairs(ctx, cfg, f, mh, pairs, [(W, H)] * 2, seeds=seeds)
    for k in range(len(sizes)):
        w = _check(oracle, res[k], lists[k], ca, cb, (W, H), (W, H), seeds[k])
        assert w["ok"] and len(w["inliers"]) > 0.5 * sizes[k]
    mh.free(); f.free()


def test_small_match_lists_span_several_stream_chunks(ctx, oracle, cfg):
    """With few matches almost every draw repeats an index already in the sample (m = 8: ~22 draws per
    8-point sample), so 1500 hypotheses need far more than one LDS chunk of the mt19937 stream: the
    sample kernel carries its incomplete sample across chunks.  Same winner / inliers as the oracle."""
    from openpano_amd import hip
    from openpano_amd.config import PanoConfig
    views = synth.image_set(2, 400, 600, seed=5, overlap=0.5)
    f = hip.sift_batch(ctx, cfg, views)
    coors = [f.get(i)[1] for i in range(2)]
    full = hip.match_pairs(ctx, cfg, f, [(0, 1)])[0]
    assert len(full) > 70
    sizes = [8, 9, 10, 12, 16, 20, 33, 64, 65, len(full)]
    lists = [full[:: max(1, len(full) // k)][:k] for k in sizes]
    assert [len(x) for x in lists] == sizes
    mh = hip.Matches.from_host(lists)
    pairs = [(0, 1)] * len(sizes)
    seeds = [900 + k for k in range(len(sizes))]
    cyl = PanoConfig(CYLINDER=1, ESTIMATE_CAMERA=0, ORDERED_INPUT=1)
    for c in (cfg, cyl):                                 # 8-point homography and 7-point affine samples
        res = hip.ransac_pairs(ctx, c, f, mh, pairs, [(600, 400)] * 2, seeds=seeds)
        for k in range(len(sizes)):
            _check(oracle, res[k], lists[k], coors[0], coors[1], (600, 400), (600, 400), seeds[k], cfg=c if c is cyl else None)
    mh.free(); f.free()


def test_pairwise_table_equals_match_image_bookkeeping(ctx, oracle, cfg):
    """op_pairwise_table = Stitcher::match_image's bookkeeping (stitch/stitcher.cc:79-93) for the whole job: both directed
    entries of every accepted pair -- (i, j): the pair's MatchInfo; (j, i): homo.inverse() scaled by 1 / inv[8], matches
    reversed -- rebuilt here from the per-pair results and the reference's own Homography::inverse (oracle/_ref when it is
    there, else the host library's restatement)."""
    import ctypes as C
    import os
    from openpano_amd import hip
    n = 5
    views = synth.image_set(n, 400, 600, seed=22, overlap=0.45)
    f = hip.sift_batch(ctx, cfg, views)
    pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]
    shapes = [(600, 400)] * n
    mh = hip.match_pairs_handle(ctx, cfg, f, pairs)
    lists = mh.lists()
    res = hip.ransac_pairs(ctx, cfg, f, mh, pairs, shapes, base_seed=42)
    ij, conf, homo, cnt, pts, nconn = hip.ransac_pairwise_table(ctx, cfg, f, mh, pairs, shapes, base_seed=42)
    coors = [f.get(i)[1] for i in range(n)]
    host = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "openpano_amd", "libpano_host.so"))
    f64 = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
    host.pano_homography_inverse.argtypes = [f64, f64]
    e = at = 0
    for p, (i, j) in enumerate(pairs):
        r = res[p]
        if not r["ok"]:
            continue
        inl = r["inliers"]
        a = coors[i][lists[p][inl, 0]]; b = coors[j][lists[p][inl, 1]]
        inv = np.zeros(9); assert host.pano_homography_inverse(np.ascontiguousarray(r["homo"].reshape(9)), inv) == 1
        inv = inv * (1.0 / inv[8])
        assert ij[e].tolist() == [i, j] and ij[e + 1].tolist() == [j, i]
        assert conf[e] == np.float32(r["confidence"]) and conf[e + 1] == conf[e]
        assert np.array_equal(homo[e], r["homo"].reshape(9)) and np.array_equal(homo[e + 1], inv)
        assert cnt[e] == len(inl) and cnt[e + 1] == len(inl)
        assert np.array_equal(pts[at: at + len(inl)], np.concatenate([a, b], 1))
        assert np.array_equal(pts[at + len(inl): at + 2 * len(inl)], np.concatenate([b, a], 1))
        at += 2 * len(inl); e += 2
    assert e == len(ij) == 2 * nconn and at == len(pts) and nconn >= 4
    # the table is only defined for the pair list the result was computed for: a permuted list of the same length is refused
    with pytest.raises(hip.OpenPanoHipError, match="not the pair list"):
        hip.ransac_pairwise_table(ctx, cfg, f, mh, pairs, shapes, base_seed=42, table_pairs=pairs[::-1])
    mh.free(); f.free()


def test_joined_match_handles_give_the_single_call_results(ctx, cfg):
    """op_matches_concat: a pair list matched in two calls and joined == the list matched in one call -- the lists, and every
    pair's RANSAC result through ONE op_ransac_pairs call on the joined handle (what a rank of a sharded job does with the
    pairs it matched during the exchange and the rest: openpano_amd/distributed.py)."""
    from openpano_amd import hip
    n = 6
    views = synth.image_set(n, 400, 600, seed=31, overlap=0.45)
    f = hip.sift_batch(ctx, cfg, views)
    pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]
    shapes = [(600, 400)] * n
    seeds = [1000 + 7 * k for k in range(len(pairs))]
    whole = hip.match_pairs_handle(ctx, cfg, f, pairs)
    want_lists = whole.lists()
    want = hip.ransac_pairs(ctx, cfg, f, whole, pairs, shapes, seeds=seeds)
    cut = 4
    a = hip.match_pairs_handle(ctx, cfg, f, pairs[:cut]); b = hip.match_pairs_handle(ctx, cfg, f, pairs[cut:])
    for fetch_first in (False, True):            # with and without host mirrors on the two sides
        if fetch_first:
            a.lists(); b.lists()
        j = hip.Matches.concat(ctx, a, b)
        got_lists = j.lists()
        assert len(got_lists) == len(pairs) and all(np.array_equal(g, w) for g, w in zip(got_lists, want_lists))
        got = hip.ransac_pairs(ctx, cfg, f, j, pairs, shapes, seeds=seeds)
        for g, w in zip(got, want):
            assert g["ok"] == w["ok"] and g["best_hyp"] == w["best_hyp"] and g["best_count"] == w["best_count"]
            assert np.array_equal(g["homo"], w["homo"]) and np.array_equal(g["inliers"], w["inliers"]) and g["confidence"] == w["confidence"]
        j.free()
    assert sum(1 for w in want if w["ok"]) >= 3
    a.free(); b.free(); whole.free(); f.free()


def test_results_do_not_depend_on_the_host_pool_size(tmp_path):
    """The acceptance epilogue runs on the library's host pool (csrc/host_pool.hpp); which thread judges a pair, how many
    threads there are (OPENPANO_HOST_THREADS) and which ISA clone counts the keypoints change no bit of any pair's result:
    the same job in three processes -- the calling thread only, three threads, the default pool -- gives one digest."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, zlib, numpy as np; sys.path.insert(0, %r)\n"
            "from openpano_amd import hip, synth\nfrom openpano_amd.config import PanoConfig\n"
            "cfg = PanoConfig(); c = hip.Context(0)\n"
            "n = 8; views = synth.image_set(n, 400, 600, seed=22, overlap=0.45)\n"
            "f = hip.sift_batch(c, cfg, views); pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]\n"
            "mh = hip.match_pairs_handle(c, cfg, f, pairs)\n"
            "crc = 0\n"
            "for rep in range(3):\n"
            "    for r in hip.ransac_pairs(c, cfg, f, mh, pairs, [(600, 400)] * n, base_seed=11):\n"
            "        crc = zlib.crc32(np.asarray([r['ok'], r['best_hyp'], r['best_count']], np.int64).tobytes(), crc)\n"
            "        crc = zlib.crc32(np.float32(r['confidence']).tobytes(), crc)\n"
            "        crc = zlib.crc32(np.ascontiguousarray(r['homo'], np.float64).tobytes(), crc)\n"
            "        crc = zlib.crc32(np.ascontiguousarray(r['inliers'], np.int32).tobytes(), crc)\n"
            "print('DIGEST', crc, sum(1 for r in hip.ransac_pairs(c, cfg, f, mh, pairs, [(600, 400)] * n, base_seed=11) if r['ok']))\n") % root
    digests = []
    for threads in ("1", "3", None):
        env = dict(os.environ)
        env.pop("OPENPANO_HOST_THREADS", None)
        if threads:
            env["OPENPANO_HOST_THREADS"] = threads
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST")][-1].split()
        digests.append((line[1], line[2]))
    assert digests[0] == digests[1] == digests[2], digests
    assert int(digests[0][1]) >= 3          # overlapping neighbours of the strip are accepted: the gates' counts were exercised
