"""GPU parity tests of the MFMA matcher (C-ABI op_match_pairs) against the exact-matcher oracle
(FeatureMatcher::match restated, oracle/match_oracle.c).  Match sets must be IDENTICAL."""
import os

import numpy as np
import pytest

from openpano_amd import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def ctx():
    from openpano_amd import hip
    c = hip.Context(0)
    yield c
    c.close()


def test_match_golden(ctx, cfg):
    from openpano_amd import hip
    a = np.load(os.path.join(HERE, "golden", "sift_a_240x320.npz"))["desc"]
    b = np.load(os.path.join(HERE, "golden", "sift_b_240x320.npz"))["desc"]
    want = np.load(os.path.join(HERE, "golden", "match_ab.npz"))["pairs"]
    f = hip.Features.from_host(ctx, [a, b])
    got = hip.match_pairs(ctx, cfg, f, [(0, 1), (1, 0)])
    assert np.array_equal(got[0], want)
    assert sorted(map(tuple, got[1][:, ::-1])) == sorted(map(tuple, want))
    f.free()


def test_all_pairs_on_sift_features(ctx, oracle, cfg):
    """SIFT on device -> all-pairs match on device == oracle exact matcher on the same descriptors"""
    from openpano_amd import hip
    views = synth.image_set(5, 400, 600, seed=22, overlap=0.45)
    f = hip.sift_batch(ctx, cfg, views)
    descs = [f.get(i)[0] for i in range(5)]
    pairs = [(i, j) for i in range(5) for j in range(i + 1, 5)]
    got = hip.match_pairs(ctx, cfg, f, pairs)
    nmatch = 0
    for (i, j), g in zip(pairs, got):
        want = oracle.match_exact(descs[i], descs[j])
        assert np.array_equal(g, want), (i, j, len(g), len(want))
        nmatch += len(want)
    assert nmatch > 100
    f.free()


def test_edge_cases(ctx, oracle, cfg):
    from openpano_amd import hip
    rng = np.random.default_rng(1)
    a = np.load(os.path.join(HERE, "golden", "sift_a_240x320.npz"))["desc"]
    sets = [
        a[:0],                      # 0: empty
        a[:1],                      # 1: single descriptor
        a[:3],                      # 2: fewer than the top-4 buffer
        a[:129],                    # 3: one row past a 128-row block
        a[:200].copy(),             # 4
        np.concatenate([a[:100], a[:100]]),        # 5: exact duplicates -> ties, nothing distinctive
        (rng.random((300, 128)) * 40).astype(np.float32),   # 6: un-normalised random vectors
        (a[:200] + rng.normal(0, 3.0, (200, 128)).astype(np.float32)).clip(0).astype(np.float32),  # 7: noisy copy of 4
        np.zeros((5, 128), np.float32),             # 8: all-zero descriptors
        # 9/10: more than 8 identical descriptors: every ranked candidate falls inside the error
        # margin, so the rows take the exact full-scan path (forward and reverse)
        np.concatenate([np.repeat(a[5:6], 12, axis=0), a[:60], np.repeat(a[70:71], 9, axis=0)]),
        np.concatenate([a[:40], np.repeat(a[5:6], 11, axis=0), a[60:90]]),
        # 11: a NaN descriptor (zero-weight window) never matches and never wins a ratio test
        np.concatenate([a[:30], np.full((1, 128), np.nan, np.float32), a[30:50]]),
    ]
    f = hip.Features.from_host(ctx, sets)
    pairs = [(i, j) for i in range(len(sets)) for j in range(len(sets)) if i != j]
    got = hip.match_pairs(ctx, cfg, f, pairs)
    for (i, j), g in zip(pairs, got):
        want = oracle.match_exact(sets[i], sets[j])
        assert np.array_equal(g, want), (i, j, g.tolist()[:5], want.tolist()[:5])
    # a set against itself: every descriptor matches itself when all are distinct
    g = hip.match_pairs(ctx, cfg, f, [(4, 4)])[0]
    assert np.array_equal(g, oracle.match_exact(sets[4], sets[4]))
    f.free()


def test_full_size_properties(ctx, oracle, cfg):
    """config-4 sized descriptor sets: symmetry + spot oracle checks"""
    from openpano_amd import hip
    views = synth.image_set(6, 867, 1300, seed=38, overlap=0.45, rows=2)
    f = hip.sift_batch(ctx, cfg, views)
    pairs = [(i, j) for i in range(6) for j in range(6) if i != j]
    got = dict(zip(pairs, hip.match_pairs(ctx, cfg, f, pairs)))
    for i in range(6):
        for j in range(i + 1, 6):
            # match(i,j) and match(j,i) are the same set with swapped columns
            assert sorted(map(tuple, got[(i, j)])) == sorted(map(tuple, got[(j, i)][:, ::-1]))
            # one-to-one: an index appears at most once on either side
            if len(got[(i, j)]):
                assert len(set(got[(i, j)][:, 0])) == len(got[(i, j)]) == len(set(got[(i, j)][:, 1]))
    for (i, j) in [(0, 1), (2, 5), (4, 3)]:
        want = oracle.match_exact(f.get(i)[0], f.get(j)[0])
        assert np.array_equal(got[(i, j)], want)
    f.free()


def test_config5_sized_sets(ctx, oracle, cfg):
    """BASELINE config 5 scale for the matcher: K ~ 4000 descriptors per image (the MFMA
    match-matrix stress).  Properties over all pairs of 6 sets + exact oracle check of two pairs."""
    from openpano_amd import hip
    rng = np.random.default_rng(5)
    a = np.load(os.path.join(HERE, "golden", "sift_d_500x700.npz"))["desc"]
    base = np.concatenate([a] * (4000 // len(a) + 1))[:4000]
    sets = []
    for k in range(6):
        # RootSIFT-like rows: a noisy, shuffled copy of a common base so that true matches exist
        x = np.abs(base + rng.normal(0, 6.0 + 2 * k, base.shape)).astype(np.float32)
        x = (np.sqrt(x / x.sum(axis=1, keepdims=True)) * 512).astype(np.float32)
        sets.append(x[rng.permutation(len(x))[: 3600 + 80 * k]])
    f = hip.Features.from_host(ctx, sets)
    pairs = [(i, j) for i in range(6) for j in range(6) if i != j]
    got = dict(zip(pairs, hip.match_pairs(ctx, cfg, f, pairs)))
    for i in range(6):
        for j in range(i + 1, 6):
            assert sorted(map(tuple, got[(i, j)])) == sorted(map(tuple, got[(j, i)][:, ::-1]))
            if len(got[(i, j)]):
                assert len(set(got[(i, j)][:, 0])) == len(got[(i, j)]) == len(set(got[(i, j)][:, 1]))
    for (i, j) in [(0, 1), (5, 2)]:
        assert np.array_equal(got[(i, j)], oracle.match_exact(sets[i], sets[j]))
    f.free()
