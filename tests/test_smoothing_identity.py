"""The orientation histogram smoothing of k_orientation (openpano_amd/csrc/keypoints.hip) replaces the reference's
double arithmetic   hist[i] = (float)((double)hist[i] * 0.5 + (double)(prev + next) * 0.25)   (feature/orientation.cc:70-75)
by ONE fp32 fma   fmaf(prev + next, 0.25f, hist[i] * 0.5f)   whenever no bin lies in (0, 2^-100).  This test checks that
identity with exact rational arithmetic: the fma is emulated as "exact value, one rounding to fp32" on fractions, the
reference side is numpy's float64 / float32 arithmetic (IEEE, like the CPU the reference runs on)."""
from fractions import Fraction

import numpy as np


def _round_f32(x: Fraction) -> np.float32:
    """round-to-nearest-even of a non-negative rational to fp32 (normal range only)"""
    if x == 0:
        return np.float32(0.0)
    assert x > 0
    e = x.numerator.bit_length() - x.denominator.bit_length() - 24      # x / 2^e is near 2^24
    while x / Fraction(2) ** e >= 1 << 24:
        e += 1
    while x / Fraction(2) ** e < 1 << 23:
        e -= 1
    assert -149 < e < 104, "outside the normal range this test covers"
    scaled = x / Fraction(2) ** e
    q = scaled.numerator // scaled.denominator
    rem = scaled - q
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and q % 2 == 1):
        q += 1
    return np.float32(np.ldexp(np.float64(q), e))


def test_fma_form_equals_the_double_form():
    rng = np.random.default_rng(7)
    n = 30000
    # magnitudes over a wide range of exponents (gaps beyond 29 bits between the terms included), zeros, equal values,
    # values just above the guard
    def draw():
        m = rng.uniform(1.0, 2.0, n).astype(np.float32)
        e = rng.integers(-95, 60, n)
        v = np.ldexp(m, e).astype(np.float32)
        v[rng.random(n) < 0.1] = 0.0
        return v
    h, p, q = draw(), draw(), draw()
    k = rng.random(n) < 0.2
    p[k] = h[k]; q[k] = h[k]                                     # flat neighbourhoods
    k = rng.random(n) < 0.1
    h[k] = np.float32(2.0) ** -100                               # the guard's edge
    s = (p + q).astype(np.float32)                               # (prev + next) in float, as the reference writes it
    ref = (h.astype(np.float64) * 0.5 + s.astype(np.float64) * 0.25).astype(np.float32)
    half = (h * np.float32(0.5)).astype(np.float32)
    assert np.array_equal(half.astype(np.float64), h.astype(np.float64) * 0.5)       # h * 0.5f is exact above the guard
    bad = []
    for i in range(n):
        fast = _round_f32(Fraction(float(s[i])) / 4 + Fraction(float(half[i])))
        if fast.tobytes() != ref[i].tobytes():
            bad.append((float(h[i]), float(p[i]), float(q[i]), float(fast), float(ref[i])))
    assert not bad, bad[:5]


def test_lockstep_recurrence_equals_the_sequential_walk():
    """bin b in lane b, all lanes recompute from the left neighbour's latest value: after step t bins 0..t are final"""
    rng = np.random.default_rng(3)
    for _ in range(200):
        hist = (rng.random(36) * rng.choice([0.0, 1.0, 1e-3, 1e3], 36)).astype(np.float32)
        seq = hist.copy()
        for _k in range(2):
            for i in range(36):
                prev = seq[35 if i == 0 else i - 1]; nxt = seq[0 if i == 35 else i + 1]
                seq[i] = np.float32(np.float64(seq[i]) * 0.5 + np.float64(np.float32(prev + nxt)) * 0.25)
        hv = hist.copy()
        for _k in range(2):
            half = (hv * np.float32(0.5)).astype(np.float32)
            nxt = np.roll(hv, -1).copy()                        # old hist[b + 1]
            pv0 = hv[35]                                         # lane 0 keeps the old hist[35]
            cur = hv.copy()
            for t in range(36):
                pv = np.concatenate(([pv0], cur[:-1])).astype(np.float32)
                cur = (half.astype(np.float64) + (pv + nxt).astype(np.float32).astype(np.float64) * 0.25).astype(np.float32)
                if t == 0:
                    nxt[35] = cur[0]
            hv = cur
        assert np.array_equal(hv, seq)
