"""k_descriptor (openpano_amd/csrc/descriptor.hip) does not walk the reference's whole (2 radius + 1)^2 window
(feature/sift.cc:110-126): per window column it enumerates a row INTERVAL that is claimed to contain every sample the
reference's exact float tests keep (inside the circle, inside the rotated 4 x 4-bin square), and runs those exact
tests on the interval only.  A sample dropped by the interval would silently change descriptors, so the claim is
checked here on its own: the kernel's interval arithmetic restated in numpy fp32, against the exact tests, over
thousands of (scale, orientation) pairs and every sample of their windows."""
import numpy as np

F = np.float32


def _interval(fxx, radius, hist_w, cosort, sinort):
    """rows [ilo, ihi] of window column xx, before the image-bounds clamps (descriptor.hip: candidate enumeration)"""
    fr2 = F(radius) * F(radius)
    yc = np.floor((np.sqrt((fr2 - fxx * fxx).astype(F)).astype(F) + F(1e-3)).astype(F)).astype(F)
    ylo = -yc; yhi = yc.copy()
    m = (F(0.005) * hist_w + F(0.05)).astype(F)
    blo = (F(-2.5) * hist_w - m).astype(F); bhi = (F(1.5) * hist_w + m).astype(F)
    if abs(cosort) > F(0.01):
        rc = F(1) / cosort
        a = ((blo + fxx * sinort).astype(F) * rc).astype(F); b = ((bhi + fxx * sinort).astype(F) * rc).astype(F)
        ylo = np.maximum(ylo, np.minimum(a, b)); yhi = np.minimum(yhi, np.maximum(a, b))
    if abs(sinort) > F(0.01):
        rs = F(1) / sinort
        a = ((blo - fxx * cosort).astype(F) * rs).astype(F); b = ((bhi - fxx * cosort).astype(F) * rs).astype(F)
        ylo = np.maximum(ylo, np.minimum(a, b)); yhi = np.minimum(yhi, np.maximum(a, b))
    ilo = np.ceil(ylo).astype(np.int64); ihi = np.floor(yhi).astype(np.int64)
    return np.maximum(ilo, -radius), np.minimum(ihi, radius)


def _kept(xx, yy, radius, hist_w, cosort, sinort):
    """the reference's tests (sift.cc:113-126) in the kernel's arithmetic (descriptor.hip: phase 1a)"""
    fxx, fyy = xx.astype(F), yy.astype(F)
    fr2 = F(radius) * F(radius)
    inside = ~((fxx * fxx + fyy * fyy).astype(F) > fr2)
    rd = np.float64(1.0) / np.float64(hist_w)
    y_rot = (((-fxx) * sinort + fyy * cosort).astype(F).astype(np.float64) * rd).astype(F)
    x_rot = ((fxx * cosort + fyy * sinort).astype(F).astype(np.float64) * rd).astype(F)
    ybin = ((y_rot + F(2)) - F(0.5)).astype(F); xbin = ((x_rot + F(2)) - F(0.5)).astype(F)
    return inside & (ybin >= F(-1)) & (ybin <= F(3)) & (xbin >= F(-1)) & (xbin <= F(3))


def test_intervals_contain_every_kept_sample():
    rng = np.random.default_rng(17)
    angles = np.concatenate([rng.uniform(0, 2 * np.pi, 3000), np.arange(0, 64) * (np.pi / 32), np.arange(0, 64) * (np.pi / 32) + 1e-4,
                             np.arange(0, 64) * (np.pi / 32) - 1e-4,
                             [np.arccos(0.01), np.arcsin(0.01), np.arccos(0.0100001), np.arcsin(0.0099999), np.arccos(0.05), np.arcsin(0.05)]])
    kept_total = cand_total = 0
    for k, ort in enumerate(angles):
        sf = F(rng.uniform(1.2, 3.0))                    # keypoint scale factors: sigma 1.6 * 2^(s / 3) and below the 64-column window limit
        hist_w = F(sf * F(3.0))                          # DESC_HIST_SCALE_FACTOR
        radius = int(round(0.70710678118654752440 * float(hist_w) * 5))
        if 2 * radius + 1 > 64:
            continue                                     # such windows walk the whole window in the kernel
        cosort, sinort = F(np.cos(np.float64(F(ort)))), F(np.sin(np.float64(F(ort))))
        cols = np.arange(-radius, radius + 1)
        ilo, ihi = _interval(cols.astype(F), radius, hist_w, cosort, sinort)
        xx, yy = np.meshgrid(cols, cols, indexing="ij")
        keep = _kept(xx, yy, radius, hist_w, cosort, sinort)
        inside_interval = (yy >= ilo[:, None]) & (yy <= ihi[:, None])
        lost = keep & ~inside_interval
        assert not lost.any(), (float(ort), float(hist_w), radius, np.argwhere(lost)[:4])
        kept_total += int(keep.sum()); cand_total += int(inside_interval.sum())
    assert kept_total > 500_000
    assert cand_total < 1.05 * kept_total                  # and the intervals are tight enough to be worth having
