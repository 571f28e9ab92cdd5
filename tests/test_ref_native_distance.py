"""CPU: distance between the reference's PARITY build (-ffp-contract=off, what every bit-exact
claim in this repo is pinned to) and the reference AS SHIPPED (-O3 -march=native, GCC's default
contraction; /root/reference/CMakeLists.txt:40), same sources, same inputs.  The two are different
roundings of one algorithm; this test measures how different and bounds it (the table for DESIGN.md
section 1.1 is written by scripts/ref_native_distance.py into profiles/)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def test_shipped_flags_build_stays_within_a_handful_of_flips():
    from checkers import ref_available, ref_native_usable
    if not (ref_available() and ref_native_usable()):
        pytest.skip("oracle/_ref builds absent or -march=native build not runnable on this host")
    import ref_native_distance as rnd
    res = rnd.run(quick=True)
    s = res["summary"]
    print(s)
    assert s["views"] >= 6
    # north_star: "identical keypoint counts": a thresholded float compare may flip under FMA --
    # at most a handful per image (extrema.cc:82,94,166,179 are the deciding comparisons)
    assert s["max_abs_count_delta"] <= 3, res["views"]
    assert s["keypoints_only_in_one_build"] <= 0.005 * s["total_k_parity"]
    assert s["max_coordinate_delta_px"] < 1e-2
    # descriptors: same up to rounding except for rare trilinear-bin / orientation-bin flips
    assert s["descriptors_off_by_more_than_0p05"] <= 0.003 * s["total_k_parity"]
    assert s["min_jaccard"] >= 0.97
