#!/usr/bin/env python3
"""How far is the PARITY build of the reference (oracle/_ref/libopenpano_ref.so: -ffp-contract=off
-march=x86-64-v3) from the reference AS SHIPPED (libopenpano_ref_native.so: -O3 -march=native, GCC's
default contraction -- /root/reference/CMakeLists.txt:40)?  Same sources, same inputs; the shipped
flags let GCC fuse a*b+c on FMA hosts and vectorise wider, so thresholded results can flip.

Per view: keypoint-count delta, descriptors present in only one build (matched by coordinates),
max |descriptor delta| over common keypoints.  Per pair: match-set sizes and Jaccard index.
Usage: python scripts/ref_native_distance.py [--out profiles/r02_ref_native_distance.json] [--quick]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


TOL_PX = 0.02     # keypoints of the two builds are "the same" if their coordinates agree to 0.02 px


def _pairing(cp, cn):
    """greedy one-to-one pairing of two coordinate lists within TOL_PX (several orientations of one
    keypoint share coordinates: paired in list order) -> (idx_parity, idx_shipped) arrays"""
    from scipy.spatial import cKDTree
    if not len(cp) or not len(cn):
        return np.zeros(0, int), np.zeros(0, int)
    tree = cKDTree(cn)
    used = np.zeros(len(cn), bool)
    ia, ib = [], []
    for i, c in enumerate(cp):
        for j in sorted(tree.query_ball_point(c, TOL_PX)):
            if not used[j]:
                used[j] = True; ia.append(i); ib.append(j); break
    return np.array(ia, int), np.array(ib, int)


def compare_views(par, nat, views):
    rows, feats = [], []
    for name, v in views:
        dp, cp = par.detect_feature(v)
        dn, cn = nat.detect_feature(v)
        ia, ib = _pairing(cp, cn)
        dd = np.abs(dp[ia] - dn[ib]).max(axis=1) if len(ia) else np.zeros(0)
        rows.append(dict(view=name, k_parity=len(dp), k_shipped=len(dn), k_delta=len(dn) - len(dp),
                         only_parity=len(dp) - len(ia), only_shipped=len(dn) - len(ib),
                         max_abs_descriptor_delta=float(dd.max()) if len(dd) else 0.0,
                         descriptors_off_by_more_than_0p05=int((dd > 0.05).sum()),
                         max_coordinate_delta_px=float(np.abs(cp[ia] - cn[ib]).max()) if len(ia) else 0.0))
        feats.append((dp, cp, dn, cn, dict(zip(ia.tolist(), ib.tolist()))))
    return rows, feats


def compare_pairs(par, nat, names, feats, pairs):
    rows = []
    for i, j in pairs:
        mp = par.match_exact(feats[i][0], feats[j][0]); mn = nat.match_exact(feats[i][2], feats[j][2])
        # parity-build indices mapped onto the shipped build's keypoints (unpaired -> unique negative ids)
        mi, mj = feats[i][4], feats[j][4]
        sp = {(mi.get(int(a), -1 - int(a)), mj.get(int(b), -1 - int(b))) for a, b in mp}
        sn = {(int(a), int(b)) for a, b in mn}
        u = len(sp | sn)
        rows.append(dict(pair=[names[i], names[j]], matches_parity=len(mp), matches_shipped=len(mn),
                         only_parity=len(sp - sn), only_shipped=len(sn - sp), jaccard=(len(sp & sn) / u) if u else 1.0))
    return rows


def workload(quick):
    from openpano_amd import synth
    import natural
    views = []
    n2, n3, n4 = (2, 0, 2) if quick else (4, 3, 6)
    for k, v in enumerate(synth.image_set(11, 400, 600, seed=22, overlap=0.40, first=n2) if n2 else []):
        views.append((f"cfg2_synth_{k}", v))
    for k, v in enumerate(synth.image_set(13, 1112, 1500, seed=33, overlap=0.40, first=n3) if n3 else []):
        views.append((f"cfg3_synth_{k}", v))
    for k, v in enumerate(synth.image_set(4 if quick else 38, 867, 1300, seed=38, overlap=0.45, rows=2, first=n4) if n4 else []):   # grid order: neighbours overlap (quick: a 4-view world -- the 38-view one takes minutes to paint)
        views.append((f"cfg4_synth_{k}", v))
    if natural.available():
        for k, v in enumerate(natural.config_views(1)[: 1 if quick else None]):
            views.append((f"cfg1_nat_uav_{k}", natural.u8_to_f32(v)))
        for k, v in enumerate(natural.config_views(2, 2 if quick else 4)):
            views.append((f"cfg2_nat_cmu_{k}", natural.u8_to_f32(v)))
        for c in range(1 if quick else 4):
            views.append((f"cfg4_nat_uav_{c}", natural.u8_to_f32(natural.crop_u8("uav", 60, 60 + 140 * c, 867, 1300, seed=3800 + c))))
    return views


def run(quick=False):
    from checkers import Ref, REF_NATIVE_SO
    from openpano_amd.config import PanoConfig
    cfg = PanoConfig()
    par, nat = Ref(cfg), Ref(cfg, REF_NATIVE_SO)
    views = workload(quick)
    names = [n for n, _ in views]
    vrows, feats = compare_views(par, nat, views)
    pairs = [(i, i + 1) for i in range(len(views) - 1) if names[i].rsplit("_", 1)[0] == names[i + 1].rsplit("_", 1)[0]]
    prows = compare_pairs(par, nat, names, feats, pairs)
    summ = dict(views=len(vrows), total_k_parity=sum(r["k_parity"] for r in vrows), total_k_shipped=sum(r["k_shipped"] for r in vrows),
                views_with_count_delta=sum(1 for r in vrows if r["k_delta"] != 0),
                max_abs_count_delta=max(abs(r["k_delta"]) for r in vrows),
                keypoints_only_in_one_build=sum(r["only_parity"] + r["only_shipped"] for r in vrows),
                max_abs_descriptor_delta=max(r["max_abs_descriptor_delta"] for r in vrows),
                descriptors_off_by_more_than_0p05=sum(r["descriptors_off_by_more_than_0p05"] for r in vrows),
                max_coordinate_delta_px=max(r["max_coordinate_delta_px"] for r in vrows),
                pairs=len(prows), min_jaccard=min(r["jaccard"] for r in prows) if prows else None,
                pairs_with_different_match_count=sum(1 for r in prows if r["matches_parity"] != r["matches_shipped"]))
    return dict(parity_build="-O3 -march=x86-64-v3 -ffp-contract=off (oracle/Makefile)",
                shipped_build="-O3 -march=native, default -ffp-contract=fast (reference CMakeLists.txt:40)",
                summary=summ, views=vrows, pairs=prows)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_ref_native_distance.json"))
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    res = run(a.quick)
    json.dump(res, open(a.out, "w"), indent=1)
    print(json.dumps(res["summary"], indent=1))
    for r in res["views"]:
        print(f'{r["view"]:18s} K {r["k_parity"]:5d} -> {r["k_shipped"]:5d}  only-one-build {r["only_parity"]}+{r["only_shipped"]}  max|d desc| {r["max_abs_descriptor_delta"]:.4g} (>{0.05}: {r["descriptors_off_by_more_than_0p05"]})  max|d coor| {r["max_coordinate_delta_px"]:.3g} px')
    for r in res["pairs"]:
        print(f'{r["pair"][0]:18s} x {r["pair"][1]:18s} matches {r["matches_parity"]:4d} / {r["matches_shipped"]:4d}  only-one-build {r["only_parity"]}+{r["only_shipped"]}  jaccard {r["jaccard"]:.4f}')
