"""bench.py's `configs` blocks: BASELINE.json's other single-GPU configurations, driver-timed in the same invocation.

  2          11 ordered 600x400 views, ESTIMATE_CAMERA      (README.md:123-127: 3.2 s on the author's i7-6700HQ, whole CLI)
  3          13 ordered 1500x1112 views, ESTIMATE_CAMERA    (README.md: 6 s)
  4_natural  38 unordered 1300x867 views, ESTIMATE_CAMERA   (README.md: 51 s) -- SURVEY 8(d) row 4: natural-texture crops

All three are natural-texture crops of the reference's published panoramas (tests/natural.py; no example data can be
downloaded).  Per block: SIFT keypoints+descriptors/s over the resident images (the headline metric on that workload),
the match + RANSAC loops over the configuration's pair list (ORDERED_INPUT: linear_pairwise_match's (i, i+1 mod n),
stitcher.cc:116-136; else all pairs, :96-113), the whole ESTIMATE_CAMERA pipeline (bench_e2e.run_pipeline) and -- outside
every timed region -- parity of what the timed loops left in HBM against the CPU oracle.
"""
import os
import time
import zlib

import numpy as np
import torch

PUBLISHED = {"2": 3.2, "3": 6.0, "4_natural": 51.0}    # seconds, whole CLI run, Intel Core i7-6700HQ (README.md:123-127)


def _parity(hip, ctx, cfg, orc, views, feats, mh, pairs, shapes, seeds, rres, max_pairs=96):
    from concurrent.futures import ThreadPoolExecutor
    nt = min(64, os.cpu_count() or 1)
    n = len(views)
    with ThreadPoolExecutor(nt) as ex:
        want = list(ex.map(lambda i: orc.detect_feature(views[i]), range(n)))
    got = [feats.get(i) for i in range(n)]
    bad_img = [i for i in range(n) if not (np.array_equal(got[i][0], want[i][0]) and np.array_equal(got[i][1], want[i][1]))]
    sel = list(range(len(pairs)))[:max_pairs]
    lists = mh.lists()
    with ThreadPoolExecutor(nt) as ex:
        wantm = list(ex.map(lambda k: orc.match_exact(want[pairs[k][0]][0], want[pairs[k][1]][0]), sel))
    bad_match = [list(pairs[k]) for k, w in zip(sel, wantm) if not np.array_equal(lists[k], w)]

    def one(k):
        i, j = pairs[k]
        return orc.ransac(lists[k], want[i][1], want[j][1], shapes[i], shapes[j], seeds[k])
    with ThreadPoolExecutor(nt) as ex:
        wr = list(ex.map(one, sel))
    bad_ransac = []
    for k, w in zip(sel, wr):
        g = rres[k]
        if not (g["best_hyp"] == w["best_hyp"] and g["best_count"] == w["best_count"] and g["ok"] == w["ok"] and g["confidence"] == w["confidence"]
                and np.array_equal(g["inliers"], w["inliers"]) and (not w["ok"] or np.array_equal(g["homo"], w["homo"]))):
            bad_ransac.append(list(pairs[k]))
    return {"checked": True, "images": n, "descriptors": int(sum(len(w[0]) for w in want)), "images_differing": bad_img,
            "descriptor_crc32": zlib.crc32(b"".join(np.ascontiguousarray(g[0]).tobytes() for g in got)),
            "oracle_descriptor_crc32": zlib.crc32(b"".join(np.ascontiguousarray(w[0]).tobytes() for w in want)),
            "pairs": len(sel), "matches": int(sum(len(w) for w in wantm)), "pairs_differing": bad_match,
            "ransac_accepted": int(sum(1 for w in wr if w["ok"])), "ransac_pairs_differing": bad_ransac,
            "ok": not bad_img and not bad_match and not bad_ransac}


def run_config(hip, ctx, key, args, dev, log, parity=True):
    """-> the `configs[key]` block; key in ("2", "3", "4_natural")"""
    import natural
    from openpano_amd.config import PanoConfig
    from bench_e2e import run_pipeline
    k = int(key[0])
    ordered = k in (2, 3)
    cfg = PanoConfig(ESTIMATE_CAMERA=1, ORDERED_INPUT=1 if ordered else 0, TRANS=0)
    t0 = time.perf_counter()
    views = [natural.u8_to_f32(v) for v in natural.config_views(k)]          # Mat32f as read_img makes them (lib/imgio.cc:43-60)
    n = len(views)
    H, W = views[0].shape[:2]
    log(f"config {key}: {n} natural-texture views {W}x{H} cut in {time.perf_counter() - t0:.1f} s")
    d_imgs = [torch.from_numpy(v).to(dev) for v in views]
    inputs = [(t.data_ptr(), H, W) for t in d_imgs]
    pairs = [(i, (i + 1) % n) for i in range(n)] if ordered else [(i, j) for i in range(n) for j in range(i + 1, n)]
    shapes = [(W, H)] * n
    # ---- SIFT: the headline metric on this workload (inputs resident in HBM)
    call = hip.SiftCall(ctx, cfg, inputs)
    feats = None
    for _ in range(max(5, min(args.warmup, 20))):
        if feats is not None:
            feats.free()
        feats = call()
    steps = max(1, args.steps if key == "4_natural" else min(args.steps, 100))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        feats.free(); feats = call()
    torch.cuda.synchronize(); t_sift = time.perf_counter() - t0
    K = int(feats.total)
    ctx.set_profiling(True); ctx.profile_reset()
    for _ in range(5):
        feats.free(); feats = call()
    stage = {kk: round(v[0] / 5, 4) for kk, v in ctx.profile().items()}
    ctx.set_profiling(False)
    out = {"workload": f"BASELINE config {k}: {n} {'ordered' if ordered else 'unordered'} {W}x{H} views, ESTIMATE_CAMERA; natural-texture crops of the reference's "
                       f"published panoramas (tests/natural.py config_views({k})), fp32 Mat32f resident in HBM",
           "images": n, "image": [H, W], "descriptors": K, "keypoints_per_image": K / n,
           "keypoints_per_s": K * steps / t_sift, "sift_ms_per_step": t_sift / steps * 1e3, "sift_steps": steps, "sift_stage_ms": stage,
           "published_cpu_seconds_whole_cli": PUBLISHED[key], "published_on": "Intel Core i7-6700HQ, /root/reference README.md:123-127 (context, not a baseline of this box)"}
    # ---- match + RANSAC over the configuration's pair list
    mh = hip.match_pairs_handle(ctx, cfg, feats, pairs)                       # warm-up + the lists RANSAC and the parity use
    msteps = max(1, min(args.steps, 20))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(msteps):
        hip.match_pairs_handle(ctx, cfg, feats, pairs).free()
    torch.cuda.synchronize(); tm = time.perf_counter() - t0
    seeds = [(1 + i * n + j) & 0xFFFFFFFF for i, j in pairs]
    rres = hip.ransac_pairs(ctx, cfg, feats, mh, pairs, shapes, seeds=seeds)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(msteps):
        hip.ransac_pairs_summary(ctx, cfg, feats, mh, pairs, shapes, seeds=seeds)
    torch.cuda.synchronize(); tr = time.perf_counter() - t0
    nm = mh.total
    out.update({"pair_list": "linear_pairwise_match: (i, i + 1 mod n)" if ordered else "pairwise_match: all unordered pairs",
                "image_pairs": len(pairs), "matches": nm, "match_ms_per_call": tm / msteps * 1e3, "image_pairs_per_s": len(pairs) * msteps / tm,
                "matches_per_s": nm * msteps / tm, "ransac_ms_per_call": tr / msteps * 1e3, "ransac_image_pairs_per_s": len(pairs) * msteps / tr,
                "ransac_accepted_pairs": int(sum(1 for r in rres if r["ok"]))})
    if parity:
        from checkers import Oracle
        t0 = time.perf_counter()
        out["parity"] = _parity(hip, ctx, cfg, Oracle(cfg), views, feats, mh, pairs, shapes, seeds, rres)
        log(f"config {key}: parity against the oracle took {time.perf_counter() - t0:.1f} s -> ok={out['parity']['ok']}")
    mh.free(); feats.free()
    del d_imgs
    # ---- the whole pipeline: SIFT -> match -> RANSAC -> host camera estimation + bundle adjustment -> spherical linear blend
    if not args.no_e2e:
        out["stitch_e2e"] = run_pipeline(hip, ctx, cfg, views, pairs, log)
    torch.cuda.empty_cache()
    return out
