#!/usr/bin/env python3
"""A/B of op_ransac_pairs across builds of libopenpano_hip.so in ONE process on the same match lists (GPU box).

    python scripts/ransac_ab.py [--steps N] lib_a.so ...         ("product" is always first)

The config-4 job (38 synthetic 1300x867 views, 703 pairs) and one rank's share of it at 8 ranks (88 pairs): wall time per
call with events off, the stage table with events on, and a digest of every pair's result (ok, confidence, homography,
inlier list, winner) -- builds that disagree show at once."""
import argparse, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("libs", nargs="*")
    a = ap.parse_args()
    import numpy as np
    from openpano_amd import hip, synth
    from openpano_amd.config import PanoConfig
    cfg = PanoConfig()
    views = synth.image_set(38, 867, 1300, seed=38, overlap=0.45, rows=2, shuffle=True)
    product = hip.LIB_PATH
    for name in ["product"] + list(a.libs):
        path = product if name == "product" else os.path.abspath(name)
        hip._lib = None; hip.LIB_PATH = path
        ctx = hip.Context(0)
        f = hip.sift_batch(ctx, cfg, views)
        allp = [(i, j) for i in range(38) for j in range(i + 1, 38)]
        shapes = [(1300, 867)] * 38
        short = os.path.basename(path).replace("libopenpano_hip_", "").replace(".so", "")
        for label, pairs in (("703 pairs", allp), ("88 pairs", allp[::8])):
            mh = hip.match_pairs_handle(ctx, cfg, f, pairs)
            res = hip.ransac_pairs(ctx, cfg, f, mh, pairs, shapes, base_seed=7)
            crc = 0
            for r in res:
                crc = zlib.crc32(np.asarray([r["ok"], r["best_hyp"], r["best_count"]], np.int64).tobytes(), crc)
                crc = zlib.crc32(np.float32(r["confidence"]).tobytes(), crc)
                crc = zlib.crc32(np.ascontiguousarray(r["homo"], np.float64).tobytes(), crc)
                crc = zlib.crc32(np.ascontiguousarray(r["inliers"], np.int32).tobytes(), crc)
            best = None
            for rep in range(3):
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    hip.ransac_pairs_summary(ctx, cfg, f, mh, pairs, shapes, base_seed=7)
                t = (time.perf_counter() - t0) / a.steps * 1e3
                best = t if best is None else min(best, t)
            ctx.set_profiling(True); ctx.profile_reset()
            for _ in range(a.steps):
                hip.ransac_pairs_summary(ctx, cfg, f, mh, pairs, shapes, base_seed=7)
            prof = {k.replace("ransac ", ""): round(v[0] / a.steps, 4) for k, v in ctx.profile().items() if k.startswith("ransac")}
            ctx.set_profiling(False)
            print(f"{short:12s} {label:10s} call {best:.3f} ms  accepted {sum(1 for r in res if r['ok'])}  crc {crc:08x}  {prof}", flush=True)
            mh.free()
        f.free(); ctx.close()


if __name__ == "__main__":
    main()
