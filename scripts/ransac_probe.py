#!/usr/bin/env python3
"""GPU probe: where does op_ransac_pairs spend its device time?  Times the stage labels for several
RANSAC_ITERATIONS on the config-4 pair list (run on the GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from openpano_amd import hip, synth
from openpano_amd.config import PanoConfig

ctx = hip.Context(0)
cfg = PanoConfig()
views = synth.image_set(38, 867, 1300, seed=38, overlap=0.45, rows=2, shuffle=True)
f = hip.sift_batch(ctx, cfg, views)
pairs = [(i, j) for i in range(38) for j in range(i + 1, 38)]
mh = hip.match_pairs_handle(ctx, cfg, f, pairs)
ms = [len(x) for x in mh.lists()]
print("pairs with >= 8 matches:", sum(1 for m in ms if m >= 8), "of", len(ms), " match-count quantiles", np.percentile(ms, [50, 90, 99, 100]))
shapes = [(1300, 867)] * 38
for iters in (1, 100, 1500, 3000):
    c = PanoConfig(RANSAC_ITERATIONS=iters)
    hip.ransac_pairs_summary(ctx, c, f, mh, pairs, shapes, base_seed=1)
    ctx.set_profiling(True); ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(5):
        hip.ransac_pairs_summary(ctx, c, f, mh, pairs, shapes, base_seed=1)
    t = (time.perf_counter() - t0) / 5
    prof = {k: round(v[0] / 5, 4) for k, v in ctx.profile().items() if k.startswith("ransac")}
    ctx.set_profiling(False)
    print(iters, "wall ms", round(t * 1e3, 3), prof)
