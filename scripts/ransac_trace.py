#!/usr/bin/env python3
"""GPU probe: per-phase cycle split of k_ransac_samples (variant library: scripts/build_variant.sh rs9 ransac.hip
-DOP_RANSAC_EXPERIMENT=9, which applies scripts/experiments/ransac_timing_experiments.patch) on the config-4 pair list."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["OPENPANO_HIP_LIB"] = os.path.join(ROOT, "ab", "libopenpano_hip_rs9.so")
import numpy as np
from openpano_amd import hip, synth
from openpano_amd.config import PanoConfig
L = hip.lib(); L.op_debug_rs_timers.argtypes = [C.c_void_p]
ctx = hip.Context(0); cfg = PanoConfig()
views = synth.image_set(38, 867, 1300, seed=38, overlap=0.45, rows=2, shuffle=True)
f = hip.sift_batch(ctx, cfg, views)
pairs = [(i, j) for i in range(38) for j in range(i + 1, 38)]
mh = hip.match_pairs_handle(ctx, cfg, f, pairs)
ms = np.array([len(x) for x in mh.lists()])
print("pairs", len(ms), "live (>= 8 matches)", int((ms >= 8).sum()), "with more than 64 matches", int((ms > 64).sum()), "quantiles", np.percentile(ms, [10, 50, 90, 99, 100]))
shapes = [(1300, 867)] * 38
buf = (C.c_ulonglong * 8)()
hip.ransac_pairs_summary(ctx, cfg, f, mh, pairs, shapes, base_seed=1); L.op_debug_rs_timers(buf)
ctx.set_profiling(True); ctx.profile_reset()
hip.ransac_pairs_summary(ctx, cfg, f, mh, pairs, shapes, base_seed=1)
prof = {k: round(v[0], 4) for k, v in ctx.profile().items() if k.startswith("ransac")}
L.op_debug_rs_timers(buf); t = list(buf)
wg = max(t[5], 1)
names = ["twist + temper", "next(i) table", "pointer jumping", "count + emit", "carry"]
print("workgroups", t[5], "chunks per workgroup %.2f" % (t[6] / wg), "lifetime cycles per workgroup %.0f" % (t[7] / wg))
print("cycles per workgroup:", {n: round(t[k] / wg) for k, n in enumerate(names)}, "sum", round(sum(t[:5]) / wg))
print("stage ms (with the trace's own barriers and clock reads in the kernel):", prof)
