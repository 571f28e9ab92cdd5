#!/usr/bin/env python3
"""GPU probe: per-phase cycle split of the matcher sweep (variant library built with -DOP_MATCH_EXPERIMENT=9)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["OPENPANO_HIP_LIB"] = os.path.join(ROOT, "openpano_amd", "variants", "libopenpano_hip_match9.so")
import numpy as np
from openpano_amd import hip, synth
from openpano_amd.config import PanoConfig
L = hip.lib(); L.op_debug_match_timers.argtypes = [C.c_void_p]
ctx = hip.Context(0); cfg = PanoConfig()
def report(tag):
    t = (C.c_ulonglong * 8)(); L.op_debug_match_timers(t); t = list(t)
    n = max(t[5], 1); names = ["fetch issue", "LDS reads + MFMA", "top-4", "commit (vmcnt + LDS write)", "barrier"]
    tot = sum(t[:5])
    print(tag, "tiles", t[5], "workgroups", t[7], "cycles/tile %.0f" % (tot / n), {k: round(v / n) for k, v in zip(names, t[:5])}, "epilogue cycles/wg %.0f" % (t[6] / max(t[7], 1)))
rng = np.random.default_rng(1)
for K, nimg in ((1200, 12), (4000, 8)):
    sets = []
    for k in range(nimg):
        x = np.abs(rng.normal(0, 1, (K, 128))).astype(np.float32)
        sets.append((np.sqrt(x / x.sum(axis=1, keepdims=True)) * 512).astype(np.float32))
    f = hip.Features.from_host(ctx, sets)
    pairs = [(i, j) for i in range(nimg) for j in range(i + 1, nimg)]
    hip.match_pairs_handle(ctx, cfg, f, pairs).free(); report("warm")
    hip.match_pairs_handle(ctx, cfg, f, pairs).free(); report(f"K={K} pairs={len(pairs)} (fwd+rev)")
    f.free()
