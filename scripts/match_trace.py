#!/usr/bin/env python3
"""GPU probe: per-phase cycle split of the matcher sweep (variant library: scripts/build_variant.sh match9 match.hip -DOP_MATCH_EXPERIMENT=9, which applies scripts/experiments/match_timing_experiments.patch)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["OPENPANO_HIP_LIB"] = os.path.join(ROOT, "openpano_amd", "variants", "libopenpano_hip_%s.so" % os.environ.get("OPENPANO_TRACE_LIB", "match9"))
import numpy as np
from openpano_amd import hip, synth
from openpano_amd.config import PanoConfig
L = hip.lib(); L.op_debug_match_timers.argtypes = [C.c_void_p]; L.op_debug_match_occupancy.argtypes = [C.c_void_p]
ctx = hip.Context(0); cfg = PanoConfig()
def report(tag):
    t = (C.c_ulonglong * 10)(); L.op_debug_match_timers(t); t = list(t)
    n = max(t[5], 1); names = ["fetch issue", "LDS reads + MFMA", "top-4", "commit (vmcnt + LDS write)", "barrier"]
    tot = sum(t[:5])
    print(tag, "tiles", t[5], "workgroups", t[7], "cycles/tile %.0f" % (tot / n), {k: round(v / n) for k, v in zip(names, t[:5])}, "epilogue cycles/wg %.0f" % (t[6] / max(t[7], 1)))
rng = np.random.default_rng(1)
ctx.set_profiling(True)
for K, nimg in ((1200, 12), (4000, 8), (4000, 40)):
    sets = []
    for k in range(nimg):
        x = np.abs(rng.normal(0, 1, (K, 128))).astype(np.float32)
        sets.append((np.sqrt(x / x.sum(axis=1, keepdims=True)) * 512).astype(np.float32))
    f = hip.Features.from_host(ctx, sets)
    pairs = [(i, j) for i in range(nimg) for j in range(i + 1, nimg)]
    hip.match_pairs_handle(ctx, cfg, f, pairs).free(); report("warm")
    ctx.profile_reset()
    hip.match_pairs_handle(ctx, cfg, f, pairs).free()
    t = (C.c_ulonglong * 10)(); L.op_debug_match_timers(t); tot = sum(list(t)[:5]) + t[6]
    occ = (C.c_ulonglong * 8)(); L.op_debug_match_occupancy(occ); print('   residents per CU seen by a starting workgroup (1..7):', list(occ)[1:])
    prof = ctx.profile()
    ms = prof.get("matcher mfma forward", (0, 0))[0] + prof.get("matcher mfma reverse", (0, 0))[0]
    # every resident workgroup slot (3 per CU x 256 CUs) accumulates wave-0 cycles all the time the kernels run
    print(f"K={K} pairs={len(pairs)}: sweeps+slow {ms:.3f} ms, wave-0 cycles summed {tot:.3e}  ->  shader clock {t[9] / max(t[8], 1) * 0.1:.2f} GHz (clock64 / wall_clock64), busy workgroup slots {t[8] / 1e8 / (ms * 1e-3):.0f}", {k: round(v / max(t[5], 1)) for k, v in zip(["fetch", "mfma", "top4", "commit", "barrier"], list(t)[:5])})
    f.free()
