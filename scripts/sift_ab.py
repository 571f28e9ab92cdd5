#!/usr/bin/env python3
"""A/B timing of SIFT-stage kernels across builds of libopenpano_hip.so (timing experiments; GPU box).

    python scripts/sift_ab.py [--steps N] lib_a.so lib_b.so ...      ("product" = openpano_amd/libopenpano_hip.so)

Every library runs the bench's config-4 SIFT step (38 synthetic 1300x867 views resident in HBM) in ONE process on the
same inputs: per-stage HIP-event times, the step's wall time and a CRC of all descriptors + coordinates, so that a
variant that changes results is visible at once (timing experiments that break results on purpose say so in their name).
"""
import argparse
import json
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--images", type=int, default=38)
    ap.add_argument("--json", default=None)
    ap.add_argument("--config5", action="store_true", help="BASELINE config 5 instead: 128 (or --images) synthetic 4000x3000 uint8 images")
    ap.add_argument("--no-product", action="store_true", help="only the named libraries (e.g. under a profiler)")
    ap.add_argument("libs", nargs="*")
    a = ap.parse_args()
    import numpy as np
    import torch
    from openpano_amd import hip, synth
    from openpano_amd.config import PanoConfig
    cfg = PanoConfig()
    dev = torch.device("cuda", 0)
    if a.config5:
        H, W = 3000, 4000
        d_imgs = synth.config5_views(range(a.images if a.images != 38 else 128), dev)
        inputs = [(t.data_ptr(), H, W, "u8") for t in d_imgs]
    else:
        H, W = 867, 1300
        views = synth.image_set(a.images, H, W, seed=38, overlap=0.45, rows=2, shuffle=True)
        d_imgs = [torch.from_numpy(v).to(dev) for v in views]
        inputs = [(t.data_ptr(), H, W) for t in d_imgs]
    torch.cuda.synchronize()
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    libs = ([] if a.no_product else ["product"]) + list(a.libs)
    out = {}
    for name in libs:
        path = os.path.join(ROOT, "openpano_amd", "libopenpano_hip.so") if name == "product" else os.path.abspath(name)
        hip._lib = None
        hip.LIB_PATH = path
        ctx = hip.Context(0, stream.cuda_stream)
        call = hip.SiftCall(ctx, cfg, inputs)
        f = None
        for _ in range(3):
            if f is not None:
                f.free()
            f = call()
        best = None
        for rep in range(3):                       # best of 3 timed loops (clock ramp, neighbours on the box)
            ctx.set_profiling(True); ctx.profile_reset()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(a.steps):
                f.free(); f = call()
            torch.cuda.synchronize(); t = (time.perf_counter() - t0) / a.steps * 1e3
            prof = {k: v[0] / a.steps for k, v in ctx.profile().items()}
            ctx.set_profiling(False)
            if best is None or t < best[0]:
                best = (t, prof)
        L = hip.lib()
        if hasattr(L, "op_debug_pyr_timers"):          # trace build (-DOP_PYR_EXPERIMENT=9): cycles of wave 0 per phase, per workgroup step
            import ctypes as C
            buf = (C.c_ulonglong * 12)()
            L.op_debug_pyr_timers(buf)
            f.free(); f = call(); torch.cuda.synchronize()
            L.op_debug_pyr_timers(buf)
            v = list(buf); wg = max(v[9], 1); st = max(v[10], 1)
            names = ["column pass", "sV write + barrier 1", "row pass + DoG", "ring write + stores", "gate + queue", "barrier 2", "scan + slide"]
            print("  pyramid trace: workgroups %d, steps/wg %.2f, lifetime %.0f cycles, prologue %.0f" % (wg, st / wg, v[8] / wg, v[7] / wg))
            print("  per step: " + ", ".join("%s %.0f" % (n, v[k] / st) for k, n in enumerate(names)) + "  = %.0f" % (sum(v[:7]) / st))
        crc = 0
        for i in range(f.num_images):
            d, c = f.get(i)
            crc = zlib.crc32(c.tobytes(), zlib.crc32(d.tobytes(), crc))
        k = int(f.total)
        f.free(); ctx.close()
        short = os.path.basename(path).replace("libopenpano_hip_", "").replace(".so", "")
        out[short] = {"ms_per_step": round(best[0], 4), "descriptors": k, "crc32": crc, "stage_ms": {kk: round(v, 4) for kk, v in best[1].items()}}
        print(f"{short:28s} step {best[0]:.4f} ms  K {k}  crc {crc:08x}  " + "  ".join(f"{kk.split()[0]} {v:.4f}" for kk, v in best[1].items() if not kk.endswith("(host)")), flush=True)
    if a.json:
        json.dump(out, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
