#!/usr/bin/env python3
"""A/B timing of the matcher across builds of libopenpano_hip.so in ONE process on the same features (GPU box).

    python scripts/match_ab.py [--c5-images 32] lib_a.so lib_b.so ...     ("product" is always first)

Config 4 (38 x 1300x867 synthetic views, 703 pairs) and a config-5-shaped job (--c5-images of the 128 4000x3000 uint8
images, all pairs, K ~ 4 k): per-stage HIP-event times of op_match_pairs, the call's wall time, exact-scan pressure is
visible in the forward / reverse stages (they include the exact-scan kernels), and a digest of all match lists.
Boxes differ by tens of percent on this power-bound kernel: only numbers of one run compare."""
import argparse
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--c5-images", type=int, default=32)
    ap.add_argument("libs", nargs="*")
    a = ap.parse_args()
    import numpy as np
    import torch
    from openpano_amd import hip, synth
    from openpano_amd.config import PanoConfig
    cfg = PanoConfig()
    dev = torch.device("cuda", 0)
    views = synth.image_set(38, 867, 1300, seed=38, overlap=0.45, rows=2, shuffle=True)
    d4 = [torch.from_numpy(v).to(dev) for v in views]
    d5 = synth.config5_views(range(a.c5_images), dev) if a.c5_images else []
    torch.cuda.synchronize()
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    jobs = [("config4", [(t.data_ptr(), 867, 1300) for t in d4])]
    if d5:
        jobs.append((f"config5[{a.c5_images}]", [(t.data_ptr(), 3000, 4000, "u8") for t in d5]))
    product = hip.LIB_PATH                       # openpano_amd/libopenpano_hip.so, or OPENPANO_HIP_LIB (one build per process: profilers)
    for name in ["product"] + list(a.libs):
        path = product if name == "product" else os.path.abspath(name)
        hip._lib = None
        hip.LIB_PATH = path
        ctx = hip.Context(0, stream.cuda_stream)
        short = os.path.basename(path).replace("libopenpano_hip_", "").replace(".so", "")
        for jname, inputs in jobs:
            f = hip.sift_batch(ctx, cfg, inputs)
            n = f.num_images
            pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]
            mh = hip.match_pairs_handle(ctx, cfg, f, pairs)
            lists = mh.lists()
            crc = 0
            for m in lists:
                crc = zlib.crc32(np.ascontiguousarray(m).tobytes(), crc)
            nm = sum(len(m) for m in lists)
            mh.free()
            steps = a.steps if jname == "config4" else max(2, a.steps // 4)
            best = None
            for rep in range(3):
                ctx.set_profiling(True); ctx.profile_reset()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(steps):
                    hip.match_pairs_handle(ctx, cfg, f, pairs).free()
                torch.cuda.synchronize(); t = (time.perf_counter() - t0) / steps * 1e3
                prof = {k: v[0] / steps for k, v in ctx.profile().items() if k.startswith("matcher") and not k.endswith("(host)")}
                ctx.set_profiling(False)
                if best is None or t < best[0]:
                    best = (t, prof)
            print(f"{short:22s} {jname:14s} call {best[0]:8.3f} ms  pairs {len(pairs)}  matches {nm}  crc {crc:08x}  " +
                  "  ".join(f"{k.replace('matcher ', '')} {v:.3f}" for k, v in best[1].items()), flush=True)
            f.free()
        ctx.close()


if __name__ == "__main__":
    main()
