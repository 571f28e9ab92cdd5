#!/usr/bin/env python3
"""A/B timing of the blend kernels across builds of libopenpano_hip.so in ONE process (GPU box): bench.py's blend section
(38 resident 1300x867 views, spherical projection, linear and 5-band) per library + a CRC of the two canvases.

    python scripts/blend_ab.py lib_a.so ...      ("product" is always first)"""
import argparse
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("libs", nargs="*")
    a = ap.parse_args()
    import numpy as np
    import torch
    import bench
    from openpano_amd import hip, synth
    from openpano_amd.config import PanoConfig
    H, W = 867, 1300
    dev = torch.device("cuda", 0)
    views = synth.image_set(38, H, W, seed=38, overlap=0.45, rows=2, shuffle=True)
    d_imgs = [torch.from_numpy(v).to(dev) for v in views]
    inputs = [(t.data_ptr(), H, W) for t in d_imgs]
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    for name in ["product"] + list(a.libs):
        path = os.path.join(ROOT, "openpano_amd", "libopenpano_hip.so") if name == "product" else os.path.abspath(name)
        hip._lib = None
        hip.LIB_PATH = path
        ctx = hip.Context(0, stream.cuda_stream)
        res = bench.run_blend(hip, ctx, PanoConfig(), inputs, H, W, a, lambda m: None)
        homos = res.pop("_homos")
        crcs = []
        for mb in (0, 5):
            cv = hip.blend(ctx, PanoConfig(MULTIBAND=mb), inputs, homos, 2, len(inputs) // 2)
            crcs.append("%08x" % zlib.crc32(cv.numpy().tobytes())); cv.free()
        short = os.path.basename(path).replace("libopenpano_hip_", "").replace(".so", "")
        print(f"{short:20s} crc {crcs}  " + "  ".join(f"{k}: {v['ms_per_blend']:.4f} ms {v['stage_ms']}" for k, v in res.items()), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
