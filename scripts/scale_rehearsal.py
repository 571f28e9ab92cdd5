#!/usr/bin/env python3
"""One-device rehearsal of the 1 / 2 / 4 / 8-rank strong-scaled jobs (bench_match.rehearse): BASELINE config 4 and config 5.
Writes gpurun_out/<tag>_scale_rehearsal.json (copy it to profiles/scale_rehearsal_latest.json: bench.py --gpus N puts the
matching `predicted` block next to its measured phases).   python scripts/scale_rehearsal.py [tag] [--c5-images 128]"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    from openpano_amd import hip
    from openpano_amd.config import PanoConfig
    from bench_match import rehearse
    tag = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "r05"
    c5 = int(sys.argv[sys.argv.index("--c5-images") + 1]) if "--c5-images" in sys.argv else 128
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx = hip.Context(0, stream.cuda_stream)
    cfg = PanoConfig()
    log = lambda m: print("[rehearsal]", m, file=sys.stderr, flush=True)      # noqa: E731
    out = {"_meta": {"lib_sha256_16": hashlib.sha256(open(hip.LIB_PATH, "rb").read()).hexdigest()[:16], "device": torch.cuda.get_device_name(0), "tag": tag}}
    for kind in ("config4", "config5"):
        out[kind] = rehearse(hip, ctx, cfg, kind, (1, 2, 4, 8), dev, log, c5_images=c5)
    ctx.close()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", f"{tag}_scale_rehearsal.json")
    json.dump(out, open(path, "w"), indent=1)
    for kind in ("config4", "config5"):
        print(kind, {n: (w["job_ms"], w["phase_ms"]) for n, w in out[kind]["worlds"].items()})
    print("wrote", path)


if __name__ == "__main__":
    main()
