#!/usr/bin/env python3
"""GPU probe: what is inside a rank's `feature all-gather` lap of the one-device rehearsal (bench_match.rehearse)?
cProfile of ShardedJob.exchange() for rank 3 of 8, config 4 (38 images) and a config-5-shaped job (--c5-images, default 64)."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from openpano_amd import hip, synth
from openpano_amd.config import PanoConfig
from openpano_amd.distributed import HipEngine, ShardedJob, shard_images

c5 = int(sys.argv[sys.argv.index("--c5-images") + 1]) if "--c5-images" in sys.argv else 64
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
stream = torch.cuda.Stream(device=dev); torch.cuda.set_stream(stream)
ctx = hip.Context(0, stream.cuda_stream); cfg = PanoConfig()
for kind in ("config4", "config5"):
    if kind == "config4":
        n, H, W = 38, 867, 1300
        d_imgs = [torch.from_numpy(v).to(dev) for v in synth.image_set(n, H, W, seed=38, overlap=0.45, rows=2, shuffle=True)]
        inputs = [(t.data_ptr(), H, W) for t in d_imgs]
    else:
        n, H, W = c5, 3000, 4000
        d_imgs = synth.config5_views(list(range(n)), dev)
        inputs = [(t.data_ptr(), H, W, "u8") for t in d_imgs]
    eng = HipEngine(ctx, cfg, dev)
    one = ShardedJob(eng, n, dev)
    one.sift(hip.SiftCall(ctx, cfg, inputs)); one.exchange()
    table = (one.desc.clone(), one.coor.clone(), list(one.counts)); one.close()
    rank, world = 3, 8
    ids = shard_images(n, rank, world)
    job = ShardedJob(eng, n, dev, overlap=True, rehearsal=(rank, world, table))
    call = hip.SiftCall(ctx, cfg, [inputs[g] for g in ids])
    for _ in range(3):
        job.sift(call); job.exchange(); job.match(); job.ransac_summary([(W, H)] * n, 1)
    job.sift(call); torch.cuda.synchronize()
    t0 = time.perf_counter(); job.exchange(); torch.cuda.synchronize(); lap = (time.perf_counter() - t0) * 1e3
    job.match(); job.sift(call); torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable(); job.exchange(); torch.cuda.synchronize(); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18)
    print(f"== {kind}: {n} images, rank {rank} of {world}: exchange lap {lap:.3f} ms, own pairs {len(job.local_sel)}, pairs {len(job.my_pairs)}")
    print("\n".join(l[:150] for l in s.getvalue().splitlines()[4:30]))
    job.close()
    if eng._feats is not None:
        eng._feats.free(); eng._feats = None
    del d_imgs; torch.cuda.empty_cache()
ctx.close()
