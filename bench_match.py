"""All-pairs matching loop of bench.py (second metric of BASELINE.json: matches/sec).

N = 1: the 703 unordered pairs of this rank's 38 images.
N > 1: descriptors are all-gathered over RCCL/xGMI (variable sizes: counts first, then padded
payload), every rank rebuilds the global image-indexed feature table in its HBM and matches an
interleaved 1/N share of the global unordered pair list (stitcher.cc:100 partitioned by rank).
"""
import time

import numpy as np
import torch


def run_match_loop(hip, ctx, cfg, feats, args, dist, dev, rank, world, barrier, log):
    nloc = feats.num_images
    counts = [feats.count(i) for i in range(nloc)]
    gather_ms = None
    if dist is not None:
        from openpano_amd.distributed import allgather_descriptors
        total = int(feats.total)
        # zero-copy torch view of the library-owned descriptor buffer (same HIP runtime)
        dloc = torch.as_tensor(feats.desc_device_array(), device=dev) if total else torch.zeros((0, 128), device=dev)
        allgather_descriptors(dloc, counts)                              # warm-up (RCCL init)
        barrier()
        t0 = time.perf_counter()
        glob, all_counts = allgather_descriptors(dloc, counts)
        barrier()
        gather_ms = (time.perf_counter() - t0) * 1e3
        gfeats = hip.Features.from_device(ctx, glob.data_ptr(), all_counts)
        nglob = len(all_counts)
    else:
        gfeats = feats
        nglob = nloc
        all_counts = counts
    pairs = [(i, j) for i in range(nglob) for j in range(i + 1, nglob)]
    from openpano_amd.distributed import partition_pairs
    mine = partition_pairs(pairs, rank, world, all_counts if dist is not None else None)
    flops = sum(2.0 * 128 * all_counts[i] * all_counts[j] for i, j in mine)
    m = hip.match_pairs(ctx, cfg, gfeats, mine)                      # warm-up (and the match count)
    nmatch = sum(len(x) for x in m)
    gather_results_ms = None
    if dist is not None:                                             # results to every rank (rank 0 runs the host stages)
        from openpano_amd.distributed import gather_match_results
        barrier(); t0 = time.perf_counter()
        allm = gather_match_results(mine, m, dev)
        barrier(); gather_results_ms = (time.perf_counter() - t0) * 1e3
        assert len(allm) == len(pairs)
    ctx.set_profiling(True); ctx.profile_reset()
    steps = max(1, min(args.steps, 10))
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        mh = hip.match_pairs_handle(ctx, cfg, gfeats, mine)         # results stay in the op_matches handle
        mh.free()
    barrier()
    t = time.perf_counter() - t0
    prof = {k: v[0] / steps for k, v in ctx.profile().items() if k.startswith("matcher")}
    ctx.set_profiling(False)
    tt = torch.tensor([t], dtype=torch.float64, device=dev)
    agg = torch.tensor([float(len(mine)), float(nmatch), flops], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
    tmax = float(tt[0]); npairs, nm, fl = (float(x) for x in agg)
    # forward sweep (every row of the smaller set) + reverse strip (survivors only), both including
    # their exact re-score epilogues; algorithmic work = 2*128*Ki*Kj flop per unordered pair
    mfma_ms = (prof.get("matcher mfma forward") or 0.0) + (prof.get("matcher mfma reverse") or 0.0)
    res = {
        "image_pairs_per_s": npairs * steps / tmax, "matches_per_s": nm * steps / tmax,
        "image_pairs": int(npairs), "matches": int(nm), "steps": steps, "ms_per_step": tmax / steps * 1e3,
        "descriptor_allgather_ms": gather_ms, "match_results_gather_ms": gather_results_ms, "stage_ms": {k: round(v, 4) for k, v in prof.items()},
        "roofline": None,
    }
    if mfma_ms:
        # the sweeps rank with a two-term bf16 split: three v_mfma_f32_32x32x16_bf16 per 16 elements, i.e.
        # 3x the algorithmic flop are executed on the bf16 matrix pipe (dense peak ~2.5 PFLOP/s,
        # MI355X_MICROARCH.md); the fp32-MFMA figure of SURVEY 8(d) is kept next to it for comparison
        alg = flops / (mfma_ms * 1e-3) / 1e12             # this rank's share, this rank's kernel time
        res["roofline"] = {"kernel": "matcher mfma forward + reverse", "bound": "mfma", "achieved": 3.0 * alg, "peak": 2500.0,
                           "unit": "TFLOP/s", "frac": 3.0 * alg / 2500.0, "traffic": None,
                           "dtype": "bf16 x3 split, fp32 accumulate; exact fp32 re-score decides",
                           "algorithmic_tflops": alg, "algorithmic_over_fp32_mfma_peak": alg / 157.3,
                           "algorithmic_flop_per_launch": flops, "avg_launch_ms": mfma_ms}
    # ---- RANSAC over the same pairs (batched TransformEstimation::get_transform + acceptance) ----
    if world == 1:
        shapes = [(args.W, args.H)] * nglob
        mh = hip.match_pairs_handle(ctx, cfg, gfeats, mine)
        ok, inl = hip.ransac_pairs_summary(ctx, cfg, gfeats, mh, mine, shapes, base_seed=1)     # warm-up
        rsteps = max(1, min(args.steps, 5))
        ctx.set_profiling(True); ctx.profile_reset()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(rsteps):
            hip.ransac_pairs_summary(ctx, cfg, gfeats, mh, mine, shapes, base_seed=1)
        torch.cuda.synchronize(); tr = time.perf_counter() - t0
        rprof = {k: v[0] / rsteps for k, v in ctx.profile().items() if k.startswith("ransac")}
        ctx.set_profiling(False)
        mh.free()
        res["ransac"] = {"image_pairs_per_s": len(mine) * rsteps / tr, "ms_per_step": tr / rsteps * 1e3, "pairs": len(mine),
                         "accepted_pairs": ok, "inliers": inl, "iterations": cfg.RANSAC_ITERATIONS,
                         "stage_ms": {k: round(v, 4) for k, v in rprof.items()}}
    if gfeats is not feats:
        gfeats.free()
    return res
