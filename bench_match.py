"""Job-level sections of bench.py: exchange + all-pairs match + RANSAC + result gather through
openpano_amd.distributed.ShardedJob (second metric of BASELINE.json: matches/sec).

The same code runs at every N: at N = 1 the exchange is the identity (unless OPENPANO_FORCE_DIST=1
initialised a one-rank RCCL group, in which case the all-gather really runs), at N > 1 features are
all-gathered over RCCL/xGMI in one bucket and each rank matches / RANSACs a K_i*K_j-balanced share
of the unordered pair list (stitcher.cc:100), results are all-gathered.
"""
import time

import numpy as np
import torch


def _mfma_roofline(prof, flops, pmc=None, pmc_src=None):
    """forward sweep (every row of the smaller set) + reverse strip (survivors only), both including
    their exact re-score epilogues; algorithmic work = 2*128*Ki*Kj flop per unordered pair"""
    mfma_ms = (prof.get("matcher mfma forward") or 0.0) + (prof.get("matcher mfma reverse") or 0.0)
    if not mfma_ms:
        return None
    # the sweeps rank with a two-term bf16 split: three v_mfma_f32_32x32x16_bf16 per 16 elements, i.e.
    # 3x the algorithmic flop are executed on the bf16 matrix pipe (dense peak ~2.5 PFLOP/s,
    # MI355X_MICROARCH.md); the fp32-MFMA figure of SURVEY 8(d) is kept next to it for comparison
    alg = flops / (mfma_ms * 1e-3) / 1e12             # this rank's share, this rank's kernel time
    return {"kernel": "matcher mfma forward + reverse", "bound": "mfma", "achieved": 3.0 * alg, "peak": 2500.0,
            "unit": "TFLOP/s", "frac": 3.0 * alg / 2500.0, "traffic": None,
            "dtype": "bf16 x3 split, fp32 accumulate; exact fp32 re-score decides",
            "algorithmic_tflops": alg, "algorithmic_over_fp32_mfma_peak": alg / 157.3,
            "algorithmic_flop_per_launch": flops, "avg_launch_ms": mfma_ms,
            # the clock under matrix load depends on the operand data: bf16 operands like split descriptors sustain
            # 1.93 PFLOP/s on 50 ms launches of nothing but MFMAs (scripts/ubench/mfma_power.hip)
            "data_ceiling": 1930.0, "frac_of_data_ceiling": 3.0 * alg / 1930.0,
            # matrix-pipe utilisation from the hardware counters (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs) and
            # SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 flop; scripts/gpu_pmc.sh `mfma` pass over a config-4 bench run of this build)
            "mfma_busy": ((pmc or {}).get("k_match_sweep") or {}).get("mfma_busy"),
            "mfma_flop_executed_per_launch_config4": ((pmc or {}).get("k_match_sweep") or {}).get("mfma_flop_executed"),
            "mfma_counter_source": pmc_src if ((pmc or {}).get("k_match_sweep") or {}).get("mfma_busy") is not None else None}


def _reduce(dist, dev, tmax_vals, sum_vals):
    t = torch.tensor(tmax_vals, dtype=torch.float64, device=dev)
    s = torch.tensor(sum_vals, dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return [float(x) for x in t], [float(x) for x in s]


def run_job_loops(hip, ctx, cfg, feats, n_total, shapes, args, dist, dev, rank, world, barrier, log, keep_ransac_inputs=False):
    """Timed loops over the features the SIFT loop of bench.py left in HBM."""
    from openpano_amd.distributed import HipEngine, ShardedJob
    eng = HipEngine(ctx, cfg, dev)
    job = ShardedJob(eng, n_total, dev)
    job.adopt(feats)
    gather_ms = None
    job.exchange()                                                   # warm-up (RCCL init, allocation)
    if dist is not None:
        job.exchange()
        gather_ms = None
        for _ in range(3):                                           # best of 3: the first calls also grow torch's allocator
            barrier(); t0 = time.perf_counter()
            job.exchange()
            barrier(); dt = (time.perf_counter() - t0) * 1e3
            gather_ms = dt if gather_ms is None else min(gather_ms, dt)
    mine = job.my_pairs
    flops = sum(2.0 * 128 * job.gcounts[i] * job.gcounts[j] for i, j in mine)
    nmatch = job.match()                                             # warm-up, and the lists RANSAC consumes
    # stage times: a few untimed calls with every stage bracketed by HIP events; the timed loop below runs without them
    # (each bracketed stage costs two event records and the gap they put between kernels)
    ctx.set_profiling(True); ctx.profile_reset()
    psteps = 5
    for _ in range(psteps):
        eng.match_only(job.tab, mine)
    prof = {k: v[0] / psteps for k, v in ctx.profile().items() if k.startswith("matcher")}
    ctx.set_profiling(False)
    steps = max(1, min(args.steps, 50))
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.match_only(job.tab, mine)                                # results stay in the op_matches handle
    barrier()
    t = time.perf_counter() - t0
    (tmax,), (npairs, nm, fl) = _reduce(dist, dev, [t], [float(len(mine)), float(nmatch), flops])
    res = {
        "image_pairs_per_s": npairs * steps / tmax, "matches_per_s": nm * steps / tmax,
        "image_pairs": int(npairs), "matches": int(nm), "steps": steps, "ms_per_step": tmax / steps * 1e3,
        "descriptor_allgather_ms": gather_ms, "allgather_bytes_per_rank": int(max(sum(job.counts), 1) * 528) if dist is not None else None,
        "stage_ms": {k: round(v, 4) for k, v in prof.items()},
        "roofline": _mfma_roofline(prof, flops, getattr(args, "pmc", None), getattr(args, "pmc_src", None)),
    }
    # ---- RANSAC over this rank's pairs (batched TransformEstimation::get_transform + acceptance) ----
    seeds = job.seeds(1)
    ok, inl = eng.ransac_summary(job.tab, job.mh, mine, shapes, seeds)                     # warm-up
    ctx.set_profiling(True); ctx.profile_reset()
    for _ in range(3):
        eng.ransac_summary(job.tab, job.mh, mine, shapes, seeds)
    rprof = {k: v[0] / 3 for k, v in ctx.profile().items() if k.startswith("ransac")}
    ctx.set_profiling(False)
    rsteps = max(1, min(args.steps, 30))
    barrier(); t0 = time.perf_counter()
    for _ in range(rsteps):
        eng.ransac_summary(job.tab, job.mh, mine, shapes, seeds)
    barrier(); tr = time.perf_counter() - t0
    (trmax,), (okt, inlt) = _reduce(dist, dev, [tr], [float(ok), float(inl)])
    res["ransac"] = {"image_pairs_per_s": npairs * rsteps / trmax, "ms_per_step": trmax / rsteps * 1e3, "pairs": int(npairs),
                     "accepted_pairs": int(okt), "inliers": int(inlt), "iterations": cfg.RANSAC_ITERATIONS,
                     "stage_ms": {k: round(v, 4) for k, v in rprof.items()}}
    # ---- results (match lists + RANSAC vectors) to every rank: rank 0 runs the host stages ----
    if dist is not None:
        job.ransac(shapes, 1)
        barrier(); t0 = time.perf_counter()
        allres = job.gather()
        barrier(); res["match_results_gather_ms"] = (time.perf_counter() - t0) * 1e3
        assert len(allres) == n_total * (n_total - 1) // 2
    else:
        res["match_results_gather_ms"] = None
    if keep_ransac_inputs:          # bench.py's CPU baseline of the RANSAC loop runs over exactly these lists
        res["_ransac_inputs"] = ([np.asarray(m, np.int32).reshape(-1, 2) for m in job.lists], list(mine),
                                 [feats.get(i)[1] for i in range(n_total)], list(shapes))
    job.close()
    return res


def job_parity(hip, ctx, cfg, job, d_imgs, u8, shapes, log, npairs_sample=96):
    """Outside every timed region (N = 1): the job the timed pass just ran, against the CPU oracle --
    descriptors and coordinates of EVERY image (CRC of the whole set), and for a seeded sample of pairs (half of them
    overlapping views with hundreds of matches) the match set (count + order-free digest), the RANSAC winner, its
    inlier count, the accepted flag and the homography.  -> dict with ok = everything bit-identical."""
    import os
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    from checkers import Oracle
    orc = Oracle(cfg)
    nt = min(64, os.cpu_count() or 1)
    n = job.n
    tab = job.tab

    def host_f32(i):
        a = d_imgs[i].cpu().numpy()
        return (a.astype(np.float64) / 255.0).astype(np.float32) if u8 else a
    with ThreadPoolExecutor(nt) as ex:
        want = list(ex.map(lambda i: orc.detect_feature(host_f32(i)), range(n)))
    got = [tab.get(i) for i in range(n)]
    bad_img = [i for i in range(n) if not (np.array_equal(got[i][0], want[i][0]) and np.array_equal(got[i][1], want[i][1]))]
    pairs = job.my_pairs
    rng = np.random.default_rng(5)
    same = [k for k, (i, j) in enumerate(pairs) if i // 8 == j // 8] if u8 else []
    half = min(len(same), npairs_sample // 2)
    sel = set(rng.choice(same, half, replace=False).tolist()) if half else set()
    sel |= set(rng.choice(len(pairs), min(len(pairs), npairs_sample - half), replace=False).tolist())
    sel = sorted(sel)
    lists = job.lists
    cnt, dig = orc.match_pairs_digest([w[0] for w in want], [pairs[k] for k in sel], os.cpu_count() or 8)
    bad_match = [list(pairs[k]) for k, c, d in zip(sel, cnt, dig) if len(lists[k]) != int(c) or orc.match_digest(lists[k]) != int(d)]
    seeds = job.seeds(1)
    res = hip.ransac_pairs(ctx, cfg, tab, job.mh, pairs, shapes, seeds=seeds)

    def one(k):
        i, j = pairs[k]
        return orc.ransac(lists[k], want[i][1], want[j][1], shapes[i], shapes[j], seeds[k])
    with ThreadPoolExecutor(nt) as ex:
        wr = list(ex.map(one, sel))
    bad_ransac = []
    for k, w in zip(sel, wr):
        g = res[k]
        same_r = (g["best_hyp"] == w["best_hyp"] and g["best_count"] == w["best_count"] and g["ok"] == w["ok"] and g["confidence"] == w["confidence"]
                  and np.array_equal(g["inliers"], w["inliers"]) and (not w["ok"] or np.array_equal(g["homo"], w["homo"])))
        if not same_r:
            bad_ransac.append(list(pairs[k]))
    out = {"checked": True, "images": n, "descriptors": int(sum(len(w[0]) for w in want)), "images_differing": bad_img,
           "descriptor_crc32": zlib.crc32(b"".join(np.ascontiguousarray(g[0]).tobytes() for g in got)),
           "oracle_descriptor_crc32": zlib.crc32(b"".join(np.ascontiguousarray(w[0]).tobytes() for w in want)),
           "pairs": len(sel), "matches": int(cnt.sum()), "longest_match_list": int(max((len(lists[k]) for k in sel), default=0)),
           "pairs_differing": bad_match, "ransac_pairs": len(sel), "ransac_accepted": int(sum(1 for w in wr if w["ok"])),
           "ransac_inliers": int(sum(len(w["inliers"]) for w in wr if w["ok"])), "ransac_pairs_differing": bad_ransac,
           "ok": not bad_img and not bad_match and not bad_ransac}
    if not out["ok"]:
        log(f"PARITY FAILURE ({'config 5' if u8 else 'job'}): images {bad_img}, match pairs {bad_match}, ransac pairs {bad_ransac}")
    return out


def run_strong_job(hip, ctx, cfg, kind, args, dist, dev, rank, world, barrier, log, parity=False):
    """ONE job dealt over the N ranks (strong scaling), one warm pass + one timed pass with a barrier
    between phases: SIFT on the local shard, feature all-gather, match + RANSAC on this rank's share
    of the pair list, result gather.
      config4: BASELINE config 4, the 38 seeded 1300x867 views (fp32, as bench.py's headline)
      config5: BASELINE config 5, 128 synthetic 4000x3000 uint8 images, 8128 pairs (MFMA stress)"""
    from openpano_amd import synth
    from openpano_amd.distributed import HipEngine, ShardedJob, shard_images
    t0 = time.perf_counter()
    if kind == "config4":
        n, H, W = 38, 867, 1300
        allv = synth.image_set(n, H, W, seed=38, overlap=0.45, rows=2, shuffle=True)
        ids = shard_images(n, rank, world)
        d_imgs = [torch.from_numpy(allv[g]).to(dev) for g in ids]
        inputs = [(t.data_ptr(), H, W) for t in d_imgs]
        what = "38 unordered 1300x867 fp32 views (seed 38)"
    else:
        n, H, W = args.c5_images, 3000, 4000
        ids = shard_images(n, rank, world)
        d_imgs = synth.config5_views(ids, dev)
        inputs = [(t.data_ptr(), H, W, "u8") for t in d_imgs]
        what = f"{n} synthetic 4000x3000 uint8 images, groups of 8 share a texture (openpano_amd/synth.py config5_views)"
    torch.cuda.synchronize()
    log(f"strong {kind}: {len(ids)} local of {n} images generated in {time.perf_counter() - t0:.1f} s")
    eng = HipEngine(ctx, cfg, dev)
    job = ShardedJob(eng, n, dev, overlap=True)        # N > 1: the pairs of two own images are matched while the exchange is in flight
    shapes = [(W, H)] * n
    call = hip.SiftCall(ctx, cfg, inputs) if inputs else None
    sift_in = (lambda: call()) if call is not None else []

    def one_pass(timed):
        ph = {}

        def lap(name, fn):
            barrier(); t = time.perf_counter(); r = fn(); barrier(); ph[name] = (time.perf_counter() - t) * 1e3
            return r
        if timed:
            ctx.set_profiling(True); ctx.profile_reset()
        k = lap("sift", lambda: job.sift(sift_in))
        lap("feature all-gather", job.exchange)
        nm = lap("match", job.match)
        prof = {}
        if timed:
            prof = {kk: v[0] for kk, v in ctx.profile().items()}
            ctx.set_profiling(False)
        ok, inl = lap("ransac", lambda: job.ransac_summary(shapes, 1))
        if dist is not None:
            job.rres = None
            lap("result gather", job.gather)
        return ph, k, nm, ok, inl, prof

    one_pass(False)                                   # warm (allocation pool, RCCL, kernels)
    one_pass(False)                                   # (the output buffers take their steady size class -- 1.25 x the previous batch -- in the second call)
    barrier(); t0 = time.perf_counter()
    ph, k, nm, ok, inl, prof = one_pass(True)
    wall = (time.perf_counter() - t0) * 1e3
    # the RANSAC phase's own stage table, from one more (untimed) call with the stage events on
    ctx.set_profiling(True); ctx.profile_reset()
    job.ransac_summary(shapes, 1)
    rprof = {kk: v[0] for kk, v in ctx.profile().items() if kk.startswith("ransac")}
    ctx.set_profiling(False)
    mine = job.my_pairs
    flops = sum(2.0 * 128 * job.gcounts[i] * job.gcounts[j] for i, j in mine)
    names = list(ph)
    tm, sm = _reduce(dist, dev, [ph[x] for x in names] + [wall], [float(k), float(len(mine)), float(nm), float(ok), float(inl), flops])
    phm = dict(zip(names, tm[:-1]))
    # every rank's own phase times (the maxima above decide the job; the spread says where a rank waits for another)
    per_rank = None
    if dist is not None:
        mine_t = torch.tensor([ph[x] for x in names], dtype=torch.float64, device=dev)
        allt = torch.empty(world * len(names), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(allt, mine_t)
        per_rank = [{x: round(float(v), 4) for x, v in zip(names, row)} for row in allt.view(world, len(names)).cpu()]
    # At N > 1 the pairs of two own images are matched INSIDE the 'feature all-gather' lap (ShardedJob(overlap=True)), the
    # rest in 'match': the match rates divide ALL pairs by the sum of the two laps (dividing by 'match' alone would credit
    # that lap with pairs it did not handle -- 12 % of them at N = 8 on config 5).  N = 1: no own-pair pass, 'match' alone.
    overlapped = world > 1 and len(job.local_sel) > 0
    match_span = phm["match"] + (phm["feature all-gather"] if overlapped else 0.0)
    match_span_note = "feature all-gather + match (own pairs are matched during the exchange)" if overlapped else "match"
    res = {"workload": f"{what}; ONE job dealt round-robin over {world} GPU(s); inputs resident in HBM", "scaling": "strong",
           "images": n, "image": [H, W], "n_gpus": world, "descriptors": int(sm[0]), "keypoints_per_image": sm[0] / n,
           "image_pairs": int(sm[1]), "matches": int(sm[2]), "accepted_pairs": int(sm[3]), "inliers": int(sm[4]),
           "phase_ms": {x: round(v, 4) for x, v in phm.items()}, "per_rank_phase_ms": per_rank, "job_wall_ms": tm[-1],
           "own_pairs_matched_during_exchange": len(job.local_sel),
           "keypoints_per_s": sm[0] / (phm["sift"] * 1e-3), "image_pairs_per_s": sm[1] / (match_span * 1e-3),
           "matches_per_s": sm[2] / (match_span * 1e-3), "ransac_image_pairs_per_s": sm[1] / (phm["ransac"] * 1e-3),
           "match_rate_span": match_span_note,
           "match_stage_ms": {x: round(v, 4) for x, v in prof.items() if x.startswith("matcher")},
           "sift_stage_ms": {x: round(v, 4) for x, v in prof.items() if not x.startswith("matcher")},
           "ransac_stage_ms": {x: round(v, 4) for x, v in rprof.items()},
           "match_roofline": _mfma_roofline(prof, flops, getattr(args, "pmc", None), getattr(args, "pmc_src", None)),
           "allgather_bytes_per_rank": int(max(sum(job.counts), 1) * 528) if dist is not None else None}
    if kind == "config5" and res["match_roofline"] is not None:
        # The matrix-pipe counters of THESE launches (the per-kernel averages replayed above are dominated by config 4's short
        # workgroups): per-dispatch MfmaUtil and the shader clock the chip holds under the sweep, from a counter pass over this
        # build (scripts/pmc_config5_mfma.py -> profiles/config5_mfma_latest.json, hash-checked like pmc_latest.json)
        c5 = getattr(args, "c5mfma", None)
        if c5:
            f5 = c5["config5_forward"]
            rl = res["match_roofline"]
            rl["mfma_busy_config5_forward"] = f5["mfma_busy"]
            rl["shader_clock_ghz_under_the_sweep"] = f5["shader_clock_ghz"]
            rl["peak_at_that_clock_tflops"] = f5["peak_tflops_at_measured_clock"]
            rl["frac_of_peak_at_that_clock"] = rl["achieved"] / f5["peak_tflops_at_measured_clock"]
            rl["mfma_busy_config5_source"] = f"replayed from profiles/config5_mfma_latest.json (lib_sha256_16 {c5['_meta']['lib_sha256_16']}; not collected in this run)"
    if parity and world == 1:
        t0 = time.perf_counter()
        # config 5: as many pairs as ONE rank of an 8-GPU job owns (8128 / 8 = 1016), half of them overlapping views
        res["parity"] = job_parity(hip, ctx, cfg, job, d_imgs, kind == "config5", shapes, log,
                                   npairs_sample=(len(job.my_pairs) + 7) // 8 if kind == "config5" else 96)
        log(f"strong {kind}: parity of the timed job against the oracle took {time.perf_counter() - t0:.1f} s -> ok={res['parity']['ok']}")
    job.close()
    if eng._feats is not None:
        eng._feats.free(); eng._feats = None
    del d_imgs
    torch.cuda.empty_cache()
    return res


def rehearse(hip, ctx, cfg, kind, worlds, dev, log, c5_images=128):
    """One-device rehearsal of the strong-scaled job: every rank's share of an N-rank job (N in ``worlds``) run one after the
    other through ShardedJob(overlap=True, rehearsal=...) -- the code a rank runs at N > 1 -- with each phase timed, and the
    union of the ranks' results compared with the single-rank job (match lists by CRC, RANSAC by accepted pairs / inliers).
    -> {"N": {"phase_ms": max over ranks per phase, "job_ms": their sum (phases are separated by barriers in the real job),
    "per_rank_phase_ms": [...], ...}}: the share model a measured 1 -> 8 curve can be read against (`predicted` in bench.py's
    N > 1 lines).  What it cannot contain is the xGMI part of the exchange: the peers' slices are device-to-device copies here."""
    import zlib
    from openpano_amd import synth
    from openpano_amd.distributed import HipEngine, ShardedJob, shard_images
    if kind == "config4":
        n, H, W = 38, 867, 1300
        allv = synth.image_set(n, H, W, seed=38, overlap=0.45, rows=2, shuffle=True)
        d_imgs = [torch.from_numpy(v).to(dev) for v in allv]
        inputs = [(t.data_ptr(), H, W) for t in d_imgs]
    else:
        n, H, W = c5_images, 3000, 4000
        d_imgs = synth.config5_views(list(range(n)), dev)
        inputs = [(t.data_ptr(), H, W, "u8") for t in d_imgs]
    torch.cuda.synchronize()
    shapes = [(W, H)] * n
    eng = HipEngine(ctx, cfg, dev)
    one = ShardedJob(eng, n, dev)
    one.sift(hip.SiftCall(ctx, cfg, inputs)); one.exchange(); one.match()
    want = {p: zlib.crc32(np.ascontiguousarray(m).tobytes()) for p, m in zip(one.my_pairs, one.lists)}
    want_r = one.ransac_summary(shapes, 1)
    table = (one.desc.clone(), one.coor.clone(), list(one.counts))
    ktot = sum(table[2])
    one.close()
    out = {}
    for world in worlds:
        per_rank, seen, tot = [], {}, [0, 0]
        for rank in range(world):
            ids = shard_images(n, rank, world)
            job = ShardedJob(eng, n, dev, overlap=True, rehearsal=(rank, world, table))
            call = hip.SiftCall(ctx, cfg, [inputs[g] for g in ids]) if ids else []
            best = None
            for rep in range(3):                       # the third pass is the one kept: buffers at their steady size class
                ph = {}

                def lap(name, fn):
                    torch.cuda.synchronize(); t = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ph[name] = (time.perf_counter() - t) * 1e3
                    return r
                lap("sift", lambda: job.sift(call))
                lap("feature all-gather", job.exchange)
                lap("match", job.match)
                okr = lap("ransac", lambda: job.ransac_summary(shapes, 1))
                best = ph
            for p, m in zip(job.my_pairs, job.lists):
                assert p not in seen, (world, rank, p)
                seen[p] = zlib.crc32(np.ascontiguousarray(m).tobytes())
            tot[0] += okr[0]; tot[1] += okr[1]
            per_rank.append({k: round(v, 4) for k, v in best.items()} | {"images": len(ids), "pairs": len(job.my_pairs), "own_pairs": len(job.local_sel)})
            job.close()
        assert seen == want, f"rehearsal world {world}: match lists differ from the single-rank job"
        assert tuple(tot) == tuple(want_r), (world, tot, want_r)
        phm = {k: max(r[k] for r in per_rank) for k in ("sift", "feature all-gather", "match", "ransac")}
        job_ms = sum(phm.values())
        out[str(world)] = {"phase_ms": {k: round(v, 4) for k, v in phm.items()}, "job_ms": round(job_ms, 4), "per_rank_phase_ms": per_rank,
                           "keypoints_per_s": ktot / (phm["sift"] * 1e-3), "image_pairs_per_s": len(want) / ((phm["feature all-gather"] + phm["match"]) * 1e-3 if world > 1 else phm["match"] * 1e-3),
                           "ransac_image_pairs_per_s": len(want) / (phm["ransac"] * 1e-3), "equals_single_rank_job": True}
        log(f"rehearsal {kind} N={world}: phases {out[str(world)]['phase_ms']} job {job_ms:.3f} ms")
    if eng._feats is not None:
        eng._feats.free(); eng._feats = None
    del d_imgs
    torch.cuda.empty_cache()
    return {"kind": kind, "images": n, "descriptors": int(ktot), "image_pairs": len(want), "accepted_pairs": int(want_r[0]), "inliers": int(want_r[1]),
            "note": "every rank's share run one after the other on ONE device through ShardedJob(overlap=True, rehearsal=...); phase time of an N-rank job = max over its ranks; "
                    "the exchange's remote part (xGMI) is a device-to-device copy here", "worlds": out}
