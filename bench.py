#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native OpenPano hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): SIFT keypoints+descriptors/sec (``value``) and all-pairs matches/sec
(``match``) on BASELINE config 4 -- 38 unordered 1300x867 images -- restated as seeded synthetic
views (openpano_amd/synth.py; the reference's example data needs the network).

One step = one pass of the hot path over one batch: op_sift_batch over this rank's 38 images
(inputs already resident in HBM), descriptors left in HBM.  A second timed loop measures the
all-pairs exact match over the same descriptors (RCCL all-gather of descriptors first when N>1).
Scaling is weak: every rank owns 38 images; the job is the unordered set of 38*N images.

The JSON line also carries
  roofline      live HIP-event timing of the dominant kernel vs its algorithmic HBM bytes,
  cpu_baseline  the reference's CPU path (oracle/_ref when it travelled, else the C oracle) timed
                on this box's host cores on a bounded sample of the same images (rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402  (device memory, streams, torch.distributed: plumbing only)

HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 achievable)
MFMA_F32_PEAK_TFLOPS = 157.3


class _StdoutToStderr:
    """The reference's classes print to the C stdout (e.g. "BuildTrees: ..."); the bench contract is
    ONE JSON line on stdout, so file descriptor 1 points at stderr while they run."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(self._saved, 1)
        os.close(self._saved)


def pyramid_pixels(cfg, h, w):
    """P = sum of octave pixels for an h x w source (feature.cc:33-35, dog.cc:105-107)."""
    ratio = np.float32(cfg.SIFT_WORKING_SIZE * 2.0) / np.float32(w + h)
    wh, ww = int(np.float32(h) * ratio), int(np.float32(w) * ratio)
    P = 0
    for i in range(cfg.NUM_OCTAVE):
        f = np.float32(np.float64(np.float32(cfg.SCALE_FACTOR)) ** (-i))
        P += (int(np.ceil(np.float32(ww) * f)) * int(np.ceil(np.float32(wh) * f))) if i else wh * ww
    return P, wh, ww


def cpu_baseline(cfg, views, log):
    """Reference CPU path on this box's host cores, bounded sample (~10-30 s of CPU work)."""
    from checkers import Oracle, Ref, ref_available
    cores = os.cpu_count() or 1
    nsample = int(min(max(16, 2 * cores), 96))
    sample = [views[i % len(views)] for i in range(nsample)]
    kind = "port"
    try:
        if ref_available():
            eng = Ref(cfg); kind = "reference"
        else:
            eng = Oracle(cfg)
    except OSError as e:   # _ref built for another libstdc++/CPU: fall back to the C port
        log(f"oracle/_ref unusable ({e}); using the C oracle")
        eng = Oracle(cfg)
    eng.calc_feature_batch(sample[:2], 1)                      # warm
    t0 = time.perf_counter(); k1 = eng.calc_feature_batch(sample[:4], 1); t1 = time.perf_counter() - t0
    t0 = time.perf_counter(); kall = eng.calc_feature_batch(sample, cores); tall = time.perf_counter() - t0
    return {
        "value": kall / tall, "unit": "keypoints+descriptors/s", "cores": cores, "kind": kind,
        "sample": f"{nsample} of the workload's 1300x867 views, {'OpenMP parallel-for over images like StitcherBase::calc_feature' } with {cores} threads; "
                  f"wall {tall:.2f} s; single-thread rate {k1 / t1:.0f}/s on 4 views",
        "single_thread_value": k1 / t1,
    }


def match_cpu_baseline(cfg, feats, log):
    """The reference's match loop on this box's host cores, bounded sample: the first 128 pairs of
    the workload's pair list with (a) PairWiseMatcher as shipped (FLANN kd-forest, incl. build) and
    (b) the exact FeatureMatcher (the parity oracle), OpenMP over pairs like stitcher.cc:106-109."""
    from checkers import Oracle, Ref, ref_available
    cores = os.cpu_count() or 1
    n = feats.num_images
    descs = [feats.get(i)[0] for i in range(n)]
    pairs = [(i, j) for i in range(n) for j in range(i + 1, n)][:128]
    out = {"cores": cores, "sample": f"first {len(pairs)} of the {n * (n - 1) // 2} pairs, OpenMP over pairs with {cores} threads"}
    eng = None
    try:
        if ref_available():
            eng = Ref(cfg); out["kind"] = "reference"
    except OSError as e:
        log(f"oracle/_ref unusable ({e})")
    if eng is None:
        eng = Oracle(cfg); out["kind"] = "port"
    eng.match_pairs_batch(descs, pairs[:8], cores)                  # warm
    t0 = time.perf_counter(); m = eng.match_pairs_batch(descs, pairs, cores); t = time.perf_counter() - t0
    out["exact_image_pairs_per_s"] = len(pairs) / t; out["exact_matches"] = int(m)
    if out["kind"] == "reference":
        t0 = time.perf_counter(); m2 = eng.match_pairs_batch(descs, pairs, cores, flann=True); t2 = time.perf_counter() - t0
        out["flann_image_pairs_per_s"] = len(pairs) / t2; out["flann_matches"] = int(m2)
    return out


def run_ingest(hip, ctx, cfg, views, k_rank, args):
    """SIFT over HOST-resident images (the reference hands over Mat32f in host RAM, stitcherbase.cc:16):
    every call pays the H2D copies.  fp32 Mat32f (12 B/px) vs decoder bytes (3 B/px, converted on the
    device like read_img) -- SURVEY 8(f).1.  Reported next to `value`, never as `value`."""
    res = {}
    u8 = [(v * 255 + 0.5).astype(np.uint8) for v in views]
    f32 = [(v.astype(np.float64) / 255.0).astype(np.float32) for v in u8]
    for key, imgs in (("host_fp32", f32), ("host_uint8", u8)):
        # page-locked like a decoder's output pool: ONE pinned block sliced into the images (38 separate
        # pin_memory() calls went through PyTorch's caching host allocator, whose small-block path
        # produced 5x slower H2D copies for the uint8 images when large device buffers had been
        # cycled before -- an artefact of the driver program, not of the library)
        stride = (imgs[0].nbytes + 255) & ~255
        block = torch.empty(stride * len(imgs), dtype=torch.uint8).pin_memory()
        flat = block.numpy()
        pinned = []
        for k, x in enumerate(imgs):
            v = flat[k * stride: k * stride + x.nbytes].view(x.dtype).reshape(x.shape)
            v[...] = x
            pinned.append(v)
        f = hip.sift_batch(ctx, cfg, pinned); k = int(f.total); f.free()
        steps = max(1, min(args.steps, 5))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            f = hip.sift_batch(ctx, cfg, pinned); f.free()
        torch.cuda.synchronize(); t = time.perf_counter() - t0
        nbytes = sum(x.nbytes for x in imgs)
        res[key] = {"ms_per_step": t / steps * 1e3, "keypoints_per_s": k * steps / t, "h2d_bytes_per_step": nbytes,
                    "h2d_gb_per_s_floor": nbytes * steps / t / 1e9, "descriptors": k}
    return res


def run_blend(hip, ctx, cfg, inputs, H, W, args, log):
    """ConnectedImages::blend of the rank's images under the homographies of a 2-row camera sweep
    (spherical projection, ESTIMATE_CAMERA mode): LinearBlender as the default config selects
    (MULTIBAND 0) and MultiBandBlender(5).  Inputs resident in HBM; the canvas stays in HBM."""
    from openpano_amd.config import PanoConfig
    n = len(inputs)
    cols = -(-n // 2)
    f = 3.2 * W
    homos = []
    for i in range(n):
        r, c = divmod(i, cols)
        yaw = (c - cols / 2) * 0.55 * W / f; pitch = (r - 0.5) * 0.55 * H / f
        Ry = np.array([[np.cos(yaw), 0, np.sin(yaw)], [0, 1, 0], [-np.sin(yaw), 0, np.cos(yaw)]])
        Rx = np.array([[1, 0, 0], [0, np.cos(pitch), np.sin(pitch)], [0, -np.sin(pitch), np.cos(pitch)]])
        homos.append(Ry @ Rx @ np.diag([1.0 / f, 1.0 / f, 1.0]))
    homos = np.stack(homos)
    res = {}
    for key, over in (("linear", dict(MULTIBAND=0)), ("multiband5", dict(MULTIBAND=5))):
        bcfg = PanoConfig(**over)
        cv = hip.blend(ctx, bcfg, inputs, homos, 2, n // 2); cv.free()          # warm-up
        ctx.set_profiling(True); ctx.profile_reset()
        steps = max(1, min(args.steps, 5))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            cv = hip.blend(ctx, bcfg, inputs, homos, 2, n // 2)
            hw = (cv.h, cv.w); cv.free()
        torch.cuda.synchronize(); t = time.perf_counter() - t0
        prof = {k: v[0] / steps for k, v in ctx.profile().items() if k.startswith(("blend", "multiband"))}
        ctx.set_profiling(False)
        alg = 12.0 * H * W * n + 12.0 * hw[0] * hw[1]            # every source pixel once + canvas write
        if bcfg.MULTIBAND > 0:                                   # ... + 2*16*sum(ROI) per level (WeightedPixel planes)
            g, _, ranges = hip.blend_prepare(bcfg, [(W, H)] * n, homos, 2, n // 2)
            roi = 0
            for r in ranges:
                x0 = int((r[0] - g.proj_min[0]) / g.resolution[0]); y0 = int((r[1] - g.proj_min[1]) / g.resolution[1])
                x1 = int((r[2] - g.proj_min[0]) / g.resolution[0]); y1 = int((r[3] - g.proj_min[1]) / g.resolution[1])
                roi += (x1 - x0 + 1) * (y1 - y0 + 1)
            alg += 2.0 * 16 * roi * bcfg.MULTIBAND
            res_roi = roi
        kms = sum(prof.values())
        res[key] = {"ms_per_blend": t / steps * 1e3, "canvas": [hw[0], hw[1]], "roi_pixels": (res_roi if bcfg.MULTIBAND > 0 else None), "output_mpix_per_s": hw[0] * hw[1] * steps / t / 1e6,
                    "stage_ms": {k: round(v, 4) for k, v in prof.items()},
                    "roofline": {"bound": "hbm", "achieved": alg / (kms * 1e-3) / 1e9 if kms else None, "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": (alg / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS) if kms else None,
                                 "algorithmic_bytes_per_launch": alg, "avg_launch_ms": kms}}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--images", type=int, default=38, help="images per rank (BASELINE config 4: 38)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-match", action="store_true")
    ap.add_argument("--no-blend", action="store_true")
    ap.add_argument("--no-ingest", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the whole-pipeline (ESTIMATE_CAMERA) section")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
        args.gpus = world
    dist = None
    if world > 1 or os.environ.get("OPENPANO_FORCE_DIST"):     # FORCE_DIST: exercise the RCCL path with one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    from openpano_amd import hip, synth
    from openpano_amd.config import PanoConfig
    cfg = PanoConfig()
    H, W = 867, 1300
    nimg = args.images
    # rank r owns views [r*nimg, (r+1)*nimg) of one seeded unordered set (config 4 restated)
    t0 = time.perf_counter()
    views = synth.image_set(nimg, H, W, seed=38 + 1000 * rank, overlap=0.45, rows=2, shuffle=True)
    log(f"synthetic views: {nimg} x {W}x{H} in {time.perf_counter() - t0:.1f} s")
    dev = torch.device("cuda", local_rank)
    d_imgs = [torch.from_numpy(v).to(dev) for v in views]         # inputs resident in HBM
    torch.cuda.synchronize()
    stream = torch.cuda.Stream(device=dev)                         # the stream every kernel is launched on
    torch.cuda.set_stream(stream)
    ctx = hip.Context(local_rank, stream.cuda_stream)
    inputs = [(t.data_ptr(), H, W) for t in d_imgs]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---------------- SIFT loop ----------------
    feats = None
    sift_call = hip.SiftCall(ctx, cfg, inputs)                       # op_image array / op_config marshalled once, like a C host would
    for _ in range(args.warmup):
        if feats is not None:
            feats.free()
        feats = sift_call()
    ctx.set_profiling(True)
    ctx.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if feats is not None:
            feats.free()
        feats = sift_call()
    barrier()
    t_sift = time.perf_counter() - t0
    prof = ctx.profile()
    ctx.set_profiling(False)
    k_rank = int(feats.total)
    tt = torch.tensor([t_sift, float(k_rank)], dtype=torch.float64, device=dev)
    if dist is not None:
        tmax = tt.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tt.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        t_sift_max, k_total = float(tmax[0]), int(tsum[1])
    else:
        t_sift_max, k_total = t_sift, k_rank
    value = k_total * args.steps / t_sift_max

    # ---------------- roofline of the dominant kernel (HIP events, this rank) ----------------
    P, wh, ww = pyramid_pixels(cfg, H, W)
    stage_ms = {k: v[0] / max(args.steps, 1) for k, v in prof.items()}       # device time per batch (a label may bracket several launches)
    dominant = max(stage_ms, key=stage_ms.get) if stage_ms else None
    # algorithmic HBM bytes per launch (DESIGN.md "kernels"), per image:
    alg = {
        # fused scale space + extrema scan: grey in; 6 DoG + 4 Gaussian planes out (DESIGN.md section 3;
        # the scan runs on the DoG layers while they are in LDS and mag/ort are never materialised)
        "build pyramid": 4 * P + 24 * P + 16 * P,
        "resize + octave grey": 12 * H * W + 4 * P,                      # source in, grey octave bases out (working image stays in LDS)
        "sift descriptor": (k_rank / nimg) * (8 * 37 * 37 + 528),        # mag+ort window gathers + output
        "orientation": (k_rank / nimg) * (8 * 16 * 16),
    }
    pmc = {}
    pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc_path):
        try:
            pmc = json.load(open(pmc_path))
        except Exception:
            pmc = {}

    def stage_roofline(name):
        b = alg.get(name)
        dur_s = stage_ms[name] * 1e-3
        ach = (b * nimg / dur_s / 1e9) if b else None
        return {"kernel": name, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (ach / HBM_PEAK_GBS) if ach else None,
                "traffic": pmc.get(name, {}).get("hbm_bytes_per_launch"),      # rocprofv3 PMC, (2*FETCH_SIZE + WRITE_SIZE) * 1024
                "algorithmic_bytes_per_launch": b * nimg if b else None, "avg_launch_ms": stage_ms[name]}

    roofline = stage_roofline(dominant) if dominant is not None else None
    stage_rooflines = {k: stage_roofline(k) for k in stage_ms if k in alg}
    # whole SIFT path against SURVEY 8(d): 12WH + 88P + G + 528K per image
    G = (k_rank / nimg) * 8 * (16 * 16 + 37 * 37)
    b_path = nimg * (12 * H * W + 88 * P + G + 528 * (k_rank / nimg))
    path_gbs = b_path * args.steps / t_sift / 1e9

    out = {
        "metric": "SIFT keypoints+desc/sec and all-pairs matches/sec at 1/2/4/8 GPUs",
        "value": value, "unit": "keypoints+descriptors/s",
        "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": t_sift_max / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE config 4 restated: {nimg} unordered {W}x{H} views per GPU (x{args.gpus} GPUs), "
                               "default config.cfg, inputs resident in HBM",
                   "images_per_gpu": nimg, "image": [H, W], "keypoints_per_image": k_total / (nimg * args.gpus),
                   "parallelism": f"images sharded {nimg}/GPU x {args.gpus}"},
        "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
        "roofline": roofline,
        "stage_rooflines": stage_rooflines,
        "sift_path_roofline": {"bound": "hbm", "achieved": path_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": path_gbs / HBM_PEAK_GBS, "algorithmic_bytes_per_step": b_path},
    }

    # ---------------- all-pairs match loop (+ RANSAC at N=1) ----------------
    args.H, args.W = H, W
    if hasattr(hip, "match_pairs") and not args.no_match:
        from bench_match import run_match_loop
        out["match"] = run_match_loop(hip, ctx, cfg, feats, args, dist, dev, rank, world, barrier, log)
        if out["match"].get("ransac") is not None:
            out["ransac"] = out["match"].pop("ransac")

    # ---------------- final warp + blend of this rank's images (N=1 only: rank 0 renders) ----------------
    if world == 1 and not args.no_blend:
        out["blend"] = run_blend(hip, ctx, cfg, inputs, H, W, args, log)

    # ---------------- host-fed ingest (PCIe inclusive; never `value`) ----------------
    if world == 1 and not args.no_ingest:
        out["ingest"] = run_ingest(hip, ctx, cfg, views, k_rank, args)

    # ---------------- whole Stitcher::build() on rendered rotating-camera views (N=1 only) ----------------
    if world == 1 and not args.no_e2e and not args.no_match and not args.no_blend:
        from bench_e2e import run_e2e
        out["stitch_e2e"] = run_e2e(hip, ctx, args, log)

    # ---------------- CPU baseline (rank 0, N=1 only) ----------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        t0 = time.perf_counter()
        with _StdoutToStderr():
            out["cpu_baseline"] = cpu_baseline(cfg, views, log)
            if out.get("match"):
                out["match"]["cpu_baseline"] = match_cpu_baseline(cfg, feats, log)
        out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        log(f"cpu baseline took {time.perf_counter() - t0:.1f} s")
    elif rank == 0:
        out["cpu_baseline"] = None

    feats.free()
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
