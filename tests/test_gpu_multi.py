"""GPU: the hot path sharded over several devices of ONE process (C-ABI op_group).

The GPU box has one MI355X, so the group lists device 0 twice: two contexts, two host threads, the
image / pair deals, the device-to-device gather and the table replication all run -- only the
copies stay on one device instead of crossing xGMI.  Results must equal the single-device calls."""
import os
import struct
import subprocess

import numpy as np
import pytest

from openpano_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _view(h, w, seed):
    world = synth.make_world(seed, h + 48, w + 48, work_scale=1600.0 / (h + w))
    return synth.cut_view(world, 24, 24, h, w, seed)


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]], ids=["2ctx", "3ctx"])
def test_group_sift_and_match_equal_single_device(cfg, devices):
    from openpano_amd import hip
    imgs = [_view(400, 600, 1), _view(300, 500, 2), _view(400, 600, 3), _view(300, 500, 4), _view(400, 600, 5)]     # mixed sizes, odd count
    ctx = hip.Context(0)
    grp = hip.Group(devices)
    single = hip.sift_batch(ctx, cfg, imgs)
    multi = grp.sift_batch(cfg, imgs)
    assert multi.num_images == single.num_images == 5 and multi.total == single.total
    for i in range(5):
        ds, cs = single.get(i); dm, cm = multi.get(i)
        assert len(ds) > 200 and np.array_equal(ds, dm) and np.array_equal(cs, cm), i
        assert np.array_equal(single.get_real(i), multi.get_real(i))
        assert multi.offset(i) == single.offset(i)
    pairs = [(i, j) for i in range(5) for j in range(i + 1, 5)] + [(3, 1)]
    want = hip.match_pairs_handle(ctx, cfg, single, pairs).lists()
    got = grp.match_pairs_handle(cfg, multi, pairs).lists()          # the table lives on context 0 of the group
    got2 = grp.match_pairs_handle(cfg, single, pairs).lists()        # ... or on any context of the same device
    assert sum(len(m) for m in want) > 100
    for k in range(len(pairs)):
        assert np.array_equal(want[k], got[k]) and np.array_equal(want[k], got2[k]), pairs[k]
    # RANSAC over the group follows the pair deal (lists resident per device) and equals the single-device call,
    # with explicit seeds and with seeds derived from the base seed and the JOB's pair index
    shapes = [(im.shape[1], im.shape[0]) for im in imgs]
    mh1 = hip.match_pairs_handle(ctx, cfg, single, pairs)
    mhg = grp.match_pairs_handle(cfg, multi, pairs)
    for kw in (dict(seeds=[70 + k for k in range(len(pairs))]), dict(base_seed=5)):
        r1 = hip.ransac_pairs(ctx, cfg, single, mh1, pairs, shapes, **kw)
        rg = grp.ransac_pairs(cfg, multi, mhg, pairs, shapes, **kw)
        rh = grp.ransac_pairs(cfg, multi, mh1, pairs, shapes, **kw)      # matched on one device: runs on context 0
        for k in range(len(pairs)):
            for r in (rg, rh):
                assert r[k]["ok"] == r1[k]["ok"] and r[k]["best_hyp"] == r1[k]["best_hyp"] and r[k]["confidence"] == r1[k]["confidence"], pairs[k]
                assert np.array_equal(r[k]["inliers"], r1[k]["inliers"]) and np.array_equal(r[k]["homo"], r1[k]["homo"]), pairs[k]
    mh1.free(); mhg.free()
    # fewer items than devices; one image; no pairs
    one = grp.sift_batch(cfg, imgs[:1])
    assert np.array_equal(one.get(0)[0], single.get(0)[0])
    assert grp.match_pairs_handle(cfg, single, [(0, 1)]).lists()[0].tolist() == want[0].tolist()
    assert grp.match_pairs_handle(cfg, single, []).lists() == []
    one.free(); multi.free(); single.free(); grp.close(); ctx.close()


def test_stitch_demo_over_a_device_group(tmp_path):
    """The standalone C++ program with OPENPANO_DEVICES=0,0: HipSIFTDetector::calc_feature and
    HipPairWiseMatcher shard over the group; the output file is byte-identical to the one-context run."""
    demo = os.path.join(ROOT, "openpano_amd", "host", "stitch_demo")
    views = synth.image_set(5, 240, 320, seed=5, overlap=0.5)
    fin = tmp_path / "in.bin"
    with open(fin, "wb") as f:
        f.write(struct.pack("<3i", 5, 240, 320))
        for v in views:
            f.write(np.ascontiguousarray(v, np.float32).tobytes())
    outs = []
    for tag, devs in (("one", None), ("group", "0,0")):
        env = dict(os.environ); env.pop("LD_PRELOAD", None); env.pop("OPENPANO_DEVICES", None)
        if devs:
            env["OPENPANO_DEVICES"] = devs
        fout = tmp_path / f"out_{tag}.bin"
        r = subprocess.run([demo, str(fin), str(fout), "42"], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(open(fout, "rb").read())
    assert len(outs[0]) > 1000 and outs[0] == outs[1]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_job_overlap_path_on_the_device(cfg, monkeypatch, world):
    """openpano_amd.distributed.ShardedJob(overlap=True) with the PRODUCT engine (HipEngine), one emulated rank after the
    other on the one device of the box: the exchange is replaced by the table the ranks' own SIFT calls add up to (the
    collectives themselves run on gloo ranks in tests/test_distributed_cpu.py and on a one-rank RCCL group in bench.py),
    everything else is the code a rank runs at N > 1 -- own-pairs-first deal, the pairs of two own images matched on a
    table adopted over the rank's own SIFT output (local indices), the rest on the global table, RANSAC over both handles
    with seeds from the global pair ids, ransac_summary -- and must reproduce the single-rank job pair for pair."""
    import torch
    from openpano_amd import hip, distributed as D
    base = synth.make_world(77, 360, 1200, work_scale=1600.0 / (300 + 400), density=900.0)
    imgs = [synth.cut_view(base, 20 + 4 * k, 20 + 110 * k, 300, 400, 70 + k) for k in range(7)]     # a sweep: neighbours overlap
    n = len(imgs)
    shapes = [(400, 300)] * n
    dev = torch.device("cuda", 0)
    ctx = hip.Context(0)
    # the single-rank job
    eng1 = D.HipEngine(ctx, cfg, dev)
    job1 = D.ShardedJob(eng1, n, dev)
    job1.sift(imgs); job1.exchange(); job1.match(); job1.ransac(shapes, base_seed=3)
    want = {p: (m.copy(), r) for p, m, r in zip(job1.my_pairs, job1.lists, job1.rres)}
    gdesc, gcoor, gcounts = job1.desc.clone(), job1.coor.clone(), list(job1.counts)
    ok1, inl1 = job1.ransac_summary(shapes, 3)
    job1.close()
    seen = {}
    tot_ok = tot_inl = 0
    for rank in range(world):
        eng = D.HipEngine(ctx, cfg, dev)
        job = D.ShardedJob(eng, n, dev, overlap=True)
        job.world, job.rank, job.dist = world, rank, True
        job.local_ids = D.shard_images(n, rank, world)
        monkeypatch.setattr(D, "allgatherv_features", lambda d, c, cnt, nn, group=None, wait=True: (gdesc, gcoor, gcounts, []) if not wait else (gdesc, gcoor, gcounts))
        k_local = job.sift([imgs[g] for g in job.local_ids])
        assert k_local == sum(gcounts[g] for g in job.local_ids)
        job.exchange()
        blk = set(job.local_ids)
        own = [p for p in D.all_pairs(n) if p[0] in blk and p[1] in blk]
        assert all(p in job.my_pairs for p in own) and len(job.local_sel) == len(own)
        job.match(); job.ransac(shapes, base_seed=3)
        for p, m, r in zip(job.my_pairs, job.lists, job.rres):
            assert p not in seen
            seen[p] = True
            wm, wr = want[p]
            assert np.array_equal(m, wm), (rank, p)
            assert r["ok"] == wr["ok"] and r["best_hyp"] == wr["best_hyp"] and r["confidence"] == wr["confidence"], (rank, p)
            assert np.array_equal(r["inliers"], wr["inliers"]) and np.array_equal(r["homo"], wr["homo"]), (rank, p)
        a, b = job.ransac_summary(shapes, 3)
        tot_ok += a; tot_inl += b
        job.close()
        if eng._feats is not None:
            eng._feats.free(); eng._feats = None
    assert sorted(seen) == D.all_pairs(n)
    assert (tot_ok, tot_inl) == (ok1, inl1) and ok1 >= 3
    ctx.close()


def test_rehearsal_of_2_4_8_rank_jobs_on_one_device(cfg):
    """bench_match.rehearse: all ranks' shares of the N = 2, 4, 8 strong-scaled jobs, one after the other on this device, through
    ShardedJob(overlap=True, rehearsal=...) with the product engine -- a config-5-shaped job (32 of the 128 4000x3000 uint8 images,
    496 pairs; scripts/scale_rehearsal.py runs config 4 and the whole config 5 the same way and its record is what bench.py --gpus N
    prints as `predicted`).  rehearse() itself asserts that the union of every world's results is the single-rank job (match
    lists by CRC, RANSAC accepted pairs and inliers); here: every world ran, the shares add up, and the per-rank SIFT share
    shrinks with N."""
    import sys
    import torch
    from openpano_amd import hip
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench_match import rehearse
    dev = torch.device("cuda", 0)
    ctx = hip.Context(0)
    res = rehearse(hip, ctx, cfg, "config5", (1, 2, 4, 8), dev, lambda m: None, c5_images=32)
    ctx.close()
    assert res["image_pairs"] == 496 and res["descriptors"] > 32 * 2500 and res["accepted_pairs"] >= 28
    for world in (1, 2, 4, 8):
        w = res["worlds"][str(world)]
        assert w["equals_single_rank_job"] and len(w["per_rank_phase_ms"]) == world
        assert sum(r["pairs"] for r in w["per_rank_phase_ms"]) == 496 and sum(r["images"] for r in w["per_rank_phase_ms"]) == 32
        if world > 1:                      # own pairs first: C(32 / world, 2) per rank
            k = 32 // world
            assert all(r["own_pairs"] == k * (k - 1) // 2 for r in w["per_rank_phase_ms"])
    assert res["worlds"]["8"]["phase_ms"]["sift"] < res["worlds"]["1"]["phase_ms"]["sift"]
