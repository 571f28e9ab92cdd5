"""CPU, world_size 2 over gloo: the descriptor all-gather and the pair partition that the
multi-GPU path uses (RCCL on the GPU box) reconstruct one consistent global job; the whole sharded job at world 2
and at world 3 (ragged shards) against the single-rank job."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from openpano_amd.distributed import allgather_descriptors, all_pairs, partition_pairs, gather_match_results
    rng = np.random.default_rng(100 + rank)
    counts = [5 + rank, 0, 17][: 2 + rank]          # ragged: different image counts per rank, an empty image
    local = torch.from_numpy(rng.random((sum(counts), 128), dtype=np.float32))
    glob, gcounts = allgather_descriptors(local, counts)
    pairs = all_pairs(len(gcounts))
    mine_rr = partition_pairs(pairs, rank, world)
    mine_bal = partition_pairs(pairs, rank, world, gcounts)
    # every rank "matches" its share (a deterministic stand-in: list length and content from the pair)
    fake = [np.array([[i * 7 + k, j * 5 + k] for k in range((i + 2 * j) % 4)], np.int32).reshape(-1, 2) for i, j in mine_bal]
    gathered = gather_match_results(mine_bal, fake, torch.device("cpu"))
    q.put((rank, glob.numpy().copy(), gcounts, local.numpy().copy(), counts, mine_rr, mine_bal,
           {k: v.tolist() for k, v in gathered.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_allgather_and_pair_partition_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, g0, c0, l0, lc0, rr0, b0, m0), (r1, g1, c1, l1, lc1, rr1, b1, m1) = res
    # every rank reconstructs the same global table = rank-ordered concatenation of the shards
    assert c0 == c1 == lc0 + lc1
    assert np.array_equal(g0, g1)
    assert np.array_equal(g0, np.concatenate([l0, l1]))
    n = len(c0)
    allp = [(i, j) for i in range(n) for j in range(i + 1, n)]
    for a, b in ((rr0, rr1), (b0, b1)):
        assert sorted(a + b) == allp and not (set(a) & set(b))      # a partition: each pair exactly once
    cost = lambda ps: sum(c0[i] * c0[j] for i, j in ps)             # noqa: E731
    assert abs(cost(b0) - cost(b1)) <= max(c0) ** 2                 # balanced by K_i*K_j
    # the gathered match results cover every pair exactly once, identically on both ranks
    assert m0 == m1 and sorted(m0) == allp
    for (i, j), v in m0.items():
        assert v == [[i * 7 + k, j * 5 + k] for k in range((i + 2 * j) % 4)]


# ---------------------------------------------------------------------------------------------
# The whole sharded job (openpano_amd.distributed.ShardedJob) on two gloo ranks with REAL work:
# an oracle-backed engine runs SIFT on each rank's image shard, the features go through the same
# bucketed all-gather the GPU path uses, each rank matches + RANSACs its share of the pair list,
# results are gathered -- and must equal the single-rank job item by item.
class OracleEngine:
    """CPU stand-in for HipEngine in tests: same interface, oracle/liboracle.so underneath."""

    def __init__(self, cfg):
        from checkers import Oracle
        self.o = Oracle(cfg)

    def sift(self, images):
        res = [self.o.detect_feature(im) for im in images]
        counts = [len(d) for d, _ in res]
        desc = torch.from_numpy(np.concatenate([d for d, _ in res] + [np.zeros((0, 128), np.float32)]))
        coor = torch.from_numpy(np.concatenate([c for _, c in res] + [np.zeros((0, 2), np.float64)]))
        return desc, coor, counts

    def table(self, desc, coor, counts):
        offs = np.concatenate([[0], np.cumsum(counts)])
        return [(desc[offs[i]: offs[i + 1]].numpy(), coor[offs[i]: offs[i + 1]].numpy()) for i in range(len(counts))]

    def match(self, table, pairs):
        lists = [self.o.match_exact(table[i][0], table[j][0]) for i, j in pairs]
        return lists, lists

    def ransac(self, table, mh, lists, pairs, shapes_wh, seeds):
        return [self.o.ransac(lists[k], table[i][1], table[j][1], shapes_wh[i], shapes_wh[j], seeds[k]) for k, (i, j) in enumerate(pairs)]

    def free(self, obj):
        pass


def _job_views():
    from openpano_amd import synth
    world = synth.make_world(77, 300, 900, work_scale=1600.0 / (200 + 280), density=900.0)
    return [synth.cut_view(world, 20 + 6 * k, 20 + 110 * k, 200, 280, 70 + k) for k in range(5)]


def _run_job(group_world, nviews=5, overlap=False):
    from openpano_amd.config import PanoConfig
    from openpano_amd.distributed import ShardedJob
    views = _job_views()[:nviews]
    job = ShardedJob(OracleEngine(PanoConfig()), len(views), torch.device("cpu"), overlap=overlap)
    assert job.world == group_world
    k_local = job.sift([views[g] for g in job.local_ids])
    k_total = job.exchange()
    job.match()
    job.ransac([(280, 200)] * len(views), base_seed=9)
    res = job.gather()
    job.close()
    return k_local, k_total, job.gcounts, job.my_pairs, {k: (m.tolist(), ex.tolist()) for k, (m, ex) in res.items()}


def _job_worker(rank, world, port, q, nviews=5, overlap=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    q.put((rank,) + _run_job(world, nviews, overlap))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True], ids=["exchange_then_match", "own_pairs_during_exchange"])
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_job_equals_single_rank(world, overlap):
    """world 2: the even deal; world 3: 5 images over 3 ranks (2 + 2 + 1) and 10 pairs (ragged shards, a rank
    with a single image) -- every rank must end with the single-rank job's results.  overlap: the pairs of two own
    images are matched (on the rank's own features, local indices) before the exchange is waited for."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    single = _run_job(1)                     # no process group: the world-1 path of ShardedJob
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_job_worker, args=(r, world, port, q, 5, overlap)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    k1, ktot1, counts1, pairs1, out1 = single
    assert k1 == ktot1 and len(pairs1) == 10 and min(counts1) > 30
    assert sum(r[1] for r in res) == ktot1                # SIFT sharded by image, nothing lost
    assert all(r[2] == ktot1 and r[3] == counts1 for r in res)   # the exchanged table is in global image order
    allp = [p for r in res for p in r[4]]
    assert sorted(allp) == pairs1 and len(set(allp)) == len(allp)         # a partition of the pair list
    assert all(len(r[4]) > 0 for r in res)
    if overlap:                                           # every pair of two images of one rank's block is that rank's
        from openpano_amd.distributed import shard_images
        for r in res:
            blk = set(shard_images(5, r[0], world))
            assert all(p in r[4] for p in pairs1 if p[0] in blk and p[1] in blk)
    oa = res[0][5]
    assert all(r[5] == oa for r in res)                   # every rank holds the whole job after the gather
    assert sorted(oa) == sorted(out1)
    nok = 0
    for p in out1:                                        # match sets, RANSAC decision / confidence / homography: identical
        assert oa[p][0] == out1[p][0], p
        assert np.array_equal(np.array(oa[p][1]), np.array(out1[p][1]), equal_nan=True), p
        nok += out1[p][1][0] > 0
    assert nok >= 3


def test_more_ranks_than_images():
    """world 3, two images: the third rank owns no image (an empty shard goes through every collective without a
    library call) and no pair; the job still equals the single-rank job on every rank."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    single = _run_job(1, 2)
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_job_worker, args=(r, world, port, q, 2)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res][2] == 0 and sum(r[1] for r in res) == single[1]
    assert sum(len(r[4]) for r in res) == 1
    assert all(r[5] == res[0][5] for r in res) and sorted(res[0][5]) == sorted(single[4])
    for p in single[4]:
        assert res[0][5][p][0] == single[4][p][0]


@pytest.mark.parametrize("world", [2, 4, 8])
def test_rehearsal_mode_equals_single_rank(world):
    """ShardedJob(rehearsal=(rank, world, table)): every rank of a world-N job run one after the other in ONE process
    (no process group; the peers' slices are copied out of the whole-job table).  It is the code path bench_match.rehearse
    and tests/test_gpu_multi.py use to run all N = 2, 4, 8 shares of a job on a single device: block ownership, own pairs
    first, global seeds -- the union of the ranks' results must be the single-rank job, pair for pair.  World 8 over 5
    images leaves three ranks without an image."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from openpano_amd.config import PanoConfig
    from openpano_amd.distributed import ShardedJob, all_pairs, shard_images
    views = _job_views()
    n = len(views)
    shapes = [(280, 200)] * n
    eng = OracleEngine(PanoConfig())
    one = ShardedJob(eng, n, torch.device("cpu"))
    one.sift(views); one.exchange(); one.match(); one.ransac(shapes, base_seed=9)
    want = {p: (m.tolist(), r) for p, m, r in zip(one.my_pairs, one.lists, one.rres)}
    table = (one.desc.clone(), one.coor.clone(), list(one.counts))
    seen = set()
    for rank in range(world):
        job = ShardedJob(eng, n, torch.device("cpu"), overlap=True, rehearsal=(rank, world, table))
        assert job.local_ids == shard_images(n, rank, world) and not job.dist
        job.sift([views[g] for g in job.local_ids])
        assert job.exchange() == sum(table[2])
        assert torch.equal(job._keep[0], table[0]) and torch.equal(job._keep[1], table[1])
        job.match(); job.ransac(shapes, base_seed=9)
        for p, m, r in zip(job.my_pairs, job.lists, job.rres):
            assert p not in seen
            seen.add(p)
            assert m.tolist() == want[p][0], (rank, p)
            w = want[p][1]
            assert r["ok"] == w["ok"] and r["best_hyp"] == w["best_hyp"] and r["confidence"] == w["confidence"], (rank, p)
            assert np.array_equal(r["inliers"], w["inliers"]) and np.array_equal(r["homo"], w["homo"], equal_nan=True), (rank, p)
        job.close()
    assert sorted(seen) == all_pairs(n)


def test_native_pair_deal_equals_the_python_deal():
    """partition_pairs' balanced deal runs in C when libpano_host.so is built (pano_deal_pairs: the Python loop took
    6-19 ms for the 8128 pairs of a 128-image job, inside every rank's exchange): same deal, item for item, for every
    rank, with and without the own-pairs-first blocks, on ragged keypoint counts incl. empty images."""
    import numpy as np
    from openpano_amd import distributed as D
    if not D._host_lib():
        import pytest
        pytest.skip("libpano_host.so not built")
    rng = np.random.default_rng(5)
    native = D._HOST_LIB
    try:
        for n, world in ((38, 8), (38, 2), (128, 8), (11, 3), (7, 4), (5, 8), (2, 2)):
            counts = [int(x) for x in rng.integers(0, 5000, n)]
            counts[rng.integers(0, n)] = 0
            blocks = [D.shard_images(n, r, world) for r in range(world)]
            pairs = D.all_pairs(n)
            seen = []
            for rank in range(world):
                for bl in (None, blocks):
                    D._HOST_LIB = native
                    a = D.partition_pairs(pairs, rank, world, counts, bl)
                    D._HOST_LIB = False
                    b = D.partition_pairs(pairs, rank, world, counts, bl)
                    assert a == b, (n, world, rank, bl is not None)
                seen += a
            assert sorted(seen) == pairs            # every pair dealt exactly once
    finally:
        D._HOST_LIB = native
