"""CPU, world_size 2 over gloo: the descriptor all-gather and the pair partition that the
multi-GPU path uses (RCCL on the GPU box) reconstruct one consistent global job."""
import os
import socket

import numpy as np
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
