"""Multi-GPU exchange step of the hot path.

SIFT shards by image with no communication.  All-pairs matching needs every image's
descriptors on every rank: ONE all-gather (RCCL over xGMI on GPUs, gloo in the CPU tests) of the
per-image counts followed by one of the (padded) descriptor payload; the unordered pair list of
``Stitcher::pairwise_match`` (stitch/stitcher.cc:100) is then dealt round-robin to the ranks.
torch.distributed is plumbing here: tensors in, tensors out, no model code.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def allgather_descriptors(local_desc: torch.Tensor, local_counts, group=None):
    """local_desc: (sum(local_counts), 128) float32 on this rank's device (images back to back).

    Returns (global_desc, global_counts): the same layout for the images of rank 0, 1, ... in
    rank order -- identical on every rank.
    """
    world = dist.get_world_size(group)
    dev = local_desc.device
    cnt = torch.as_tensor(list(local_counts), dtype=torch.int64, device=dev)
    ncnt = torch.tensor([cnt.numel()], dtype=torch.int64, device=dev)
    all_n = [torch.empty_like(ncnt) for _ in range(world)]
    dist.all_gather(all_n, ncnt, group=group)
    nmax = int(max(int(x) for x in all_n))
    cnt_pad = torch.zeros(nmax, dtype=torch.int64, device=dev)
    cnt_pad[: cnt.numel()] = cnt
    all_cnt = [torch.empty_like(cnt_pad) for _ in range(world)]
    dist.all_gather(all_cnt, cnt_pad, group=group)
    per_rank = [c[: int(n)] for c, n in zip(all_cnt, all_n)]
    totals = [int(c.sum()) for c in per_rank]
    mx = max(max(totals), 1)
    pad = torch.zeros((mx, 128), dtype=torch.float32, device=dev)
    pad[: local_desc.shape[0]] = local_desc
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    glob = torch.cat([o[:t] for o, t in zip(out, totals)], dim=0).contiguous()
    counts = [int(v) for c in per_rank for v in c.tolist()]
    return glob, counts


def all_pairs(n: int):
    """The unordered pair list of Stitcher::pairwise_match (stitcher.cc:100)."""
    return [(i, j) for i in range(n) for j in range(i + 1, n)]


def partition_pairs(pairs, rank: int, world: int, counts=None):
    """Deal pairs to ranks.  With ``counts`` the deal is balanced by K_i*K_j (longest first,
    greedy); without, plain round-robin.  Deterministic and identical on every rank."""
    if counts is None:
        return pairs[rank::world]
    cost = [counts[i] * counts[j] for i, j in pairs]
    order = sorted(range(len(pairs)), key=lambda k: (-cost[k], k))
    load = [0] * world
    mine = []
    for k in order:
        r = min(range(world), key=lambda q: (load[q], q))
        load[r] += cost[k]
        if r == rank:
            mine.append(pairs[k])
    return sorted(mine)


def gather_match_results(pairs, match_lists, device, group=None):
    """Step 3 of SURVEY 8(e): every rank matched its share of the pair list; the index pairs (a few
    KB per image pair) are collected so that rank 0 -- which runs the host-only camera estimation
    and the blend -- holds the whole job.  pairs: this rank's (i, j) list; match_lists: one (M, 2)
    int32 array per pair.  Returns {(i, j): (M, 2) array} of ALL ranks (identical on every rank: the
    exchange is an all-gather, cheap at this size and free of a root bottleneck over xGMI)."""
    import numpy as np
    world = dist.get_world_size(group)
    # header: (i, j, count) per pair, then the flat index pairs
    hdr = np.array([[i, j, len(m)] for (i, j), m in zip(pairs, match_lists)], np.int64).reshape(-1, 3)
    flat = np.concatenate([np.asarray(m, np.int64).reshape(-1, 2) for m in match_lists] + [np.zeros((0, 2), np.int64)])
    sizes = torch.tensor([hdr.shape[0], flat.shape[0]], dtype=torch.int64, device=device)
    all_sizes = [torch.empty_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=group)
    mh = max(1, max(int(s[0]) for s in all_sizes)); mf = max(1, max(int(s[1]) for s in all_sizes))
    buf = torch.zeros((mh * 3 + mf * 2,), dtype=torch.int64, device=device)
    buf[: hdr.size] = torch.from_numpy(hdr.reshape(-1)).to(device)
    buf[mh * 3: mh * 3 + flat.size] = torch.from_numpy(flat.reshape(-1)).to(device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    res = {}
    for o, s in zip(out, all_sizes):
        o = o.cpu().numpy()
        h = o[: int(s[0]) * 3].reshape(-1, 3); f = o[mh * 3: mh * 3 + int(s[1]) * 2].reshape(-1, 2)
        at = 0
        for i, j, c in h:
            res[(int(i), int(j))] = f[at: at + int(c)].astype(np.int32)
            at += int(c)
    return res
