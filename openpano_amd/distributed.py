"""Multi-GPU exchange step and sharded job of the hot path.

One process per GPU.  SIFT shards by image with no communication (stitcherbase.cc:14 is the axis): rank r
owns a contiguous block of the image list, so its features are one contiguous slice of the image-indexed table.
All-pairs matching needs every image's features on every rank: after a tiny all-gather of the per-image
counts every rank allocates the table in GLOBAL image order and the shards travel as an all-gather-v -- one
grouped send/recv series (RCCL over xGMI on GPUs: every rank pushes its slice to the other ranks over its
own point-to-point links; gloo in the CPU tests) straight INTO the table's slices: no bucket padding, no
concatenation, no reorder, and the library adopts the table without a copy.  The unordered pair list
of ``Stitcher::pairwise_match`` (stitch/stitcher.cc:100) is dealt to the ranks balanced by
``K_i * K_j``; RANSAC follows the pair partition; the per-pair results (KBs) are all-gathered so
that rank 0 -- which runs the host-only camera estimation and the blend -- holds the whole job.
torch.distributed is plumbing here: tensors in, tensors out, no model code.

``ShardedJob`` is engine-agnostic: the product engine is ``HipEngine`` (the C-ABI library); the
CPU tests drive the same exchange / partition / gather code with an oracle-backed engine.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


# ----------------------------------------------------------------------------- exchange
def allgather_features(local_desc: torch.Tensor, local_coor: torch.Tensor | None, local_counts, group=None):
    """local_desc: (K, 128) float32, local_coor: (K, 2) float64 or None, on this rank's device,
    images back to back; local_counts: descriptors per local image.

    Returns (global_desc, global_coor | None, global_counts, images_per_rank): the same layout
    for the images of rank 0, 1, ... in rank order -- identical on every rank.  Two collectives:
    counts (N x int64, padded), then one byte bucket per rank padded to the largest shard."""
    world = dist.get_world_size(group)
    dev = local_desc.device
    cnt = torch.as_tensor(list(local_counts), dtype=torch.int64, device=dev)
    k_local = int(local_desc.shape[0])
    assert int(cnt.sum()) == k_local
    # header: [n_images, K, counts...] padded to a common length (image counts may differ by one)
    nmax_t = torch.tensor([cnt.numel()], dtype=torch.int64, device=dev)
    dist.all_reduce(nmax_t, op=dist.ReduceOp.MAX, group=group)
    nmax = int(nmax_t)
    hdr = torch.zeros(nmax + 2, dtype=torch.int64, device=dev)
    hdr[0] = cnt.numel(); hdr[1] = k_local; hdr[2: 2 + cnt.numel()] = cnt
    all_hdr = torch.empty(world * (nmax + 2), dtype=torch.int64, device=dev)        # flat: gloo chunks along dim 0
    dist.all_gather_into_tensor(all_hdr, hdr, group=group)
    all_hdr = all_hdr.view(world, nmax + 2).cpu()
    nimg = [int(all_hdr[r, 0]) for r in range(world)]
    totals = [int(all_hdr[r, 1]) for r in range(world)]
    counts = [int(v) for r in range(world) for v in all_hdr[r, 2: 2 + nimg[r]].tolist()]
    per_kp = 128 * 4 + (16 if local_coor is not None else 0)
    kmax = max(max(totals), 1)
    bucket = torch.zeros(kmax * per_kp, dtype=torch.uint8, device=dev)
    if k_local:
        bucket[: k_local * 512] = local_desc.contiguous().view(torch.uint8).reshape(-1)
        if local_coor is not None:
            bucket[kmax * 512: kmax * 512 + k_local * 16] = local_coor.contiguous().view(torch.uint8).reshape(-1)
    out = torch.empty(world * kmax * per_kp, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(out, bucket, group=group)
    out = out.view(world, kmax * per_kp)
    descs, coors = [], []
    for r in range(world):
        k = totals[r]
        descs.append(out[r, : k * 512].view(torch.float32).reshape(k, 128))
        if local_coor is not None:
            coors.append(out[r, kmax * 512: kmax * 512 + k * 16].view(torch.float64).reshape(k, 2))
    gdesc = torch.cat(descs, 0).contiguous()
    gcoor = torch.cat(coors, 0).contiguous() if local_coor is not None else None
    return gdesc, gcoor, counts, nimg


def allgather_descriptors(local_desc: torch.Tensor, local_counts, group=None):
    """Descriptor-only form of ``allgather_features`` -> (global_desc, global_counts)."""
    g, _, counts, _ = allgather_features(local_desc, None, local_counts, group)
    return g, counts


def all_pairs(n: int):
    """The unordered pair list of Stitcher::pairwise_match (stitcher.cc:100)."""
    return [(i, j) for i in range(n) for j in range(i + 1, n)]


_HOST_LIB = None


def _host_lib():
    """libpano_host.so (host-only C, no HIP) when it is built, else False: the deal below then runs in Python"""
    global _HOST_LIB
    if _HOST_LIB is None:
        import ctypes as C
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpano_host.so")
        try:
            L = C.CDLL(path)
            L.pano_deal_pairs.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
            _HOST_LIB = L
        except (OSError, AttributeError):
            _HOST_LIB = False
    return _HOST_LIB


def _deal_native(pairs, rank, world, counts, blocks):
    """partition_pairs' balanced deal in C (pano_deal_pairs, host/pano_host_capi.cc): the same deal, item for item
    (tests/test_distributed_cpu.py compares the two), at microseconds instead of the 6-19 ms the loop below takes for the
    8128 pairs of a 128-image job -- on every rank, inside every exchange.  None when the library is not there."""
    L = _host_lib()
    if not L or not pairs:
        return None
    pr = np.ascontiguousarray(np.asarray(pairs, np.int32).reshape(-1, 2))
    cnt = np.asarray(counts, np.int64)
    nimg = len(cnt)
    if pr.min() < 0 or pr.max() >= nimg:
        return None
    cost = np.ascontiguousarray(cnt[pr[:, 0]] * cnt[pr[:, 1]])
    owner = None
    if blocks is not None:
        owner = np.full(nimg, -1, np.int32)
        for r, b in enumerate(blocks):
            for g in b:
                if 0 <= g < nimg:
                    owner[g] = r
    mine = np.zeros(len(pr), np.uint8)
    rc = L.pano_deal_pairs(len(pr), pr.ctypes.data, cost.ctypes.data, world, rank, owner.ctypes.data if owner is not None else None, nimg, mine.ctypes.data)
    if rc != 0:
        return None
    return sorted(pairs[k] for k in np.flatnonzero(mine))


def partition_pairs(pairs, rank: int, world: int, counts=None, blocks=None):
    """Deal pairs to ranks.  With ``counts`` the deal is balanced by K_i*K_j (longest first,
    greedy); without, plain round-robin.  Deterministic and identical on every rank.
    ``blocks`` (image ids owned by every rank): a pair whose two images one rank owns goes to that rank first --
    it can be matched while the feature exchange is still in flight -- and the rest is dealt on top of that load."""
    if counts is None:
        return pairs[rank::world]
    dealt = _deal_native(pairs, rank, world, counts, blocks)
    if dealt is not None:
        return dealt
    cost = [counts[i] * counts[j] for i, j in pairs]
    load = [0] * world
    mine = []
    rest = range(len(pairs))
    if blocks is not None:
        owner = {}
        for r, b in enumerate(blocks):
            for g in b:
                owner[g] = r
        rest = []
        for k, (i, j) in enumerate(pairs):
            if owner.get(i, -1) == owner.get(j, -2):
                load[owner[i]] += cost[k]
                if owner[i] == rank:
                    mine.append(pairs[k])
            else:
                rest.append(k)
    for k in sorted(rest, key=lambda k: (-cost[k], k)):
        r = min(range(world), key=lambda q: (load[q], q))
        load[r] += cost[k]
        if r == rank:
            mine.append(pairs[k])
    return sorted(mine)


def shard_images(n: int, rank: int, world: int):
    """Global image ids owned by ``rank``: a contiguous block of n/world (+1 on the first n % world ranks) images --
    images of one job have one size, so blocks are as balanced as any deal, and a rank's features are ONE slice of
    the image-indexed table (one message per peer in the exchange)."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return list(range(start, start + base + (1 if rank < rem else 0)))


def allgatherv_features(local_desc: torch.Tensor, local_coor: torch.Tensor, local_counts, n: int, group=None, wait=True):
    """The exchange step.  local_desc (K, 128) float32 / local_coor (K, 2) float64: this rank's images (the block
    ``shard_images(n, rank, world)``) back to back on its device.  Returns (desc, coor, counts) of ALL n images in
    global image order, identical on every rank: a header collective (per-image counts), then the slices are
    sent / received in place with one grouped batch of point-to-point operations -- ncclGroupStart/End under the
    nccl backend, i.e. every pair of GPUs exchanges its two slices over its own xGMI link concurrently.
    ``wait=False`` returns (desc, coor, counts, pending works) right after posting: the own slice of the table is
    already in place, the peers' slices are valid once every work has been waited for."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = local_desc.device
    blocks = [shard_images(n, r, world) for r in range(world)]
    assert len(local_counts) == len(blocks[rank])
    nmax = max(max(len(b) for b in blocks), 1)
    hdr = torch.zeros(nmax, dtype=torch.int64, device=dev)
    if len(local_counts):
        hdr[: len(local_counts)] = torch.as_tensor(list(local_counts), dtype=torch.int64)
    all_hdr = torch.empty(world * nmax, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(all_hdr, hdr, group=group)
    all_hdr = all_hdr.view(world, nmax).cpu()
    counts = [int(all_hdr[r, k]) for r in range(world) for k in range(len(blocks[r]))]
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    lo = [int(offs[b[0]]) if b else 0 for b in blocks]
    hi = [int(offs[b[-1] + 1]) if b else 0 for b in blocks]
    total = int(offs[-1])
    assert hi[rank] - lo[rank] == int(local_desc.shape[0])
    gdesc = torch.empty((total, 128), dtype=torch.float32, device=dev)
    gcoor = torch.empty((total, 2), dtype=torch.float64, device=dev)
    if hi[rank] > lo[rank]:
        gdesc[lo[rank]: hi[rank]].copy_(local_desc)
        gcoor[lo[rank]: hi[rank]].copy_(local_coor)
    ops = []
    mine_d, mine_c = gdesc[lo[rank]: hi[rank]], gcoor[lo[rank]: hi[rank]]
    for step in range(1, world):                          # peer order rotated by rank: no hot receiver
        peer = (rank + step) % world
        src = (rank - step) % world
        if hi[rank] > lo[rank]:
            ops.append(dist.P2POp(dist.isend, mine_d, peer, group))
            ops.append(dist.P2POp(dist.isend, mine_c, peer, group))
        if hi[src] > lo[src]:
            ops.append(dist.P2POp(dist.irecv, gdesc[lo[src]: hi[src]], src, group))
            ops.append(dist.P2POp(dist.irecv, gcoor[lo[src]: hi[src]], src, group))
    works = dist.batch_isend_irecv(ops) if ops else []
    if not wait:                                         # the caller overlaps work on what it already owns and waits itself
        return gdesc, gcoor, counts, works
    for w in works:
        w.wait()
    if gdesc.is_cuda:
        # wait() orders torch's current stream behind the receives (and the own-slice copy runs on it); the consumers
        # are library calls on another stream: make the table valid for every stream before handing it out
        torch.cuda.current_stream(gdesc.device).synchronize()
    return gdesc, gcoor, counts


def _allgather_blob(blob: np.ndarray, device, group=None):
    """variable-length int64 arrays of every rank -> list (rank order); one size + one payload collective"""
    world = dist.get_world_size(group)
    size = torch.tensor([blob.size], dtype=torch.int64, device=device)
    sizes = torch.empty(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(sizes, size, group=group)
    sizes = sizes.cpu().tolist()
    mx = max(max(sizes), 1)
    buf = torch.zeros(mx, dtype=torch.int64, device=device)
    if blob.size:
        buf[: blob.size] = torch.from_numpy(np.ascontiguousarray(blob, np.int64)).to(device)
    out = torch.empty(world * mx, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, buf, group=group)
    out = out.view(world, mx).cpu().numpy()
    return [out[r, : sizes[r]] for r in range(world)]


def gather_match_results(pairs, match_lists, device, group=None, extras=None):
    """Step 3 of SURVEY 8(e): every rank matched its share of the pair list; the index pairs (a few
    KB per image pair) are collected so that rank 0 -- which runs the host-only camera estimation
    and the blend -- holds the whole job.  pairs: this rank's (i, j) list; match_lists: one (M, 2)
    int32 array per pair; extras: optional per-pair float64 vectors of equal length (the RANSAC
    result: ok, confidence, 9 homography entries, ...) carried bit-exactly as int64 views.
    Returns {(i, j): (M, 2) array} or {(i, j): ((M, 2) array, extra)} of ALL ranks (identical on
    every rank: an all-gather, cheap at this size and free of a root bottleneck over xGMI)."""
    ne = len(extras[0]) if extras else 0
    hdr = np.array([[i, j, len(m)] for (i, j), m in zip(pairs, match_lists)], np.int64).reshape(-1, 3)
    flat = np.concatenate([np.asarray(m, np.int64).reshape(-1) for m in match_lists] + [np.zeros(0, np.int64)])
    ex = np.concatenate([np.ascontiguousarray(e, np.float64).view(np.int64) for e in extras] + [np.zeros(0, np.int64)]) if extras else np.zeros(0, np.int64)
    blob = np.concatenate([np.array([hdr.shape[0], ne], np.int64), hdr.reshape(-1), flat, ex])
    res = {}
    for b in _allgather_blob(blob, device, group):
        npair, ne_r = int(b[0]), int(b[1])
        h = b[2: 2 + 3 * npair].reshape(-1, 3)
        nflat = int(h[:, 2].sum()) * 2 if npair else 0
        f = b[2 + 3 * npair: 2 + 3 * npair + nflat].reshape(-1, 2)
        e = b[2 + 3 * npair + nflat:].view(np.float64).reshape(npair, ne_r) if ne_r else None
        at = 0
        for k, (i, j, c) in enumerate(h):
            m = f[at: at + int(c)].astype(np.int32)
            res[(int(i), int(j))] = (m, e[k].copy()) if e is not None else m
            at += int(c)
    return res


# ----------------------------------------------------------------------------- engines
class HipEngine:
    """The product engine: libopenpano_hip.so through the C-ABI (openpano_amd/hip.py)."""

    def __init__(self, ctx, cfg, device):
        from . import hip
        self.hip, self.ctx, self.cfg, self.device = hip, ctx, cfg, device
        self._feats = None

    def sift(self, images):
        """-> (desc (K,128) f32 tensor, coor (K,2) f64 tensor, counts); zero-copy views of the
        library-owned buffers, valid until the next sift() of this engine.  An empty shard (more ranks than
        images) is no library call."""
        if self._feats is not None:
            self._feats.free(); self._feats = None
        if not callable(images) and len(images) == 0:
            return torch.zeros((0, 128), device=self.device), torch.zeros((0, 2), dtype=torch.float64, device=self.device), []
        f = self._feats = self.hip.sift_batch(self.ctx, self.cfg, images) if not callable(images) else images()
        counts = [f.count(i) for i in range(f.num_images)]
        if int(f.total) == 0:
            return torch.zeros((0, 128), device=self.device), torch.zeros((0, 2), dtype=torch.float64, device=self.device), counts
        desc = torch.as_tensor(f.desc_device_array(), device=self.device)
        coor = torch.as_tensor(f.coor_device_array(), device=self.device)
        return desc, coor, counts

    def table(self, desc, coor, counts):
        """image-indexed feature table in this rank's HBM: the library adopts the exchanged buffers (no copy)"""
        if int(desc.shape[0]) == 0:                     # a job without a single descriptor: the library wants real pointers
            desc = torch.zeros((1, 128), dtype=torch.float32, device=self.device)
            coor = torch.zeros((1, 2), dtype=torch.float64, device=self.device)
        return self.hip.Features.adopt_device(self.ctx, desc.data_ptr(), counts, coor.data_ptr(), keep=(desc, coor))

    def match(self, table, pairs):
        """-> (handle, [(M,2) int32])"""
        mh = self.hip.match_pairs_handle(self.ctx, self.cfg, table, pairs)
        return mh, mh.lists()

    def match_only(self, table, pairs):
        self.hip.match_pairs_handle(self.ctx, self.cfg, table, pairs).free()

    def ransac(self, table, mh, lists, pairs, shapes_wh, seeds):
        return self.hip.ransac_pairs(self.ctx, self.cfg, table, mh, pairs, shapes_wh, seeds=seeds)

    def ransac_summary(self, table, mh, pairs, shapes_wh, seeds):
        """op_ransac_pairs without unpacking every pair into Python -> (accepted pairs, inliers)"""
        return self.hip.ransac_pairs_summary(self.ctx, self.cfg, table, mh, pairs, shapes_wh, seeds=seeds)

    def concat(self, mh_a, mh_b):
        """two match handles of this device as one (a's pairs, then b's): one RANSAC call instead of two"""
        return self.hip.Matches.concat(self.ctx, mh_a, mh_b)

    def free(self, obj):
        obj.free()


def ransac_extra(r):
    """RANSAC result of one pair -> fixed-length float64 vector (carried by gather_match_results)"""
    return np.concatenate([[float(r["ok"]), float(r["confidence"]), float(r["best_hyp"]), float(r["best_count"])], np.asarray(r["homo"], np.float64).reshape(9)])


class ShardedJob:
    """One stitching job (n images, all unordered pairs) sharded over the ranks of ``group``.

    Phases (each can be timed separately by the caller):
      sift(local_images)  -> this rank's features;        no communication
      exchange()          -> global feature table on every rank;  the one all-gather
      match()             -> this rank's share of the pair list
      ransac(shapes, seed)-> TransformEstimation on the same share
      gather()            -> {(gi, gj): (matches, ransac vector)} of the whole job on every rank
    Image ids are GLOBAL (0..n-1); rank r owns ``shard_images(n, r, world)``; the exchanged table is
    in global order and per-pair RANSAC seeds derive from the ids, so every result is independent of
    the world size (tests/test_distributed_cpu.py compares world 2 and 3 with world 1).

    ``overlap=True``: the exchange is posted and NOT waited for; the pairs whose two images this rank owns (they are dealt
    to it first, ``partition_pairs(blocks=...)``) are matched on its own features while the slices of the other ranks
    travel, and only then the exchange is waited for -- ``exchange()`` returns with the local pairs already matched,
    ``match()`` adds the rest.  Same pair list per job, same results per pair."""

    def __init__(self, engine, n_images: int, device, group=None, overlap=False, rehearsal=None):
        self.e = engine
        self.n = n_images
        self.device = device
        self.group = group
        self.dist = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.dist else 0
        self.world = dist.get_world_size(group) if self.dist else 1
        # ``rehearsal=(rank, world, (desc, coor, counts))``: ONE rank of a ``world``-rank job run on its own, without a
        # process group -- the slices of the other ranks are copied out of the given whole-job table (device-to-device copies
        # standing in for the xGMI transfers) instead of being received.  Everything else is the code a rank runs at
        # N > 1: block ownership, own-pairs-first deal, local match during the exchange, global seeds.  Used to run all
        # ranks' shares of a job one after the other on a single device (tests/test_gpu_multi.py, bench_match.rehearse).
        self.rehearsal = rehearsal
        if rehearsal is not None:
            self.rank, self.world, self.dist = int(rehearsal[0]), int(rehearsal[1]), False
        self.local_ids = shard_images(n_images, self.rank, self.world)
        self.overlap = bool(overlap)
        self.tab = None; self.mh = None
        self.local_tab = None; self.local_mh = None; self.local_lists = None

    def sift(self, local_images):
        assert callable(local_images) or len(local_images) == len(self.local_ids)
        # a single-rank table ADOPTS the engine's feature buffers, which the next sift() frees: results of the previous
        # generation go first, nothing may match against memory that is back in the pool
        self._free_results()
        self.desc, self.coor, self.counts = self.e.sift(local_images)
        return sum(self.counts)

    def adopt(self, feats):
        """use an existing op_features of this rank's shard (HipEngine only; zero-copy views)"""
        self.counts = [feats.count(i) for i in range(feats.num_images)]
        assert len(self.counts) == len(self.local_ids)
        self.desc = torch.as_tensor(feats.desc_device_array(), device=self.device) if int(feats.total) else torch.zeros((0, 128), device=self.device)
        self.coor = torch.as_tensor(feats.coor_device_array(), device=self.device) if int(feats.total) else torch.zeros((0, 2), dtype=torch.float64, device=self.device)
        self._adopted = feats
        return sum(self.counts)

    def _free_results(self):
        for name in ("mh", "local_mh", "tab", "local_tab"):
            obj = getattr(self, name, None)
            if obj is not None:
                self.e.free(obj); setattr(self, name, None)
        self.local_lists = None

    def exchange(self):
        # the previous table goes first: its buffers are what the allocator hands out again for the new one (a second
        # generation of a 265 MB table costs two device allocations of several milliseconds each)
        self._free_results()
        self._keep = None
        works = []
        if self.rehearsal is not None:
            wdesc, wcoor, gcounts = self.rehearsal[2]
            gcounts = list(gcounts)
            gdesc = torch.empty_like(wdesc); gcoor = torch.empty_like(wcoor)
            lo = sum(gcounts[: self.local_ids[0]]) if self.local_ids else 0
            hi = lo + sum(self.counts)
            assert [gcounts[g] for g in self.local_ids] == list(self.counts), "rehearsal: this rank's SIFT output differs from the whole-job table"
            if hi > lo:
                gdesc[lo:hi].copy_(self.desc); gcoor[lo:hi].copy_(self.coor)          # own slice, as allgatherv_features does
            torch.cuda.current_stream(gdesc.device).synchronize() if gdesc.is_cuda else None
            # the peers' slices "arrive" after the own pairs have been matched (overlap) -- see below
            self._rehearsal_fill = (wdesc, wcoor, lo, hi)
        elif self.dist:                       # also with ONE rank (OPENPANO_FORCE_DIST): the header collective really runs
            # slices arrive in place, in GLOBAL image order: pair (i, j), its match list and its RANSAC draw
            # sequence are those of the single-rank job whatever the world size
            if self.overlap:
                gdesc, gcoor, gcounts, works = allgatherv_features(self.desc, self.coor, self.counts, self.n, self.group, wait=False)
            else:
                gdesc, gcoor, gcounts = allgatherv_features(self.desc, self.coor, self.counts, self.n, self.group)
        else:
            gdesc, gcoor, gcounts = self.desc, self.coor, self.counts
        self._keep = (gdesc, gcoor)
        self.gcounts = gcounts
        blocks = [shard_images(self.n, r, self.world) for r in range(self.world)] if self.overlap else None
        self.my_pairs = partition_pairs(all_pairs(self.n), self.rank, self.world, gcounts if self.world > 1 else None, blocks if self.world > 1 else None)
        # pairs of two own images, in local indices: matched NOW, on the features this rank made, while the others' slices travel
        first = self.local_ids[0] if self.local_ids else 0
        own = set(self.local_ids)
        self.local_sel = [k for k, (i, j) in enumerate(self.my_pairs) if i in own and j in own] if (self.overlap and self.world > 1) else []
        if self.local_sel:
            self.local_tab = self.e.table(self.desc, self.coor, self.counts)
            self.local_pairs = [(self.my_pairs[k][0] - first, self.my_pairs[k][1] - first) for k in self.local_sel]
            self.local_mh, self.local_lists = self.e.match(self.local_tab, self.local_pairs)
        if self.rehearsal is not None:
            wdesc, wcoor, lo, hi = self._rehearsal_fill
            gdesc[:lo].copy_(wdesc[:lo]); gdesc[hi:].copy_(wdesc[hi:])
            gcoor[:lo].copy_(wcoor[:lo]); gcoor[hi:].copy_(wcoor[hi:])
            if gdesc.is_cuda:
                torch.cuda.current_stream(gdesc.device).synchronize()
            self._rehearsal_fill = None
        for w in works:
            w.wait()
        if works and gdesc.is_cuda:
            # w.wait() only makes torch's CURRENT stream wait for the NCCL receives; the library adopts the table and
            # matches on its own context stream, which nothing orders behind them: wait on the host for the receives
            torch.cuda.current_stream(gdesc.device).synchronize()
        self.tab = self.e.table(gdesc, gcoor, gcounts)
        return sum(gcounts)

    def _rest(self):
        taken = set(self.local_sel)
        return [k for k in range(len(self.my_pairs)) if k not in taken]

    def match(self, keep=True):
        if self.mh is not None:
            self.e.free(self.mh); self.mh = None
        rest = self._rest()
        rest_pairs = [self.my_pairs[k] for k in rest]
        if not keep:
            self.e.match_only(self.tab, rest_pairs)
            return None
        self.mh, rest_lists = self.e.match(self.tab, rest_pairs)
        self.lists = [None] * len(self.my_pairs)
        for k, m in zip(rest, rest_lists):
            self.lists[k] = m
        for k, m in zip(self.local_sel, self.local_lists or []):
            self.lists[k] = m
        return sum(len(m) for m in self.lists)

    def seeds(self, base_seed):
        return [(int(base_seed) + i * self.n + j) & 0xFFFFFFFF for i, j in self.my_pairs]

    def _joined(self):
        """The pairs matched during the exchange (on this rank's own features) and the rest (on the exchanged table) as ONE
        list on the exchanged table -- it holds the own images too, keypoint for keypoint, so the own pairs' match lists
        index it as they are -- behind one match handle: op_ransac_pairs costs about the same for 11 pairs as for 88, and a
        rank paid it twice.  -> (order of my_pairs indices, joined handle or None when one call covers everything anyway)"""
        rest = self._rest()
        if not rest or not self.local_sel or not hasattr(self.e, "concat"):
            return None, None
        return rest + list(self.local_sel), self.e.concat(self.mh, self.local_mh)

    def ransac(self, shapes_wh, base_seed=1):
        """shapes_wh: (w, h) per image id"""
        seeds = self.seeds(base_seed)
        rest = self._rest()
        out = [None] * len(self.my_pairs)
        order, joined = self._joined()
        if joined is not None:
            try:
                rr = self.e.ransac(self.tab, joined, [self.lists[k] for k in order], [self.my_pairs[k] for k in order], shapes_wh, [seeds[k] for k in order])
            finally:
                self.e.free(joined)
            for k, r in zip(order, rr):
                out[k] = r
            self.rres = out
            return sum(1 for r in self.rres if r["ok"])
        if rest:
            rr = self.e.ransac(self.tab, self.mh, [self.lists[k] for k in rest], [self.my_pairs[k] for k in rest], shapes_wh, [seeds[k] for k in rest])
            for k, r in zip(rest, rr):
                out[k] = r
        if self.local_sel:
            lshapes = [shapes_wh[g] for g in self.local_ids]
            rr = self.e.ransac(self.local_tab, self.local_mh, self.local_lists, self.local_pairs, lshapes, [seeds[k] for k in self.local_sel])
            for k, r in zip(self.local_sel, rr):
                out[k] = r
        self.rres = out
        return sum(1 for r in self.rres if r["ok"])

    def _pairs_np(self):
        """my_pairs as an (P, 2) int32 array, made once per deal (the RANSAC summary call marshals 8128 pairs per pass otherwise:
        list comprehensions and np.asarray of a list of tuples were ~3 ms of a config-5 job's 11-ms RANSAC phase)"""
        c = getattr(self, "_pairs_cache", None)
        if c is None or c[0] is not self.my_pairs:
            c = (self.my_pairs, np.ascontiguousarray(np.asarray(self.my_pairs, np.int32).reshape(-1, 2)))
            self._pairs_cache = c
        return c[1]

    def ransac_summary(self, shapes_wh, base_seed=1):
        """RANSAC over this rank's pairs without unpacking every pair into Python (HipEngine) -> (accepted pairs, inliers)"""
        pr = self._pairs_np()
        seeds = ((int(base_seed) + pr[:, 0].astype(np.int64) * self.n + pr[:, 1]) & 0xFFFFFFFF).astype(np.uint32)   # == self.seeds()
        sh = np.ascontiguousarray(np.asarray(shapes_wh, np.int32).reshape(-1, 2))
        if not self.local_sel:                       # every pair was matched in one call (N = 1, or no pair of two own images)
            return self.e.ransac_summary(self.tab, self.mh, pr, sh, seeds) if len(pr) else (0, 0)
        rest = np.asarray(self._rest(), np.int64)
        local = np.asarray(list(self.local_sel), np.int64)
        order, joined = self._joined()
        if joined is not None:
            try:
                o = np.asarray(order, np.int64)
                return self.e.ransac_summary(self.tab, joined, pr[o], sh, seeds[o])
            finally:
                self.e.free(joined)
        ok = inl = 0
        if len(rest):
            a, b = self.e.ransac_summary(self.tab, self.mh, pr[rest], sh, seeds[rest])
            ok += a; inl += b
        if len(local):
            a, b = self.e.ransac_summary(self.local_tab, self.local_mh, self.local_pairs, [shapes_wh[g] for g in self.local_ids], seeds[local])
            ok += a; inl += b
        return ok, inl

    def gather(self):
        """-> {(i, j): (matches (M,2) <idx in i, idx in j>, ransac vector | None)}, i < j, whole job"""
        extras = [ransac_extra(r) for r in self.rres] if getattr(self, "rres", None) is not None else None
        if self.world > 1:
            res = gather_match_results(self.my_pairs, self.lists, self.device, self.group, extras)
            return {p: (v if isinstance(v, tuple) else (v, None)) for p, v in res.items()}
        return {p: (m, extras[k] if extras else None) for k, (p, m) in enumerate(zip(self.my_pairs, self.lists))}

    def close(self):
        self._free_results()
