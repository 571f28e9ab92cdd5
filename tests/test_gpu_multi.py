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
