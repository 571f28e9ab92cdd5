"""GPU parity tests of the HIP SIFT path (through the C-ABI) against the CPU oracle and the
committed golden vectors.  Bit-exact at every stage: planes, keypoint lists, descriptors."""
import glob
import os

import numpy as np
import pytest

from openpano_amd import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def ctx():
    from openpano_amd import hip
    c = hip.Context(0)
    yield c
    c.close()


def u8_to_f32(u8):
    return (u8.astype(np.float64) / 255.0).astype(np.float32)


def _view(h, w, seed):
    world = synth.make_world(seed, h + 48, w + 48, work_scale=1600.0 / (h + w))
    return synth.cut_view(world, 24, 24, h, w, seed)


def test_device_math_equals_libm(ctx, oracle):
    """device twins (csrc/devmath.hpp) == the glibc calls the reference makes, on this box"""
    from openpano_amd import hip
    rng = np.random.default_rng(0)
    n = 1 << 21
    neg = -np.abs(rng.standard_normal(n) * 4).astype(np.float32)
    neg[:4] = [0.0, -0.0, -1e-30, -20.0]
    assert np.array_equal(hip.debug_math(ctx, 0, neg), oracle.libm(0, neg))
    ang = (rng.random(n) * 6.2831855).astype(np.float32)
    ang[:6] = [0.0, 1e-5, 0.78539816, 0.7853982, 3.1415927, 6.2831855]
    assert np.array_equal(hip.debug_math(ctx, 1, ang), oracle.libm(1, ang))
    assert np.array_equal(hip.debug_math(ctx, 2, ang), oracle.libm(2, ang))
    x = (rng.standard_normal(n) * 0.1).astype(np.float32)
    y = (rng.standard_normal(n) * 0.1).astype(np.float32)
    x[:5] = [0, 0, 1e-7, -3e-7, 0.5]; y[:5] = [0, 1e-7, 0, 2e-7, -0.5]
    assert np.array_equal(hip.debug_math(ctx, 3, x, y), oracle.libm(3, x, y))
    assert np.array_equal(hip.debug_math(ctx, 4, x, y), oracle.libm(4, x, y))


def _compare_stages(g, o, cfg):
    assert g.dims == o.dims
    assert np.array_equal(g.work, o.work), "working image"
    for oc in range(cfg.NUM_OCTAVE):
        assert np.array_equal(g.grey[oc], o.gauss[(oc, 0)]), ("grey", oc)
    for kind in ("dog", "mag", "ort"):
        a, b = getattr(g, kind), getattr(o, kind)
        for k in a:
            assert np.array_equal(a[k], b[k]), (kind, k, int((a[k] != b[k]).sum()))
    for k in g.raw:
        assert np.array_equal(g.raw[k], o.raw[k]), ("raw", k)
    for nm in ("refined", "oriented"):
        a, b = getattr(g, nm), getattr(o, nm)
        for f in ("ints", "real", "fl"):
            assert np.array_equal(a[f], b[f]), (nm, f)
    assert np.array_equal(g.desc, o.desc), int((g.desc != o.desc).sum())
    assert np.array_equal(g.coor, o.coor)


VIEWS = [("cfg2_600x400", 400, 600, 22), ("cfg4_1300x867", 867, 1300, 38), ("odd_333x777", 333, 777, 5),
         ("strip_150x1250", 150, 1250, 9), ("tower_1250x150", 1250, 150, 10)]      # one band / one segment extremes of the row-streaming kernel


@pytest.mark.parametrize("name,h,w,seed", VIEWS, ids=[v[0] for v in VIEWS])
def test_staged_sift_bit_exact_vs_oracle(ctx, oracle, cfg, name, h, w, seed):
    from openpano_amd import hip
    img = _view(h, w, seed)
    g = hip.sift_staged(ctx, cfg, img)
    o = oracle.sift_stages(img)
    assert len(o.desc) > 200
    _compare_stages(g, o, cfg)


GOLDEN = sorted(glob.glob(os.path.join(HERE, "golden", "sift_*.npz"))) + \
    sorted(glob.glob(os.path.join(HERE, "golden", "nat_*x*.npz")))        # natural texture (tests/natural.py)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_sift_matches_reference_golden(ctx, cfg, path):
    """HIP path vs fixtures produced by the reference's own code (tests/golden/make_golden.py)"""
    from openpano_amd import hip
    gd = np.load(path)
    st = hip.sift_staged(ctx, cfg, u8_to_f32(gd["img"]), planes=False)
    raw = np.array([[len(st.raw[(o, s)]) for s in range(1, 5)] for o in range(4)], np.int32)
    assert np.array_equal(raw, gd["raw_counts"])
    assert np.array_equal(st.refined["ints"], gd["refined_ints"])
    assert np.array_equal(st.refined["real"], gd["refined_real"])
    assert np.array_equal(st.refined["fl"][:, 1], gd["refined_sf"])
    assert np.array_equal(st.oriented["ints"], gd["oriented_ints"])
    assert np.array_equal(st.oriented["fl"][:, 0], gd["oriented_dir"])
    assert np.array_equal(st.desc, gd["desc"])
    assert np.array_equal(st.coor, gd["coor"])


def test_batch_mixed_sizes_and_device_input(ctx, oracle, cfg):
    """op_sift_batch: several images per launch, two sizes, host and device-resident inputs"""
    import torch
    from openpano_amd import hip
    imgs = [_view(400, 600, 1), _view(300, 500, 2), _view(400, 600, 3), _view(300, 500, 4)]
    dev = [torch.from_numpy(im).cuda() for im in imgs[:2]]
    torch.cuda.synchronize()
    inputs = [(dev[0].data_ptr(), 400, 600), (dev[1].data_ptr(), 300, 500), imgs[2], imgs[3]]
    f = hip.sift_batch(ctx, cfg, inputs)
    assert f.num_images == 4
    tot = 0
    for i, im in enumerate(imgs):
        d, c = f.get(i)
        od, oc = oracle.detect_feature(im)
        assert np.array_equal(d, od) and np.array_equal(c, oc), i
        assert f.offset(i) == tot
        tot += len(d)
    assert f.total == tot
    f.free()


def test_full_size_batch_properties(ctx, oracle, cfg):
    """BASELINE config-4 sized batch (1300x867): size-independent properties + spot oracle check"""
    from openpano_amd import hip
    views = synth.image_set(6, 867, 1300, seed=38, overlap=0.45, rows=2)
    f = hip.sift_batch(ctx, cfg, views)
    f2 = hip.sift_batch(ctx, cfg, views[::-1])
    for i in range(6):
        d, c = f.get(i)
        assert len(d) > 300
        # RootSIFT: every descriptor is non-negative with L2 norm DESC_INT_FACTOR (sift.cc:40-43)
        assert (d >= 0).all()
        assert np.allclose(np.linalg.norm(d.astype(np.float64), axis=1), 512.0, rtol=1e-5)
        # coordinates are centred original-image pixels (feature.cc:23-26)
        assert (np.abs(c[:, 0]) <= 650).all() and (np.abs(c[:, 1]) <= 433.5).all()
        # batch composition / order does not change an image's features (idempotence)
        d2, c2 = f2.get(5 - i)
        assert np.array_equal(d, d2) and np.array_equal(c, c2)
    od, oc = oracle.detect_feature(views[3])
    d, c = f.get(3)
    assert np.array_equal(d, od) and np.array_equal(c, oc)
    f.free(); f2.free()


def test_errors_are_reported_not_swallowed(ctx, cfg):
    from openpano_amd import hip
    with pytest.raises(hip.OpenPanoHipError):
        hip.sift_batch(ctx, cfg, [np.zeros((1, 5, 3), np.float32)])
    # a flat image has no extrema: zero features is a valid (empty) result at this level; the
    # adapter turns it into the reference's error_exit (stitcherbase.cc:20-21)
    f = hip.sift_batch(ctx, cfg, [np.full((200, 300, 3), 0.5, np.float32)])
    assert f.count(0) == 0 and f.total == 0
    f.free()


def test_uint8_ingest_equals_fp32_path(ctx, oracle, cfg):
    """OP_U8 sources (decoder bytes, host or device) are converted on the device exactly like
    read_img, (float)byte / 255.0 (lib/imgio.cc:54-56): same features as the fp32 image, bit for bit."""
    import torch
    from openpano_amd import hip
    world = synth.make_world(61, 300, 420, work_scale=1600.0 / (240 + 320), density=900.0)
    u8 = [(synth.cut_view(world, 20 + 9 * k, 20 + 40 * k, 240, 320, k) * 255 + 0.5).astype(np.uint8) for k in range(3)]
    f32 = [(v.astype(np.float64) / 255.0).astype(np.float32) for v in u8]
    dev = torch.from_numpy(u8[2]).cuda(); torch.cuda.synchronize()
    fa = hip.sift_batch(ctx, cfg, [u8[0], u8[1], (dev.data_ptr(), 240, 320, "u8"), f32[0]])   # mixed element types in one batch
    for k in range(3):
        od, oc = oracle.detect_feature(f32[k])
        d, c = fa.get(k)
        assert len(d) > 50 and np.array_equal(d, od) and np.array_equal(c, oc), k
    d3, c3 = fa.get(3); d0, c0 = fa.get(0)
    assert np.array_equal(d3, d0) and np.array_equal(c3, c0)
    fa.free()


def test_config5_sized_image(ctx, oracle, cfg):
    """BASELINE config 5 input size: a 4000x3000 decoder-byte image (36 MB as uint8, 144 MB as
    fp32) is down-scaled to the 914x685 working image on the device; features equal the oracle's on
    the fp32 image bit for bit, through both element types."""
    import torch
    from openpano_amd import hip
    rng = np.random.default_rng(50000)
    # cheap large texture: a small seeded world up-sampled by pixel replication + per-pixel noise
    base = synth.make_world(505, 375, 500, work_scale=1.0, density=60.0)
    big = np.repeat(np.repeat(base, 8, axis=0), 8, axis=1)
    big = np.clip(big + rng.normal(0, 0.02, big.shape).astype(np.float32), 0, 1)
    u8 = (big * 255 + 0.5).astype(np.uint8)
    assert u8.shape == (3000, 4000, 3)
    f32 = (u8.astype(np.float64) / 255.0).astype(np.float32)
    od, oc = oracle.detect_feature(f32)
    assert len(od) > 300
    t = torch.from_numpy(u8).cuda(); torch.cuda.synchronize()
    f = hip.sift_batch(ctx, cfg, [(t.data_ptr(), 3000, 4000, "u8"), f32])
    for k in range(2):
        d, c = f.get(k)
        assert np.array_equal(d, od) and np.array_equal(c, oc), k
    f.free()


def test_concurrent_contexts(oracle, cfg):
    """The reference calls detect_feature concurrently from OpenMP threads (stitcherbase.cc:14);
    the C-ABI is thread-compatible: one op_ctx per host thread, calls overlap (ctypes drops the GIL)."""
    import threading
    from openpano_amd import hip
    world = synth.make_world(71, 330, 900, work_scale=1600.0 / (240 + 320), density=900.0)
    views = [synth.cut_view(world, 20 + 5 * k, 20 + 60 * k, 240, 320, 50 + k) for k in range(8)]
    want = [oracle.detect_feature(v) for v in views]
    results = [None] * 8
    errors = []

    def worker(t):
        try:
            c = hip.Context(0)
            for rep in range(3):
                for k in range(t, 8, 4):
                    f = hip.sift_batch(c, cfg, [views[k]])
                    results[k] = f.get(0); f.free()
            m = hip.match_pairs(c, cfg, hip.Features.from_host(c, [want[t][0], want[t + 4][0]]), [(0, 1)])[0]
            assert np.array_equal(m, oracle.match_exact(want[t][0], want[t + 4][0]))
            c.close()
        except Exception as e:          # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for k in range(8):
        assert np.array_equal(results[k][0], want[k][0]) and np.array_equal(results[k][1], want[k][1]), k


def test_raw_capacity_overflow_reruns_instead_of_failing(oracle, cfg):
    """The raw / refined candidate lists are speculative: with a capacity far below the
    image's candidate count the batch re-runs once with the observed size -- same features, no
    OP_ERR_CAPACITY -- and the context keeps the grown capacity."""
    from openpano_amd import hip
    c = hip.Context(0)
    try:
        c.set_raw_capacity(64)
        imgs = [_view(400, 600, 1), _view(400, 600, 3)]
        f = hip.sift_batch(c, cfg, imgs)
        for i, im in enumerate(imgs):
            d, co = f.get(i)
            od, oc = oracle.detect_feature(im)
            assert len(d) > 300 and np.array_equal(d, od) and np.array_equal(co, oc), i
        f.free()
        f = hip.sift_batch(c, cfg, imgs[::-1])            # steady state after the growth
        assert np.array_equal(f.get(1)[0], oracle.detect_feature(imgs[0])[0])
        f.free()
        # a lowered PRE_COLOR_THRES (more raw candidates, same survivors' arithmetic) through the same path
        from openpano_amd.config import PanoConfig
        c.set_raw_capacity(64)
        loose = PanoConfig(PRE_COLOR_THRES=0.01)
        from checkers import Oracle
        f = hip.sift_batch(c, loose, imgs[:1])
        od, oc = Oracle(loose).detect_feature(imgs[0])
        assert np.array_equal(f.get(0)[0], od)
        f.free()
    finally:
        c.close()


def test_descriptor_two_pass_sort_equals_one_pass(oracle, cfg):
    """k_descriptor sorts a batch of 64 window samples by bin into a 640-float LDS arena (sift.cc:110-146: the order of
    every bin's additions is part of the result); a batch whose padded lists would not fit takes two passes of 32
    samples.  No image of the suite needs that, so it is forced here (arena 0: always two passes; 96 floats: a mix)
    -- same descriptors, bit for bit."""
    from openpano_amd import hip
    imgs = [_view(400, 600, 11), _view(400, 600, 12)]
    want = [oracle.detect_feature(im) for im in imgs]
    for arena in (0, 96, 640):
        c = hip.Context(0)
        try:
            c.set_desc_list_cap(arena)
            f = hip.sift_batch(c, cfg, imgs)
            for i in range(2):
                d, co = f.get(i)
                assert len(d) > 300 and np.array_equal(d, want[i][0]) and np.array_equal(co, want[i][1]), (arena, i)
            f.free()
        finally:
            c.close()


def test_host_images_at_a_constant_stride_travel_in_one_copy(ctx, oracle, cfg):
    """Host images that are slices of one array (np.stack: stride == image size; a padded pool: larger stride) take the
    single strided H2D copy of op_sift_batch; separately allocated images take one copy each.  Same features."""
    from openpano_amd import hip
    views = synth.image_set(3, 240, 320, seed=9, overlap=0.5)
    want = [oracle.detect_feature(v) for v in views]
    stack = np.ascontiguousarray(np.stack(views))                        # stride == image bytes
    pool = np.zeros((3, 240 * 320 * 3 + 77), np.float32)                 # a larger constant stride, not a multiple of 256 bytes
    pooled = []
    for k, v in enumerate(views):
        pool[k, : v.size] = v.reshape(-1)
        pooled.append(pool[k, : v.size].reshape(v.shape))
    for imgs in ([stack[0], stack[1], stack[2]], pooled, [v.copy() for v in views]):
        f = hip.sift_batch(ctx, cfg, imgs)
        for i in range(3):
            d, c = f.get(i)
            assert np.array_equal(d, want[i][0]) and np.array_equal(c, want[i][1]), i
        f.free()


def test_pipelined_host_call_equals_the_plain_call(ctx, cfg):
    """op_sift_batch_host (uploads / kernels / copy back pipelined over chunks) returns the features op_sift_batch
    returns, resident and in the caller's host buffers; a host buffer that is too small is OP_ERR_CAPACITY."""
    from openpano_amd import hip
    views = synth.image_set(17, 240, 320, seed=11, overlap=0.5)                 # 17 images -> 2 chunks of 9 + 8
    plain = hip.sift_batch(ctx, cfg, views)
    total = int(plain.total)
    for imgs in (views, [(v * 255 + 0.5).astype(np.uint8) for v in views]):
        ref = plain if imgs is views else hip.sift_batch(ctx, cfg, imgs)
        total = int(ref.total)
        hd = np.zeros((total + 8, 128), np.float32); hc = np.zeros((total + 8, 2), np.float64)
        f = hip.SiftHostCall(ctx, cfg, imgs, hd.ctypes.data, hc.ctypes.data, total + 8)()
        assert f.num_images == 17 and int(f.total) == total
        for i in range(17):
            d, c = f.get(i); rd, rc = ref.get(i)
            assert np.array_equal(d, rd) and np.array_equal(c, rc), i
            o = f.offset(i)
            assert np.array_equal(hd[o: o + len(d)], rd) and np.array_equal(hc[o: o + len(d)], rc), i
            assert np.array_equal(f.get_real(i), ref.get_real(i))
        f.free()
        with pytest.raises(hip.OpenPanoHipError):
            hip.SiftHostCall(ctx, cfg, imgs, hd.ctypes.data, hc.ctypes.data, total - 1)()
        if ref is not plain:
            ref.free()
    plain.free()


def test_dense_extrema_fill_the_workgroup_lists(ctx, cfg):
    """k_pyramid_rows collects a workgroup's raw extrema in a 192-entry LDS list and appends them to the image's list with one
    atomic (csrc/pyramid.hip); further ones would go straight to the image's list.  Blurred noise is the densest field of DoG
    extrema there is -- about one per fifty octave pixels and layer, i.e. up to ~115 in a 240 x 24 segment, so the overflow branch
    is out of reach of the shipped Gaussian bank -- and a down-scaled noise image under a low PRE_COLOR_THRES fills the lists
    to more than half (12 k raw extrema per image): every stage still equals the oracle's, raw lists included."""
    from openpano_amd import hip
    from openpano_amd.config import PanoConfig
    from checkers import Oracle
    rng = np.random.default_rng(12)
    img = np.repeat(rng.random((900, 1300, 1), dtype=np.float32), 3, axis=2)
    img = np.ascontiguousarray(0.25 + 0.5 * img)
    loose = PanoConfig(PRE_COLOR_THRES=2e-3)
    o = Oracle(loose).sift_stages(img)
    h0, w0 = o.dims[0]
    per_item = np.zeros(((h0 + 23) // 24, (w0 + 239) // 240), np.int64)
    for s in range(1, 5):
        xy = np.asarray(o.raw[(0, s)]).reshape(-1, 2)
        np.add.at(per_item, (xy[:, 1] // 24, xy[:, 0] // 240), 1)
    assert per_item.max() > 96 and sum(len(np.asarray(v).reshape(-1, 2)) for v in o.raw.values()) > 10000
    g = hip.sift_staged(ctx, loose, img)
    _compare_stages(g, o, loose)


@pytest.mark.parametrize("flags", ["-DOP_RW_RAWCAP=4", "-DOP_RW_PACKABLE_BELOW=64"], ids=["lds-list-of-4", "unpackable"])
def test_raw_extrema_overflow_branches_of_the_row_kernel(oracle, cfg, tmp_path, flags):
    """k_pyramid_rows' emit_raw has two branches the shipped constants never reach: the workgroup's LDS list is
    full (> 192 raw extrema in a 240 x 24 segment) and the octave is too large to pack a position into 26 bits (>= 8192 px).
    Both then append straight to the image's list with a global atomic.  A variant of the library compiled with the knobs
    of csrc/pyramid.hip -- an LDS list of FOUR entries / packing refused from 64 px -- runs the same two views in a
    subprocess: descriptors and coordinates equal the oracle's."""
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    csrc = os.path.join(root, "openpano_amd", "csrc")
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fopenmp", "-I" + os.path.join(root, "include"), "-I" + csrc]
    objs = []
    jobs = []
    for src in sorted(glob.glob(os.path.join(csrc, "*.hip"))):
        stem = os.path.basename(src)[:-4]
        prebuilt = os.path.join(csrc, stem + ".o")
        if stem != "pyramid" and os.path.exists(prebuilt):
            objs.append(prebuilt)
            continue
        o = str(tmp_path / (stem + ".o"))
        jobs.append(subprocess.Popen(base + ([flags] if stem == "pyramid" else []) + ["-c", src, "-o", o]))
        objs.append(o)
    for src in sorted(glob.glob(os.path.join(csrc, "*.cc"))):           # host-only translation units of the library (plain g++, csrc/Makefile)
        stem = os.path.basename(src)[:-3]
        prebuilt = os.path.join(csrc, stem + ".o")
        if os.path.exists(prebuilt):
            objs.append(prebuilt)
            continue
        o = str(tmp_path / (stem + ".o"))
        jobs.append(subprocess.Popen(["g++", "-std=c++17", "-O3", "-ffp-contract=off", "-fPIC", "-Wno-psabi", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + csrc, "-c", src, "-o", o]))
        objs.append(o)
    assert all(j.wait() == 0 for j in jobs)
    lib = str(tmp_path / "libopenpano_hip_variant.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fopenmp", "-o", lib] + objs)
    views = [_view(400, 600, 1), _view(300, 500, 5)]
    np.savez(tmp_path / "in.npz", a=views[0], b=views[1])
    code = ("import sys, numpy as np; sys.path.insert(0, %r)\n"
            "from openpano_amd import hip\nfrom openpano_amd.config import PanoConfig\n"
            "assert hip.LIB_PATH == %r\n"
            "z = np.load(%r); c = hip.Context(0); out = {}\n"
            "for k in ('a', 'b'):\n"
            "    f = hip.sift_batch(c, PanoConfig(), [z[k]]); d, co = f.get(0); out[k + '_d'] = d; out[k + '_c'] = co; f.free()\n"
            "np.savez(%r, **out); c.close()\n") % (root, lib, str(tmp_path / "in.npz"), str(tmp_path / "out.npz"))
    env = dict(os.environ, OPENPANO_HIP_LIB=lib)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.load(tmp_path / "out.npz")
    for k, im in zip("ab", views):
        od, oc = oracle.detect_feature(im)
        assert len(od) > 200 and np.array_equal(got[k + "_d"], od) and np.array_equal(got[k + "_c"], oc), k
