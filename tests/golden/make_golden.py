"""Generate tests/golden/*.npz from the REFERENCE's own code (oracle/_ref).

Run in the build container (needs /root/reference to have built oracle/_ref):
    python tests/golden/make_golden.py
The reference ships no golden vectors, so these fixtures -- outputs of the
unmodified reference sources on small seeded inputs -- are what pins the C oracle
(tests/test_oracle_golden.py, runs anywhere) and, through it, the HIP path.

Each sift_*.npz holds: the uint8 input, octave dims, per-(octave,scale) raw-extrema counts,
refined/oriented keypoints, descriptors, and CRC32s of every working/DoG/mag/ort plane.
match_*.npz holds two descriptor sets and the exact matcher's pair list.
"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

from openpano_amd.config import PanoConfig  # noqa: E402
from openpano_amd import synth  # noqa: E402
from checkers import Ref  # noqa: E402


def u8_to_f32(u8):
    # read_png (lib/imgio.cc:43-60): (float)byte / 255.0 evaluated in double
    return (u8.astype(np.float64) / 255.0).astype(np.float32)


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


def sift_case(ref, name, img_u8):
    st = ref.sift_stages(u8_to_f32(img_u8))
    desc, coor = ref.detect_feature(u8_to_f32(img_u8))
    out = dict(
        img=img_u8, dims=np.array(st.dims, np.int32), work_crc=np.uint32(crc(st.work)),
        raw_counts=np.array([[len(st.raw[(o, s)]) for s in range(1, 5)] for o in range(4)], np.int32),
        refined_ints=st.refined["ints"], refined_real=st.refined["real"], refined_sf=st.refined["fl"][:, 1],
        oriented_ints=st.oriented["ints"], oriented_dir=st.oriented["fl"][:, 0],
        desc=st.desc, coor=st.coor, n_detect=np.int32(len(desc)),
    )
    for kind in ("dog", "mag", "ort"):
        planes = getattr(st, kind)
        keys = sorted(planes)
        out[kind + "_keys"] = np.array(keys, np.int32)
        out[kind + "_crc"] = np.array([crc(planes[k]) for k in keys], np.uint32)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, img_u8.shape, "raw", out["raw_counts"].sum(), "refined", len(st.refined["ints"]), "desc", len(st.desc))
    return st


def main():
    cfg = PanoConfig()
    ref = Ref(cfg)
    # three small views: landscape, portrait-ish odd sizes, and a tiny one that is up-scaled 4x
    w1 = synth.make_world(101, 300, 520, work_scale=1600.0 / (240 + 320), density=900.0)
    a = (synth.cut_view(w1, 20, 20, 240, 320, 1) * 255 + 0.5).astype(np.uint8)
    b = (synth.cut_view(w1, 30, 150, 240, 320, 2) * 255 + 0.5).astype(np.uint8)
    w2 = synth.make_world(202, 330, 260, work_scale=1600.0 / (301 + 227), density=1500.0)
    c = (synth.cut_view(w2, 10, 10, 301, 227, 3) * 255 + 0.5).astype(np.uint8)
    sa = sift_case(ref, "sift_a_240x320", a)
    sb = sift_case(ref, "sift_b_240x320", b)
    sift_case(ref, "sift_c_301x227", c)
    # a down-scaled, finely textured view: exercises offset iterations, contrast and edge rejections
    w3 = synth.make_world(303, 540, 740, work_scale=1600.0 / (500 + 700), density=250.0)
    d = (synth.cut_view(w3, 20, 20, 500, 700, 4) * 255 + 0.5).astype(np.uint8)
    sift_case(ref, "sift_d_500x700", d)
    pairs = ref.match_exact(sa.desc, sb.desc)
    np.savez_compressed(os.path.join(HERE, "match_ab.npz"), pairs=pairs,
                        n1=np.int32(len(sa.desc)), n2=np.int32(len(sb.desc)))
    print("match_ab", len(sa.desc), len(sb.desc), "->", len(pairs))
    blend_case(ref)
    ransac_case(ref, sa, sb, pairs)
    camera_case(ref)
    natural_case(ref)


def ransac_case(ref, sa, sb, pairs):
    """TransformEstimation::get_transform of the reference (mt19937 seed injected through the
    random_device seam of oracle/ref_driver.cc) on the golden match list of views a/b (320x240):
    homography mode and, with CYLINDER=1, the 7-point affine mode."""
    ca = (sa.coor - 0.5) * np.array([320.0, 240.0]); cb = (sb.coor - 0.5) * np.array([320.0, 240.0])   # feature.cc:23-26
    out = dict(match=pairs, coor_a=ca, coor_b=cb, shape=np.array([320, 240], np.int32))
    for mode, cfgkv in (("homo", dict()), ("affine", dict(CYLINDER=1, ESTIMATE_CAMERA=0, ORDERED_INPUT=1))):
        ref.set_config(**cfgkv)
        for seed in (7, 20240917):
            r = ref.ransac(pairs, ca, cb, (320, 240), (320, 240), seed)
            out[f"{mode}_{seed}_ok"] = np.int32(r["ok"]); out[f"{mode}_{seed}_conf"] = np.float32(r["confidence"])
            out[f"{mode}_{seed}_homo"] = r["homo"]; out[f"{mode}_{seed}_pts"] = r["inlier_pts"]
            print("ransac", mode, seed, r["ok"], r["confidence"], len(r["inlier_pts"]))
        ref.set_config(CYLINDER=0, ESTIMATE_CAMERA=1, ORDERED_INPUT=0)
    np.savez_compressed(os.path.join(HERE, "ransac_ab.npz"), **out)


def blend_case(ref):
    """ConnectedImages::blend of the reference on a small spherical bundle: LinearBlender
    (LAZY_READ 0, the deterministic branch -- SURVEY F5) and MultiBandBlender(3), 1 thread."""
    views, homos = synth.pano_scene(3, 80, 112, seed=404, proj="camera")
    u8 = [(v * 255 + 0.5).astype(np.uint8) for v in views]
    f32 = [u8_to_f32(v) for v in u8]
    ref.lib.ref_set_threads(1)
    ref.set_config(LAZY_READ=0, ORDERED_INPUT=0, MULTIBAND=0)
    lin, meta = ref.blend(f32, homos, 2, 1)
    ref.set_config(MULTIBAND=3)
    mb, _ = ref.blend(f32, homos, 2, 1)
    ref.set_config(MULTIBAND=0, LAZY_READ=1)
    np.savez_compressed(os.path.join(HERE, "blend_sph_linear.npz"), views=np.stack(u8), homos=homos,
                        identity_idx=np.int32(1), canvas_linear=lin, canvas_multiband3=mb,
                        geom=meta["geom"], ranges=meta["ranges"], homo_inv=meta["homo_inv"])
    print("blend_sph_linear", lin.shape, "covered", float((lin[..., 0] >= 0).mean()))


def camera_case(ref):
    """CameraEstimator::estimate of the reference (its Eigen calls go to the stand-in of
    oracle/ref_shim -- DESIGN.md, "parity unpinned at Eigen") on the pairwise MatchInfo table of a
    synthetic rotating-camera scene, in the shipped configuration (MULTIPASS_BA 1, STRAIGHTEN 1,
    LM_LAMBDA 5) and with the one-shot bundle adjustment (MULTIPASS_BA 0, no straightening)."""
    from camera_util import ref_impl, rotating_camera_scene
    refc = ref_impl(ref)
    shapes, table, (focal, Rs) = rotating_camera_scene(11, n=8, rows=2, step_deg=15.0, npts=160)
    out = dict(shapes=shapes, ij=np.array([[t[0], t[1]] for t in table], np.int32), conf=np.array([t[2] for t in table], np.float32),
               homo=np.stack([t[3] for t in table]), cnt=np.array([len(t[4]) for t in table], np.int32),
               pts=np.concatenate([t[4] for t in table]), focal=np.float64(focal))
    for name, mode in (("shipped", dict(MULTIPASS_BA=1, STRAIGHTEN=1, LM_LAMBDA=5.0)), ("oneshot", dict(MULTIPASS_BA=0, STRAIGHTEN=0, LM_LAMBDA=5.0))):
        refc.config(**mode)
        out["cameras_" + name] = refc.estimate(shapes, table)
    refc.config(MULTIPASS_BA=1, STRAIGHTEN=1, LM_LAMBDA=5.0)
    np.savez_compressed(os.path.join(HERE, "camera_scene.npz"), **out)
    print("camera_scene", len(shapes), "images", len(table) // 2, "pairs", int(out["cnt"].sum()) // 2, "matches; focal", out["cameras_shipped"][:, 0].round(2))


def natural_case(ref):
    """Natural texture: crops of the reference's published panoramas, cut by
    tests/natural.py.  The decoded uint8 crop is stored with the reference's outputs, so the golden
    does not depend on the JPEG decoder of the machine running the tests.  nat_match_uav.npz: the
    exact matcher + TransformEstimation (seed injected) on the config-1 pair."""
    import natural
    v1 = natural.config_views(1)
    sa = sift_case(ref, "nat_uav_a_400x600", v1[0])
    sb = sift_case(ref, "nat_uav_b_400x600", v1[1])
    sift_case(ref, "nat_cmu_400x600", natural.config_views(2, 3)[2])
    sift_case(ref, "nat_uav_867x1300", natural.config_views(4, 3)[2])
    pairs = ref.match_exact(sa.desc, sb.desc)
    ca = (sa.coor - 0.5) * np.array([600.0, 400.0]); cb = (sb.coor - 0.5) * np.array([600.0, 400.0])
    out = dict(pairs=pairs, coor_a=ca, coor_b=cb, shape=np.array([600, 400], np.int32))
    for mode, cfgkv in (("homo", dict()), ("affine", dict(CYLINDER=1, ESTIMATE_CAMERA=0, ORDERED_INPUT=1))):
        ref.set_config(**cfgkv)
        r = ref.ransac(pairs, ca, cb, (600, 400), (600, 400), 38)
        out[f"{mode}_ok"] = np.int32(r["ok"]); out[f"{mode}_conf"] = np.float32(r["confidence"])
        out[f"{mode}_homo"] = r["homo"]; out[f"{mode}_pts"] = r["inlier_pts"]
        print("nat ransac", mode, r["ok"], r["confidence"], len(r["inlier_pts"]))
        ref.set_config(CYLINDER=0, ESTIMATE_CAMERA=1, ORDERED_INPUT=0)
    np.savez_compressed(os.path.join(HERE, "nat_match_uav.npz"), **out)
    print("nat_match_uav", len(sa.desc), len(sb.desc), "->", len(pairs))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "camera":
        camera_case(Ref(PanoConfig()))
    elif len(sys.argv) > 1 and sys.argv[1] == "natural":
        natural_case(Ref(PanoConfig()))
    else:
        main()
