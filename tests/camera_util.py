"""Shared by the camera-estimation tests: ctypes front-ends of the two implementations with one
signature (oracle/_ref's reference classes vs openpano_amd/libpano_host.so) and a synthetic
rotating-camera scene that produces the pairwise MatchInfo table Stitcher::pairwise_match leaves."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_SO = os.path.join(ROOT, "openpano_amd", "libpano_host.so")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


class CameraImpl:
    """prefix 'pano_' (product host library) or 'ref_' (reference compiled in place)."""

    def __init__(self, lib, prefix):
        self.lib, self.p = lib, prefix
        g = lambda n: getattr(lib, prefix + n)   # noqa: E731
        g("estimate_cameras").argtypes = [C.c_int, _i32p, C.c_int, _i32p, _f32p, _f64p, _i32p, _f64p, _f64p]
        g("rotation_to_angle").argtypes = [_f64p, _f64p]
        g("angle_to_rotation").argtypes = [_f64p, _f64p]
        g("homography_inverse").argtypes = [_f64p, _f64p]
        g("colpiv_solve").argtypes = [_f64p, C.c_int, _f64p, _f64p]
        g("config_set").argtypes = [C.c_char_p, C.c_float]
        g("iba_probe").argtypes = [C.c_int, _f64p, C.c_int, _i32p, _i32p, _f64p, C.c_int, _f64p, _f64p, _f64p]

    def config(self, **kv):
        for k, v in kv.items():
            assert getattr(self.lib, self.p + "config_set")(k.encode(), float(v)) == 0, k

    def estimate(self, shapes_wh, table):
        """table: list of (i, j, conf, homo(9), pts(M,4)) -> cameras (n, 13): focal, aspect, ppx, ppy, R."""
        n = len(shapes_wh)
        ij = np.array([[t[0], t[1]] for t in table], np.int32).reshape(-1, 2)
        conf = np.array([t[2] for t in table], np.float32)
        homo = np.ascontiguousarray(np.stack([np.asarray(t[3], np.float64).reshape(9) for t in table]))
        cnt = np.array([len(t[4]) for t in table], np.int32)
        pts = np.ascontiguousarray(np.concatenate([np.asarray(t[4], np.float64).reshape(-1, 4) for t in table] + [np.zeros((0, 4))]))
        out = np.zeros((n, 13), np.float64)
        rc = getattr(self.lib, self.p + "estimate_cameras")(n, np.ascontiguousarray(shapes_wh, np.int32).reshape(-1), len(table),
                                                            ij.reshape(-1).copy(), conf, homo.reshape(-1).copy(), cnt, pts.reshape(-1).copy(), out.reshape(-1))
        assert rc == 0
        return out

    def lm_step(self, cams, entries, identity):
        """One Levenberg-Marquardt step of IncrementalBundleAdjuster on the given cameras (n, 13):
        entries = [(from, to, pts(M,4))] as add_match takes them.  Returns (residuals, damped JtJ, update)."""
        ids = sorted({e[0] for e in entries} | {e[1] for e in entries}); ni = len(ids)
        ij = np.array([[e[0], e[1]] for e in entries], np.int32).reshape(-1)
        cnt = np.array([len(e[2]) for e in entries], np.int32)
        pts = np.ascontiguousarray(np.concatenate([np.asarray(e[2], np.float64).reshape(-1, 4) for e in entries])).reshape(-1)
        M = int(cnt.sum()); resid = np.zeros(2 * M); jtj = np.zeros(36 * ni * ni); upd = np.zeros(6 * ni)
        getattr(self.lib, self.p + "iba_probe")(len(cams), np.ascontiguousarray(cams, np.float64).reshape(-1), len(entries), ij, cnt, pts,
                                               identity, resid, jtj, upd)
        return resid, jtj.reshape(6 * ni, 6 * ni), upd

    def rotation_to_angle(self, r):
        v = np.zeros(3); getattr(self.lib, self.p + "rotation_to_angle")(np.ascontiguousarray(r, np.float64).reshape(9), v); return v

    def angle_to_rotation(self, v):
        r = np.zeros(9); getattr(self.lib, self.p + "angle_to_rotation")(np.ascontiguousarray(v, np.float64), r); return r.reshape(3, 3)

    def inverse(self, a):
        r = np.zeros(9); ok = getattr(self.lib, self.p + "homography_inverse")(np.ascontiguousarray(a, np.float64).reshape(9), r)
        return bool(ok), r.reshape(3, 3)

    def solve(self, A, b):
        x = np.zeros(len(b)); getattr(self.lib, self.p + "colpiv_solve")(np.ascontiguousarray(A, np.float64).reshape(-1), len(b), np.ascontiguousarray(b, np.float64), x)
        return x


def host_impl():
    if not os.path.exists(HOST_SO):
        import __graft_entry__ as g
        g.build()
    return CameraImpl(C.CDLL(HOST_SO), "pano_")


def ref_impl(ref):
    return CameraImpl(ref.lib, "ref_")


def rot(yaw, pitch, roll):
    cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    Rz = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])
    return Rz @ Rx @ Ry


def rotating_camera_scene(seed, n=7, rows=1, w=600, h=400, focal=880.0, step_deg=13.0, noise=0.4, npts=260):
    """n views of a camera rotating about its centre.  Returns (shapes_wh, table, true cameras):
    table holds BOTH directions of every overlapping pair like Stitcher::pairwise_match does
    (stitcher.cc:79-93): matches[a][b] = {homo: b -> a, match: (point in a, point in b)}."""
    rng = np.random.default_rng(seed)
    cols = -(-n // rows)
    Rs = []
    for k in range(n):
        r, c = divmod(k, cols)
        Rs.append(rot(np.deg2rad((c - (cols - 1) / 2) * step_deg + rng.normal(0, 0.6)),
                      np.deg2rad((r - (rows - 1) / 2) * 9.0 + rng.normal(0, 0.5)), np.deg2rad(rng.normal(0, 0.8))))
    K = np.array([[focal, 0, 0], [0, focal, 0], [0, 0, 1.0]])
    Ki = np.linalg.inv(K)
    table = []
    for a in range(n):
        for b in range(a + 1, n):
            Hab = K @ Rs[a] @ Rs[b].T @ Ki                      # b -> a, centred coordinates
            pb = np.stack([rng.uniform(-w / 2, w / 2, npts), rng.uniform(-h / 2, h / 2, npts), np.ones(npts)], 1)
            pa = pb @ Hab.T
            if np.any(pa[:, 2] <= 1e-6):
                continue
            pa = pa[:, :2] / pa[:, 2:3]
            keep = (np.abs(pa[:, 0]) < w / 2 - 2) & (np.abs(pa[:, 1]) < h / 2 - 2)
            if keep.sum() < 24:
                continue
            pa = pa[keep] + rng.normal(0, noise, (keep.sum(), 2)); pb2 = pb[keep, :2] + rng.normal(0, noise, (keep.sum(), 2))
            conf = float(np.float32(keep.sum() / (8 + 0.3 * npts)))
            Hn = Hab / Hab[2, 2]
            table.append((a, b, conf, Hn.reshape(9), np.concatenate([pa, pb2], 1)))
            Hi = np.linalg.inv(Hn); Hi = Hi / Hi[2, 2]
            table.append((b, a, conf, Hi.reshape(9), np.concatenate([pb2, pa], 1)))
    shapes = np.array([[w, h]] * n, np.int32)
    return shapes, table, (focal, Rs)


def reprojection_rms(cams, table):
    """RMS of from - H(to) over the table with H = K_from R_from R_to^T K_to^-1 (IBA::calcError)."""
    se, cnt = 0.0, 0
    for i, j, conf, homo, pts in table:
        # entry [i][j]: first = point in i, second = point in j; IBA pairs it as to = i, from = j
        Kt = np.array([[cams[i, 0], 0, cams[i, 2]], [0, cams[i, 0] * cams[i, 1], cams[i, 3]], [0, 0, 1]]); Rt = cams[i, 4:].reshape(3, 3)
        Kf = np.array([[cams[j, 0], 0, cams[j, 2]], [0, cams[j, 0] * cams[j, 1], cams[j, 3]], [0, 0, 1]]); Rf = cams[j, 4:].reshape(3, 3)
        H = Kf @ Rf @ Rt.T @ np.linalg.inv(Kt)
        p = np.concatenate([pts[:, :2], np.ones((len(pts), 1))], 1) @ H.T
        d = pts[:, 2:] - p[:, :2] / p[:, 2:3]
        se += float((d * d).sum()); cnt += 2 * len(pts)
    return np.sqrt(se / max(cnt, 1))
