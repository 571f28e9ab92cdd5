"""ctypes binding of ``libopenpano_hip.so`` (the C-ABI in ``include/openpano_hip.h``).

There is deliberately no fallback: if the HIP library is missing or no gfx950 device is
present, loading / creating a context raises.  Device buffers can be handed over as raw
pointers (e.g. ``torch.Tensor.data_ptr()``), PyTorch is never imported here.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OPENPANO_HIP_LIB") or os.path.join(_HERE, "libopenpano_hip.so")     # override: A/B builds of the same ABI (scripts/)


class OpenPanoHipError(RuntimeError):
    pass


class OpConfig(C.Structure):
    """``op_config`` (include/openpano_hip.h), POD snapshot of ``namespace config``."""
    _fields_ = [
        ("SIFT_WORKING_SIZE", C.c_int), ("NUM_OCTAVE", C.c_int), ("NUM_SCALE", C.c_int),
        ("SCALE_FACTOR", C.c_float), ("GAUSS_SIGMA", C.c_float), ("GAUSS_WINDOW_FACTOR", C.c_int),
        ("JUDGE_EXTREMA_DIFF_THRES", C.c_float), ("CONTRAST_THRES", C.c_float),
        ("PRE_COLOR_THRES", C.c_float), ("EDGE_RATIO", C.c_float),
        ("CALC_OFFSET_DEPTH", C.c_int), ("OFFSET_THRES", C.c_float),
        ("ORI_RADIUS", C.c_float), ("ORI_HIST_SMOOTH_COUNT", C.c_int),
        ("DESC_HIST_SCALE_FACTOR", C.c_int), ("DESC_INT_FACTOR", C.c_int),
        ("MATCH_REJECT_NEXT_RATIO", C.c_float), ("RANSAC_ITERATIONS", C.c_int),
        ("RANSAC_INLIER_THRES", C.c_double),
        ("INLIER_IN_MATCH_RATIO", C.c_float), ("INLIER_IN_POINTS_RATIO", C.c_float),
        ("CYLINDER", C.c_int), ("TRANS", C.c_int), ("ESTIMATE_CAMERA", C.c_int),
        ("ORDERED_INPUT", C.c_int), ("LAZY_READ", C.c_int), ("MULTIBAND", C.c_int),
        ("MAX_OUTPUT_SIZE", C.c_int), ("FOCAL_LENGTH", C.c_float),
    ]

    @classmethod
    def from_config(cls, cfg):
        c = cls()
        for name, _ in cls._fields_:
            setattr(c, name, getattr(cfg, name))
        return c


class OpImage(C.Structure):
    _fields_ = [("data", C.c_void_p), ("h", C.c_int), ("w", C.c_int), ("on_device", C.c_int), ("dtype", C.c_int)]


OP_F32, OP_U8 = 0, 1


class OpBlendImage(C.Structure):
    _fields_ = [("data", C.c_void_p), ("h", C.c_int), ("w", C.c_int), ("on_device", C.c_int),
                ("homo_inv", C.c_double * 9), ("range", C.c_double * 4), ("mat_h", C.c_int), ("mat_w", C.c_int)]


class OpBlendGeom(C.Structure):
    _fields_ = [("proj_method", C.c_int), ("proj_min", C.c_double * 2), ("proj_max", C.c_double * 2),
                ("resolution", C.c_double * 2)]


_lib = None


def _bind_torch_hip_runtime():
    """Share ONE HIP runtime with PyTorch.

    The PyTorch ROCm wheel bundles its own ``libamdhip64.so`` (same SONAME as /opt/rocm's).  Two
    runtime instances in one process cannot see each other's allocations or streams (and the
    second one finds no GPU), so when torch is installed its copy is loaded first -- by path,
    without importing torch -- and ``libopenpano_hip.so`` then resolves ``libamdhip64.so.7`` to
    it.  Without torch (plain C++ hosts) the system ROCm runtime is used.
    """
    import importlib.util
    import sys
    if os.environ.get("OPENPANO_SYSTEM_HIP"):
        return
    try:
        spec = sys.modules["torch"].__spec__ if "torch" in sys.modules else importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=os.RTLD_NOW | os.RTLD_GLOBAL)


def lib():
    """Load the HIP library (once). Raises if it has not been built -- no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OpenPanoHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C openpano_amd/csrc`; openpano_amd has no CPU fallback")
    _bind_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    L.op_last_error.restype = C.c_char_p
    L.op_abi_version.restype = C.c_int
    L.op_config_default.argtypes = [C.POINTER(OpConfig)]
    L.op_ctx_create.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    L.op_ctx_destroy.argtypes = [C.c_void_p]
    L.op_ctx_sync.argtypes = [C.c_void_p]
    L.op_ctx_set_profiling.argtypes = [C.c_void_p, C.c_int]
    if hasattr(L, "op_ctx_profile_only"):
        L.op_ctx_profile_only.argtypes = [C.c_void_p, C.c_char_p]
    L.op_ctx_profile_reset.argtypes = [C.c_void_p]
    L.op_ctx_profile_count.argtypes = [C.c_void_p]
    L.op_ctx_profile_get.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_long)]
    L.op_sift_batch.argtypes = [C.c_void_p, C.POINTER(OpConfig), C.POINTER(OpImage), C.c_int, C.POINTER(C.c_void_p)]
    if hasattr(L, "op_sift_batch_host"):
        L.op_sift_batch_host.argtypes = [C.c_void_p, C.POINTER(OpConfig), C.POINTER(OpImage), C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]
    L.op_features_num_images.argtypes = [C.c_void_p]
    L.op_features_count.argtypes = [C.c_void_p, C.c_int]
    L.op_features_offset.restype = C.c_int64
    L.op_features_offset.argtypes = [C.c_void_p, C.c_int]
    L.op_features_total.restype = C.c_int64
    L.op_features_total.argtypes = [C.c_void_p]
    L.op_features_desc_device.restype = C.c_void_p
    L.op_features_desc_device.argtypes = [C.c_void_p]
    L.op_features_coor_device.restype = C.c_void_p
    L.op_features_coor_device.argtypes = [C.c_void_p]
    L.op_features_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.op_features_copy_real.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.op_features_from_host.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
    L.op_features_from_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
    if hasattr(L, "op_features_adopt_device"):
        L.op_features_adopt_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
    L.op_features_free.argtypes = [C.c_void_p]
    L.op_sift_staged.argtypes = [C.c_void_p, C.POINTER(OpConfig), C.POINTER(OpImage), C.POINTER(C.c_void_p)]
    L.op_sift_dump_free.argtypes = [C.c_void_p]
    L.op_sift_dump_working_dims.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.op_sift_dump_octave_dims.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.op_sift_dump_plane.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.op_sift_dump_raw_count.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.op_sift_dump_raw.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.op_sift_dump_kp_count.argtypes = [C.c_void_p, C.c_int]
    L.op_sift_dump_kp.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.op_sift_dump_desc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.op_debug_math.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.op_debug_set_raw_capacity.argtypes = [C.c_void_p, C.c_int]
    L.op_debug_set_desc_list_cap.argtypes = [C.c_void_p, C.c_int]
    if hasattr(L, "op_match_pairs"):
        L.op_match_pairs.argtypes = [C.c_void_p, C.POINTER(OpConfig), C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.op_matches_count.argtypes = [C.c_void_p, C.c_int]
        L.op_matches_copy.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        if hasattr(L, "op_matches_copy_all"):
            L.op_matches_copy_all.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        if hasattr(L, "op_matches_device_list"):
            L.op_matches_device_list.restype = C.c_void_p
        if hasattr(L, "op_matches_device_list"):
            L.op_matches_device_list.argtypes = [C.c_void_p]
        L.op_matches_total.restype = C.c_int64
        L.op_matches_total.argtypes = [C.c_void_p]
        L.op_matches_free.argtypes = [C.c_void_p]
    L.op_group_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
    L.op_group_destroy.argtypes = [C.c_void_p]
    L.op_group_size.argtypes = [C.c_void_p]
    L.op_group_ctx.restype = C.c_void_p
    L.op_group_ctx.argtypes = [C.c_void_p, C.c_int]
    L.op_sift_batch_multi.argtypes = [C.c_void_p, C.POINTER(OpConfig), C.POINTER(OpImage), C.c_int, C.POINTER(C.c_void_p)]
    L.op_match_pairs_multi.argtypes = [C.c_void_p, C.POINTER(OpConfig), C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    if hasattr(L, "op_ransac_pairs_multi"):
        L.op_ransac_pairs_multi.argtypes = [C.c_void_p, C.POINTER(OpConfig), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                            C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
    L.op_matches_from_host.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
    if hasattr(L, "op_matches_concat"):
        L.op_matches_concat.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
    L.op_ransac_pairs.argtypes = [C.c_void_p, C.POINTER(OpConfig), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                  C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
    L.op_ransac_ok.argtypes = [C.c_void_p, C.c_int]
    L.op_ransac_confidence.restype = C.c_float
    L.op_ransac_confidence.argtypes = [C.c_void_p, C.c_int]
    L.op_ransac_homo.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.op_ransac_inlier_count.argtypes = [C.c_void_p, C.c_int]
    L.op_ransac_inliers.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.op_ransac_best.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    if hasattr(L, "op_ransac_summary"):
        L.op_ransac_summary.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int64)]
    L.op_ransac_free.argtypes = [C.c_void_p]
    if hasattr(L, "op_pairwise_table"):
        L.op_pairwise_table_size.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int64)]
        L.op_pairwise_table.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.op_blend_prepare.argtypes = [C.POINTER(OpConfig), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                   C.POINTER(OpBlendGeom), C.c_void_p, C.c_void_p]
    L.op_blend_canvas_dims.argtypes = [C.POINTER(OpBlendGeom), C.POINTER(OpBlendImage), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.op_blend.argtypes = [C.c_void_p, C.POINTER(OpConfig), C.POINTER(OpBlendGeom), C.POINTER(OpBlendImage), C.c_int, C.POINTER(C.c_void_p)]
    L.op_canvas_dims.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.op_canvas_device.restype = C.c_void_p
    L.op_canvas_device.argtypes = [C.c_void_p]
    L.op_canvas_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.op_canvas_free.argtypes = [C.c_void_p]
    L.op_canvas_crop.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.op_canvas_copy_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.op_cyl_warp_shape.argtypes = [C.POINTER(OpConfig), C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_int,
                                    C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p]
    L.op_cyl_warp.argtypes = [C.c_void_p, C.POINTER(OpConfig), C.POINTER(OpImage), C.c_double, C.POINTER(C.c_void_p)]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise OpenPanoHipError(f"libopenpano_hip error {rc}: {lib().op_last_error().decode(errors='replace')}")


class Context:
    """``op_ctx``: one per (process, device); ``stream`` is an optional raw ``hipStream_t``."""

    def __init__(self, device: int = 0, stream: int | None = None):
        self.handle = C.c_void_p()
        check(lib().op_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(self.handle)))
        self.device = device

    def sync(self):
        check(lib().op_ctx_sync(self.handle))

    def set_profiling(self, enable=True, only=None):
        """per-stage HIP-event timing on this context's stream; only = one stage label to bracket (None: all)"""
        L = lib()
        if hasattr(L, "op_ctx_profile_only"):
            check(L.op_ctx_profile_only(self.handle, only.encode() if only else None))
        check(L.op_ctx_set_profiling(self.handle, int(enable)))

    def profile_reset(self):
        check(lib().op_ctx_profile_reset(self.handle))

    def profile(self):
        """{stage label: (total_ms, calls)} measured with HIP events on this context's stream"""
        out = {}
        for i in range(lib().op_ctx_profile_count(self.handle)):
            lab = C.c_char_p(); ms = C.c_double(); calls = C.c_long()
            check(lib().op_ctx_profile_get(self.handle, i, C.byref(lab), C.byref(ms), C.byref(calls)))
            out[lab.value.decode()] = (ms.value, calls.value)
        return out

    def set_desc_list_cap(self, floats: int):
        """test hook: list arena one sorting pass of the descriptor kernel may use (op_debug_set_desc_list_cap)"""
        check(lib().op_debug_set_desc_list_cap(self.handle, int(floats)))

    def set_raw_capacity(self, cap: int):
        """test hook: per-image capacity of the speculative raw / refined lists (op_debug_set_raw_capacity)"""
        check(lib().op_debug_set_raw_capacity(self.handle, int(cap)))

    def close(self):
        if self.handle:
            lib().op_ctx_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _BorrowedContext(Context):
    """a context owned by an op_group (never destroyed from Python)"""

    def __init__(self, handle, device):
        self.handle = C.c_void_p(handle)
        self.device = device

    def close(self):
        self.handle = C.c_void_p()


class Group:
    """``op_group``: several GPUs driven from this process; a device may be listed twice."""

    def __init__(self, devices):
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        self.handle = C.c_void_p()
        check(lib().op_group_create(devs, len(devices), C.byref(self.handle)))
        self.devices = list(devices)
        self.ctx0 = _BorrowedContext(lib().op_group_ctx(self.handle, 0), devices[0])

    def sift_batch(self, cfg, images) -> "Features":
        arr, keep = _mk_images(images)
        ccfg = OpConfig.from_config(cfg)
        h = C.c_void_p()
        check(lib().op_sift_batch_multi(self.handle, C.byref(ccfg), arr, len(images), C.byref(h)))
        del keep
        return Features(self.ctx0, h)

    def match_pairs_handle(self, cfg, feats, pairs) -> "Matches":
        pr = np.ascontiguousarray(np.asarray(pairs, np.int32).reshape(-1, 2))
        ccfg = OpConfig.from_config(cfg)
        h = C.c_void_p()
        check(lib().op_match_pairs_multi(self.handle, C.byref(ccfg), feats.handle, pr.ctypes.data_as(C.c_void_p), len(pr), C.byref(h)))
        return Matches(h, len(pr))

    def ransac_pairs(self, cfg, feats, matches, pairs, shapes_wh, seeds=None, base_seed=0):
        """op_ransac_pairs_multi -> the same list of dicts as hip.ransac_pairs"""
        pr = np.ascontiguousarray(np.asarray(pairs, np.int32).reshape(-1, 2))
        sh = np.ascontiguousarray(np.asarray(shapes_wh, np.int32).reshape(-1, 2))
        sd = np.ascontiguousarray(np.asarray(seeds, np.uint32)) if seeds is not None else None
        ccfg = OpConfig.from_config(cfg)
        h = C.c_void_p()
        check(lib().op_ransac_pairs_multi(self.handle, C.byref(ccfg), feats.handle, matches.handle, pr.ctypes.data_as(C.c_void_p), len(pr),
                                          sh.ctypes.data_as(C.c_void_p), sd.ctypes.data_as(C.c_void_p) if sd is not None else None,
                                          int(base_seed), C.byref(h)))
        return _unpack_ransac(h, len(pr))

    def close(self):
        if self.handle:
            self.ctx0.close()
            lib().op_group_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _mk_images(images):
    """images: numpy HWC arrays (float32 in [0,1], or uint8 decoder bytes), or device buffers as
    (device_ptr, h, w) / (device_ptr, h, w, "u8") tuples."""
    arr = (OpImage * len(images))()
    keep = []
    for i, im in enumerate(images):
        if isinstance(im, tuple):
            ptr, h, w = im[:3]
            dt = OP_U8 if len(im) > 3 and im[3] in ("u8", OP_U8, np.uint8) else OP_F32
            arr[i] = OpImage(C.c_void_p(int(ptr)), int(h), int(w), 1, dt)
        else:
            im = np.asarray(im)
            a = np.ascontiguousarray(im) if im.dtype == np.uint8 else np.ascontiguousarray(im, np.float32)
            if a.ndim != 3 or a.shape[2] != 3:
                raise ValueError("image must be H x W x 3 (float32 or uint8)")
            keep.append(a)
            arr[i] = OpImage(a.ctypes.data_as(C.c_void_p), a.shape[0], a.shape[1], 0, OP_U8 if a.dtype == np.uint8 else OP_F32)
    return arr, keep


class Features:
    """``op_features``: device-resident descriptors/coordinates of a batch of images."""

    def __init__(self, ctx: Context, handle):
        self.ctx = ctx
        self.handle = handle

    @property
    def num_images(self):
        return lib().op_features_num_images(self.handle)

    def count(self, i):
        return lib().op_features_count(self.handle, i)

    @property
    def total(self):
        return lib().op_features_total(self.handle)

    def offset(self, i):
        return lib().op_features_offset(self.handle, i)

    @property
    def desc_ptr(self):
        return lib().op_features_desc_device(self.handle)

    @property
    def coor_ptr(self):
        return lib().op_features_coor_device(self.handle)

    def get(self, i):
        k = self.count(i)
        desc = np.empty((k, 128), np.float32); coor = np.empty((k, 2), np.float64)
        check(lib().op_features_copy(self.ctx.handle, self.handle, i, desc.ctypes.data_as(C.c_void_p), coor.ctypes.data_as(C.c_void_p)))
        return desc, coor

    def get_real(self, i):
        """real_coor in [0,1) of image i (what do_detect_feature itself returns, feature.cc:31-47)"""
        k = self.count(i)
        real = np.empty((k, 2), np.float64)
        check(lib().op_features_copy_real(self.ctx.handle, self.handle, i, real.ctypes.data_as(C.c_void_p)))
        return real

    @classmethod
    def from_host(cls, ctx, descs, coors=None):
        n = len(descs)
        descs = [np.ascontiguousarray(d, np.float32).reshape(-1, 128) for d in descs]
        dp = (C.c_void_p * n)(*[d.ctypes.data_as(C.c_void_p) for d in descs])
        if coors is not None:
            coors = [np.ascontiguousarray(c, np.float64).reshape(-1, 2) for c in coors]
            cp = (C.c_void_p * n)(*[c.ctypes.data_as(C.c_void_p) for c in coors])
        else:
            cp = None
        counts = (C.c_int * n)(*[len(d) for d in descs])
        h = C.c_void_p()
        check(lib().op_features_from_host(ctx.handle, dp, cp, counts, n, C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_device(cls, ctx, desc_ptr, counts, coor_ptr=None):
        """flat device buffer (images back to back, total x 128 fp32) -> Features (D2D copy)"""
        n = len(counts)
        cc = (C.c_int * n)(*[int(c) for c in counts])
        h = C.c_void_p()
        check(lib().op_features_from_device(ctx.handle, C.c_void_p(int(desc_ptr)), C.c_void_p(int(coor_ptr)) if coor_ptr else None, cc, n, C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def adopt_device(cls, ctx, desc_ptr, counts, coor_ptr, keep=None):
        """flat device buffers (images back to back) -> Features WITHOUT a copy; `keep` (e.g. the torch tensors that own
        the memory) is held until free()"""
        n = len(counts)
        cc = (C.c_int * n)(*[int(c) for c in counts])
        h = C.c_void_p()
        check(lib().op_features_adopt_device(ctx.handle, C.c_void_p(int(desc_ptr)), C.c_void_p(int(coor_ptr)), cc, n, C.byref(h)))
        f = cls(ctx, h)
        f._keep = keep
        return f

    def desc_device_array(self):
        """object exposing ``__cuda_array_interface__`` over the device descriptor buffer
        (zero-copy view for torch.as_tensor(..., device='cuda'); valid while self is alive)"""
        class _View:
            pass
        v = _View()
        v.__cuda_array_interface__ = {"shape": (int(self.total), 128), "typestr": "<f4",
                                      "data": (int(self.desc_ptr or 0), False), "version": 2, "strides": None}
        v._owner = self
        return v

    def coor_device_array(self):
        """same for the (total, 2) float64 keypoint coordinates"""
        class _View:
            pass
        v = _View()
        v.__cuda_array_interface__ = {"shape": (int(self.total), 2), "typestr": "<f8",
                                      "data": (int(self.coor_ptr or 0), False), "version": 2, "strides": None}
        v._owner = self
        return v

    def free(self):
        if self.handle:
            lib().op_features_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def sift_batch(ctx: Context, cfg, images) -> Features:
    arr, keep = _mk_images(images)
    ccfg = OpConfig.from_config(cfg)
    h = C.c_void_p()
    check(lib().op_sift_batch(ctx.handle, C.byref(ccfg), arr, len(images), C.byref(h)))
    del keep
    return Features(ctx, h)


class SiftCall:
    """op_sift_batch with its arguments marshalled once: a host program calling the C-ABI keeps its
    op_image array and op_config around; rebuilding them in Python costs ~35 us per call, which is
    2 % of a 38-image batch on this GPU."""

    def __init__(self, ctx: Context, cfg, images):
        self.ctx = ctx
        self.arr, self.keep = _mk_images(images)
        self.n = len(images)
        self.ccfg = OpConfig.from_config(cfg)
        self._fn = lib().op_sift_batch

    def __call__(self) -> Features:
        h = C.c_void_p()
        check(self._fn(self.ctx.handle, C.byref(self.ccfg), self.arr, self.n, C.byref(h)))
        return Features(self.ctx, h)


class SiftHostCall:
    """op_sift_batch_host with its arguments marshalled once: host images in, descriptors / coordinates into the given
    host buffers (raw pointers, e.g. pinned torch tensors' data_ptr()), transfers overlapped with the kernels."""

    def __init__(self, ctx: Context, cfg, images, desc_ptr, coor_ptr, capacity_rows):
        self.ctx = ctx
        self.arr, self.keep = _mk_images(images)
        self.n = len(images)
        self.ccfg = OpConfig.from_config(cfg)
        self.desc_ptr = C.c_void_p(int(desc_ptr)) if desc_ptr else None
        self.coor_ptr = C.c_void_p(int(coor_ptr)) if coor_ptr else None
        self.cap = int(capacity_rows)

    def __call__(self) -> Features:
        h = C.c_void_p()
        rc = lib().op_sift_batch_host(self.ctx.handle, C.byref(self.ccfg), self.arr, self.n, self.desc_ptr, self.coor_ptr, self.cap, C.byref(h))
        if rc != 0 and h:
            lib().op_features_free(h)           # OP_ERR_CAPACITY still hands over the resident features
        check(rc)
        return Features(self.ctx, h)


def sift_staged(ctx: Context, cfg, image, planes=True):
    """Staged single-image run -> object with the same fields as tests' ``SiftStages``."""
    L = lib()
    arr, keep = _mk_images([image])
    ccfg = OpConfig.from_config(cfg)
    hd = C.c_void_p()
    check(L.op_sift_staged(ctx.handle, C.byref(ccfg), arr, C.byref(hd)))

    class Stages:
        pass

    st = Stages()
    try:
        h, w = C.c_int(), C.c_int()
        L.op_sift_dump_working_dims(hd, C.byref(h), C.byref(w))
        st.work = np.empty((h.value, w.value, 3), np.float32)
        check(L.op_sift_dump_plane(ctx.handle, hd, 4, 0, 0, st.work.ctypes.data_as(C.c_void_p)))
        st.dims = []; st.grey = {}; st.dog = {}; st.mag = {}; st.ort = {}; st.raw = {}
        for o in range(cfg.NUM_OCTAVE):
            check(L.op_sift_dump_octave_dims(hd, o, C.byref(h), C.byref(w)))
            st.dims.append((h.value, w.value))
            if planes:
                def grab(kind, s):
                    buf = np.empty((h.value, w.value), np.float32)
                    check(L.op_sift_dump_plane(ctx.handle, hd, kind, o, s, buf.ctypes.data_as(C.c_void_p)))
                    return buf
                st.grey[o] = grab(5, 0)
                for s in range(cfg.NUM_SCALE - 1):
                    st.dog[(o, s)] = grab(1, s)
                for s in range(1, cfg.NUM_SCALE - 2):
                    st.mag[(o, s)] = grab(2, s)
                    st.ort[(o, s)] = grab(3, s)
            for s in range(1, cfg.NUM_SCALE - 2):
                n = L.op_sift_dump_raw_count(hd, o, s)
                xy = np.empty((n, 2), np.int32)
                if n:
                    L.op_sift_dump_raw(hd, o, s, xy.ctypes.data_as(C.c_void_p))
                st.raw[(o, s)] = xy
        for which, name in ((0, "refined"), (1, "oriented")):
            n = L.op_sift_dump_kp_count(hd, which)
            ints = np.empty((n, 4), np.int32); real = np.empty((n, 2), np.float64); fl = np.empty((n, 2), np.float32)
            if n:
                L.op_sift_dump_kp(hd, which, ints.ctypes.data_as(C.c_void_p), real.ctypes.data_as(C.c_void_p), fl.ctypes.data_as(C.c_void_p))
            setattr(st, name, dict(ints=ints, real=real, fl=fl))
        k = L.op_sift_dump_kp_count(hd, 1)
        st.desc = np.empty((k, 128), np.float32); st.coor = np.empty((k, 2), np.float64)
        if k:
            L.op_sift_dump_desc(hd, st.desc.ctypes.data_as(C.c_void_p), st.coor.ctypes.data_as(C.c_void_p))
    finally:
        L.op_sift_dump_free(hd)
    del keep
    return st


def debug_math(ctx: Context, which: int, x, y=None):
    x = np.ascontiguousarray(x, np.float32)
    y = np.ascontiguousarray(y, np.float32) if y is not None else x
    out = np.empty_like(x)
    check(lib().op_debug_math(ctx.handle, which, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), x.size, out.ctypes.data_as(C.c_void_p)))
    return out


class Matches:
    """``op_matches`` handle (kept for the RANSAC stage)."""

    def __init__(self, handle, npairs):
        self.handle = handle; self.npairs = npairs

    def get(self, p):
        """(M, 2) int32 <idx in first, idx in second> of pair p."""
        L = lib()
        n = L.op_matches_count(self.handle, p)
        a = np.empty((n, 2), np.int32)
        if n:
            check(L.op_matches_copy(self.handle, p, a.ctypes.data_as(C.c_void_p)))
        return a

    def lists(self):
        """every pair's (M, 2) int32 list: one library call, one D2H of the resident lists"""
        L = lib()
        offs = np.empty(self.npairs + 1, np.int64)
        check(L.op_matches_copy_all(self.handle, None, offs.ctypes.data_as(C.c_void_p)))
        flat = np.empty((int(offs[-1]), 2), np.int32)
        if len(flat):
            check(L.op_matches_copy_all(self.handle, flat.ctypes.data_as(C.c_void_p), None))
        return [flat[offs[p]: offs[p + 1]] for p in range(self.npairs)]

    @property
    def total(self):
        return int(lib().op_matches_total(self.handle))

    @classmethod
    def from_host(cls, lists):
        lists = [np.ascontiguousarray(a, np.int32).reshape(-1, 2) for a in lists]
        n = len(lists)
        ptrs = (C.c_void_p * n)(*[a.ctypes.data_as(C.c_void_p) for a in lists])
        counts = (C.c_int * n)(*[len(a) for a in lists])
        h = C.c_void_p()
        check(lib().op_matches_from_host(ptrs, counts, n, C.byref(h)))
        return cls(h, n)

    @classmethod
    def concat(cls, ctx, a, b):
        """a's pairs followed by b's as one handle (op_matches_concat): one op_ransac_pairs call for both"""
        h = C.c_void_p()
        check(lib().op_matches_concat(ctx.handle, a.handle, b.handle, C.byref(h)))
        return cls(h, a.npairs + b.npairs)

    def free(self):
        if self.handle:
            lib().op_matches_free(self.handle); self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def match_pairs_handle(ctx: Context, cfg, feats: Features, pairs) -> Matches:
    pr = np.ascontiguousarray(np.asarray(pairs, np.int32).reshape(-1, 2))
    ccfg = OpConfig.from_config(cfg)
    h = C.c_void_p()
    check(lib().op_match_pairs(ctx.handle, C.byref(ccfg), feats.handle, pr.ctypes.data_as(C.c_void_p), len(pr), C.byref(h)))
    return Matches(h, len(pr))


def ransac_pairs(ctx: Context, cfg, feats: Features, matches: Matches, pairs, shapes_wh, seeds=None, base_seed=0):
    """Batched TransformEstimation::get_transform. shapes_wh: (n_images, 2) of (w, h).
    -> list of dict(ok, confidence, homo (3,3), inliers, best_hyp, best_count) per pair."""
    L = lib()
    pr = np.ascontiguousarray(np.asarray(pairs, np.int32).reshape(-1, 2))
    sh = np.ascontiguousarray(np.asarray(shapes_wh, np.int32).reshape(-1, 2))
    sd = np.ascontiguousarray(np.asarray(seeds, np.uint32)) if seeds is not None else None
    ccfg = OpConfig.from_config(cfg)
    h = C.c_void_p()
    check(L.op_ransac_pairs(ctx.handle, C.byref(ccfg), feats.handle, matches.handle, pr.ctypes.data_as(C.c_void_p), len(pr),
                            sh.ctypes.data_as(C.c_void_p), sd.ctypes.data_as(C.c_void_p) if sd is not None else None,
                            int(base_seed), C.byref(h)))
    return _unpack_ransac(h, len(pr))


def _unpack_ransac(h, npairs):
    L = lib()
    out = []
    try:
        for p in range(npairs):
            homo = np.zeros(9, np.float64)
            check(L.op_ransac_homo(h, p, homo.ctypes.data_as(C.c_void_p)))
            n = L.op_ransac_inlier_count(h, p)
            inl = np.empty(n, np.int32)
            if n:
                check(L.op_ransac_inliers(h, p, inl.ctypes.data_as(C.c_void_p)))
            bh = C.c_int(); bc = C.c_int()
            check(L.op_ransac_best(h, p, C.byref(bh), C.byref(bc)))
            out.append(dict(ok=bool(L.op_ransac_ok(h, p)), confidence=L.op_ransac_confidence(h, p), homo=homo.reshape(3, 3),
                            inliers=inl, best_hyp=bh.value, best_count=bc.value))
    finally:
        L.op_ransac_free(h)
    return out


def ransac_pairs_summary(ctx: Context, cfg, feats: Features, matches: Matches, pairs, shapes_wh, base_seed=0, seeds=None):
    """op_ransac_pairs without unpacking every pair into Python: -> (accepted pairs, total inliers)."""
    L = lib()
    pr = np.ascontiguousarray(np.asarray(pairs, np.int32).reshape(-1, 2))
    sh = np.ascontiguousarray(np.asarray(shapes_wh, np.int32).reshape(-1, 2))
    sd = np.ascontiguousarray(np.asarray(seeds, np.uint32)) if seeds is not None else None
    ccfg = OpConfig.from_config(cfg)
    h = C.c_void_p()
    check(L.op_ransac_pairs(ctx.handle, C.byref(ccfg), feats.handle, matches.handle, pr.ctypes.data_as(C.c_void_p), len(pr),
                            sh.ctypes.data_as(C.c_void_p), sd.ctypes.data_as(C.c_void_p) if sd is not None else None,
                            int(base_seed), C.byref(h)))
    ok = C.c_int(); inl = C.c_int64()
    check(L.op_ransac_summary(h, C.byref(ok), C.byref(inl)))
    L.op_ransac_free(h)
    return ok.value, inl.value


def ransac_pairwise_table(ctx: Context, cfg, feats: Features, matches: Matches, pairs, shapes_wh, base_seed=0, seeds=None, table_pairs=None):
    """op_ransac_pairs + op_pairwise_table: RANSAC of every pair and Stitcher::match_image's bookkeeping (stitcher.cc:79-93)
    without unpacking a pair into Python.  -> (ij (E, 2) int32, conf (E,) float32, homo (E, 9) float64, cnt (E,) int32,
    pts (sum cnt, 4) float64, accepted pairs): the arguments of pano_estimate_cameras; E = 2 x accepted pairs."""
    L = lib()
    pr = np.ascontiguousarray(np.asarray(pairs, np.int32).reshape(-1, 2))
    sh = np.ascontiguousarray(np.asarray(shapes_wh, np.int32).reshape(-1, 2))
    sd = np.ascontiguousarray(np.asarray(seeds, np.uint32)) if seeds is not None else None
    ccfg = OpConfig.from_config(cfg)
    h = C.c_void_p()
    check(L.op_ransac_pairs(ctx.handle, C.byref(ccfg), feats.handle, matches.handle, pr.ctypes.data_as(C.c_void_p), len(pr),
                            sh.ctypes.data_as(C.c_void_p), sd.ctypes.data_as(C.c_void_p) if sd is not None else None,
                            int(base_seed), C.byref(h)))
    try:
        ne = C.c_int(); npt = C.c_int64()
        check(L.op_pairwise_table_size(h, C.byref(ne), C.byref(npt)))
        E, P = ne.value, npt.value
        ij = np.zeros((max(E, 1), 2), np.int32); conf = np.zeros(max(E, 1), np.float32); homo = np.zeros((max(E, 1), 9), np.float64)
        cnt = np.zeros(max(E, 1), np.int32); pts = np.zeros((max(P, 1), 4), np.float64)
        tp = pr if table_pairs is None else np.ascontiguousarray(np.asarray(table_pairs, np.int32).reshape(-1, 2))   # (tests: a list other than the call's is refused)
        check(L.op_pairwise_table(ctx.handle, feats.handle, matches.handle, h, tp.ctypes.data_as(C.c_void_p), len(tp),
                                  ij.ctypes.data_as(C.c_void_p), conf.ctypes.data_as(C.c_void_p), homo.ctypes.data_as(C.c_void_p),
                                  cnt.ctypes.data_as(C.c_void_p), pts.ctypes.data_as(C.c_void_p)))
        return ij[:E], conf[:E], homo[:E], cnt[:E], pts[:P], E // 2
    finally:
        L.op_ransac_free(h)


def match_pairs(ctx: Context, cfg, feats: Features, pairs):
    """All requested image pairs in one call -> list of (M, 2) int32 arrays of
    <idx in image i, idx in image j>, sorted by (first, second) (``MatchData``, matcher.hh:14-25)."""
    L = lib()
    pr = np.ascontiguousarray(np.asarray(pairs, np.int32).reshape(-1, 2))
    ccfg = OpConfig.from_config(cfg)
    h = C.c_void_p()
    check(L.op_match_pairs(ctx.handle, C.byref(ccfg), feats.handle, pr.ctypes.data_as(C.c_void_p), len(pr), C.byref(h)))
    mh = Matches(h, len(pr))
    try:
        return [a.copy() for a in mh.lists()]
    finally:
        mh.free()


class Canvas:
    """``op_canvas``: device-resident H x W x 3 fp32 result of a blend / warp."""

    def __init__(self, ctx: Context, handle):
        self.ctx = ctx; self.handle = handle
        h, w = C.c_int(), C.c_int()
        check(lib().op_canvas_dims(handle, C.byref(h), C.byref(w)))
        self.h, self.w = h.value, w.value

    @property
    def device_ptr(self):
        return lib().op_canvas_device(self.handle)

    def numpy(self):
        out = np.empty((self.h, self.w, 3), np.float32)
        check(lib().op_canvas_copy(self.ctx.handle, self.handle, out.ctypes.data_as(C.c_void_p)))
        return out

    def crop(self):
        """crop() of the reference (lib/imgproc.cc:200-235) -> (Canvas, (x0, y0))"""
        h = C.c_void_p(); x0, y0 = C.c_int(), C.c_int()
        check(lib().op_canvas_crop(self.ctx.handle, self.handle, C.byref(h), C.byref(x0), C.byref(y0)))
        return Canvas(self.ctx, h), (x0.value, y0.value)

    def numpy_u8(self):
        """write_rgb quantisation on the device, bytes over PCIe"""
        out = np.empty((self.h, self.w, 3), np.uint8)
        check(lib().op_canvas_copy_u8(self.ctx.handle, self.handle, out.ctypes.data_as(C.c_void_p)))
        return out

    def free(self):
        if self.handle:
            lib().op_canvas_free(self.handle); self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def blend_prepare(cfg, shapes_wh, homos, proj_method, identity_idx):
    """Host geometry of ``ConnectedImages`` (calc_inverse_homo, update_proj_range,
    get_final_resolution: stitcher_image.cc:36-114) -> (OpBlendGeom, homo_inv (n,9), ranges (n,4))."""
    sh = np.ascontiguousarray(np.asarray(shapes_wh, np.int32).reshape(-1, 2))
    n = len(sh)
    homo = np.ascontiguousarray(np.asarray(homos, np.float64).reshape(n, 9))
    ccfg = OpConfig.from_config(cfg)
    g = OpBlendGeom(); hinv = np.zeros((n, 9), np.float64); ranges = np.zeros((n, 4), np.float64)
    check(lib().op_blend_prepare(C.byref(ccfg), int(proj_method), int(identity_idx), n, sh.ctypes.data_as(C.c_void_p),
                                 homo.ctypes.data_as(C.c_void_p), C.byref(g), hinv.ctypes.data_as(C.c_void_p),
                                 ranges.ctypes.data_as(C.c_void_p)))
    return g, hinv, ranges


class BlendCall:
    """``ConnectedImages::blend()`` (stitcher_image.cc:116-155) with the op_blend_geom / op_blend_image arrays marshalled
    once, like a C host holds them; every call is one op_blend.
    images: numpy HWC float32 arrays or (device_ptr, h, w); homos: n x 3 x 3 ImageComponent::homo."""

    def __init__(self, ctx: Context, cfg, images, homos, proj_method, identity_idx):
        self.ctx = ctx
        n = self.n = len(images)
        arr_img, self._keep = _mk_images(images)
        shapes = [(arr_img[i].w, arr_img[i].h) for i in range(n)]
        self.geom, hinv, ranges = blend_prepare(cfg, shapes, homos, proj_method, identity_idx)
        arr = self.arr = (OpBlendImage * n)()
        for i in range(n):
            arr[i].data = arr_img[i].data; arr[i].h = arr_img[i].h; arr[i].w = arr_img[i].w; arr[i].on_device = arr_img[i].on_device
            for k in range(9):
                arr[i].homo_inv[k] = hinv[i, k]
            for k in range(4):
                arr[i].range[k] = ranges[i, k]
        self.ccfg = OpConfig.from_config(cfg)
        self._fn = lib().op_blend

    def __call__(self) -> Canvas:
        h = C.c_void_p()
        check(self._fn(self.ctx.handle, C.byref(self.ccfg), C.byref(self.geom), self.arr, self.n, C.byref(h)))
        return Canvas(self.ctx, h)


def blend(ctx: Context, cfg, images, homos, proj_method, identity_idx) -> Canvas:
    """``ConnectedImages::blend()`` (stitcher_image.cc:116-155) on the device.
    images: numpy HWC float32 arrays or (device_ptr, h, w); homos: n x 3 x 3 ImageComponent::homo."""
    return BlendCall(ctx, cfg, images, homos, proj_method, identity_idx)()


def cyl_warp_shape(cfg, w, h, h_factor, pts=None):
    """Host part of ``CylinderWarper::warp`` -> (new_w, new_h, offset (2,), warped centred pts)."""
    ccfg = OpConfig.from_config(cfg)
    p = np.ascontiguousarray(pts, np.float64).reshape(-1, 2).copy() if pts is not None else np.zeros((0, 2))
    nw, nh = C.c_int(), C.c_int(); off = np.zeros(2)
    check(lib().op_cyl_warp_shape(C.byref(ccfg), int(w), int(h), float(h_factor), p.ctypes.data_as(C.c_void_p) if len(p) else None,
                                  len(p), C.byref(nw), C.byref(nh), off.ctypes.data_as(C.c_void_p)))
    return nw.value, nh.value, off, p


def cyl_warp(ctx: Context, cfg, image, h_factor) -> Canvas:
    arr, keep = _mk_images([image])
    ccfg = OpConfig.from_config(cfg)
    h = C.c_void_p()
    check(lib().op_cyl_warp(ctx.handle, C.byref(ccfg), arr, float(h_factor), C.byref(h)))
    del keep
    return Canvas(ctx, h)
